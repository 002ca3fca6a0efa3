#!/bin/bash
# round 6: does merging the outliers into the PERSISTENT Q4 kernel's streaming loop pay?  (premerge.so = the tree before it); the persistent Q4 launch per item
export TMPDIR=/tmp
O=gpurun_out/b6; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export EFFORT_HIP_LIB=build/variants/premerge.so; else unset EFFORT_HIP_LIB; fi
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --tag q4x32-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --overlap 4 --steps-per-graph 8 --tag q4x16x4-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag q4x32x4-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --configs "8,2,16:-1" --tag q4x16p-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
done
done
cat $O/ab.txt
unset EFFORT_HIP_LIB
timeout 300 python tools/timeline.py --q4 1 --groups 16 --replay 10 --tune 8,2,16 --persistent 2 --out $O/tl.json 2>&1 | grep -v amdgpu | head -26 | tee $O/timeline_q4_persistent_768.txt
timeout 300 python tools/timeline.py --q4 1 --groups 32 --replay 10 --out $O/tl.json 2>&1 | grep -v amdgpu | head -26 | tee $O/timeline_q4_32.txt
