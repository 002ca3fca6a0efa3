#!/bin/bash
# round 6: the shipped stagger (Q4 persistent launches of a context without lanes) -- suite, A/B against the tree before it, per-item trace of the merged + staggered persistent kernel
export TMPDIR=/tmp
O=gpurun_out/b9; mkdir -p $O; rm -f $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export EFFORT_HIP_LIB=build/variants/premerge.so; else unset EFFORT_HIP_LIB; fi
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --tag q4x32-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag q4x32x4-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 24 --reps 2 --tag q4x24-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --tag q4x16-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
done
done
cat $O/ab.txt
for t in "8,2,16" "8,1,8"; do
  EFFORT_HIP_LIB=build/variants/labmerge12.so timeout 300 python tools/timeline.py --q4 1 --groups 16 --replay 10 --tune $t --persistent 2 --out $O/tl.json 2>&1 | grep -v amdgpu | head -22 | tee $O/timeline_q4_labmerge12_$(echo $t | tr , _).txt
done
EFFORT_HIP_LIB=build/variants/labmerge12.so timeout 300 python tools/timeline.py --q4 1 --groups 32 --replay 10 --out $O/tl.json 2>&1 | grep -v amdgpu | head -22 | tee $O/timeline_q4_labmerge12_32.txt
