#!/bin/bash
# round 6: the same late start for the second persistent workgroup of a CU in the FP16 launches (build/variants/staggerallD.so)
export TMPDIR=/tmp
O=gpurun_out/b8; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do
for v in tree staggerall6 staggerall12 staggerall20; do
  if [ $v = tree ]; then unset EFFORT_HIP_LIB; else export EFFORT_HIP_LIB=build/variants/$v.so; fi
  timeout 200 python tools/qbench.py --group 32 --reps 2 --tag big1-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag big4-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 32 --reps 2 --overlap 2 --steps-per-graph 8 --tag big2-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --shape 4096x4096 --group 32 --reps 2 --tag sq32-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 32 --effort 0.5 --reps 2 --tag big1e50-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
done
done
cat $O/ab.txt
