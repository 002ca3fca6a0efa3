#!/bin/bash
# round 6: a CU's second persistent Q4 workgroup starts D us late (build/variants/staggerD.so) -- does taking the pair out of step pay?
export TMPDIR=/tmp
O=gpurun_out/b7; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do
for v in tree stagger6 stagger12 stagger18; do
  if [ $v = tree ]; then unset EFFORT_HIP_LIB; else export EFFORT_HIP_LIB=build/variants/$v.so; fi
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --configs "8,2,16:-1;8,1,8:-1;8,2,8:1" --tag q4x16-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --configs "0,0,0:-1;8,2,16:-1" --tag q4x32-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag q4x32x4-$v 2>&1 | grep "rep 1" | cut -c1-100 >> $O/ab.txt
done
done
cat $O/ab.txt
