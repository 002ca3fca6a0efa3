#!/bin/bash
# round 6 (second session): nt on the Q4 outlier entries (build/variants/olt.so = rows nt, entries temporal) and on the dense baseline's weight
# stream (build/variants/gemv0.so = temporal); the tree = nt everywhere
export TMPDIR=/tmp
O=gpurun_out/b20; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in olt tree; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  q --group 16 --q4 1 --tag q4x16-$v
  q --group 32 --q4 1 --tag q4x32-$v
  q --mats 64 --group 16 --q4 1 --overlap 4 --steps-per-graph 8 --tag q4x16x4-$v
  q --group 1 --q4 1 --tag q4lone-$v
done
for v in gemv0 tree; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  timeout 300 python tools/lab/densebench.py --tag $v 2>&1 | grep -E "rep 2|err" >> $O/ab.txt
  timeout 300 python tools/lab/densebench.py --shape 4096x4096 --tag $v 2>&1 | grep -E "rep 2|err" >> $O/ab.txt
done
done
unset EFFORT_HIP_LIB
q --group 32 --tag g32-tree
q --group 1 --tag lone-tree
cat $O/ab.txt
