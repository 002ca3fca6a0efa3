#!/bin/bash
# round 6 (second session): the shipped form of the row stream's cache policy -- nt by default, effort_set_row_reuse(1) = temporal, one kernel with a
# uniform branch between two batches of loads -- against build/variants/aux0.so (compile-time temporal = the library before) and nt.so (compile-time nt)
export TMPDIR=/tmp
O=gpurun_out/b19; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in aux0 tree tree-reuse; do
  R=0
  case $v in aux0|nt) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; tree) unset EFFORT_HIP_LIB;; tree-reuse) unset EFFORT_HIP_LIB; R=1;; esac
  q --group 32 --row-reuse $R --tag g32-$v
  q --mats 128 --group 32 --overlap 4 --steps-per-graph 8 --row-reuse $R --tag g32x4disjoint-$v
  q --group 32 --overlap 4 --steps-per-graph 8 --row-reuse $R --tag g32x4shared-$v
  q --group 1 --row-reuse $R --tag lone-$v
  q --group 3 --row-reuse $R --tag three-$v
  q --group 16 --q4 1 --row-reuse $R --tag q4x16-$v
done
done
cat $O/ab.txt
