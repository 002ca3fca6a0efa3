#!/bin/bash
# round 6 (second session): the row stream's cache policy bits (aux: 1 sc0, 2 nt, 16 sc1) swept; and nt on rows that are NOT line-aligned (1376-byte pitch)
export TMPDIR=/tmp
O=gpurun_out/b18; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in base nt aux3 aux18 aux19 aux16; do
  if [ $v = base ]; then unset EFFORT_HIP_LIB; else export EFFORT_HIP_LIB=$PWD/build/variants/$v.so; fi
  q --group 32 --tag g32-$v
  q --mats 128 --group 32 --overlap 4 --steps-per-graph 8 --tag g32x4disjoint-$v
  q --group 1 --tag lone-$v
done
done
for v in base nt base nt; do
  if [ $v = base ]; then unset EFFORT_HIP_LIB; else export EFFORT_HIP_LIB=$PWD/build/variants/$v.so; fi
  q --group 32 --no-align 1 --tag g32-unaligned-$v
  q --mats 128 --group 32 --overlap 4 --steps-per-graph 8 --no-align 1 --tag g32x4disjoint-unaligned-$v
  q --group 1 --no-align 1 --tag lone-unaligned-$v
  q --group 32 --shape 4096x4096 --tag sq32-$v
  q --group 1 --shape 4096x14336 --tag lone14336-$v
  q --group 1 --shape 14336x4096 --tag lone14336x4096-$v
done
cat $O/ab.txt
