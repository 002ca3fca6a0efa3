#!/bin/bash
# round 6 (second session): nt on the bucket-row stream now that the bench's rows sit at the line-aligned 1408-byte pitch (the earlier nt
# measurement, worse, was at the 1376-byte pitch where neighbouring tiles share the lines a row straddles).  build/variants/nt.so = -DEFFORT_ROW_AUX=2
export TMPDIR=/tmp
O=gpurun_out/b16; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 400 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in base nt; do
  if [ $v = nt ]; then export EFFORT_HIP_LIB=build/variants/nt.so; else unset EFFORT_HIP_LIB; fi
  q --group 32 --tag g32-$v
  q --group 32 --overlap 4 --steps-per-graph 8 --tag g32x4-$v
  q --group 16 --tag g16-$v
  q --group 1 --tag lone-$v
  q --group 32 --effort 0.5 --tag g32e50-$v
done
done
cat $O/ab.txt
