#!/bin/bash
# round 6 (second session): 16 rows per batch at E = 4 (32 row pieces in flight per lane instead of 16: build/variants/k16.so = -DEFFORT_KBATCH4=16, 115 VGPRs, no
# scratch) now that the row stream is nt -- the per-item trace shows a workgroup left alone on its CU pulling 17 GB/s where the pair pulled 27
export TMPDIR=/tmp
O=gpurun_out/b22; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in tree k16; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  q --group 32 --tag g32-$v
  q --mats 128 --group 32 --overlap 4 --steps-per-graph 8 --tag g32x4disjoint-$v
  q --group 16 --tag g16-$v
  q --group 32 --effort 0.5 --tag g32e50-$v
  q --group 8 --tag g8-$v
done
done
cat $O/ab.txt
