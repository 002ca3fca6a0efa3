#!/bin/bash
# round 6 (second session): 32 rows per batch for the 2- and 4-byte pieces (E = 1, 2: lone calls and small groups; build/variants/kb32.so = -DEFFORT_KBATCH=32).
# A lone call's wave has ~58 rows in all: two batches of 16 in flight make its stream three or four dependent memory round trips; two batches of 32 = one.
export TMPDIR=/tmp
O=gpurun_out/b26; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in tree kb32; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  q --group 1 --tag lone-$v
  q --group 1 --effort 0.5 --tag lone50-$v
  q --group 1 --effort 1.0 --tag lone100-$v
  q --group 1 --shape 4096x14336 --tag lone14336-$v
  q --group 1 --shape 14336x4096 --tag lone14336x4096-$v
  q --group 1 --shape 4096x4096 --tag lonesq-$v
  q --group 3 --tag three-$v
  q --group 8 --tag g8-$v
  q --group 16 --tag g16-$v
  q --group 1 --q4 1 --tag q4lone-$v
  q --group 16 --q4 1 --tag q4x16-$v
  echo "layer_probe $v" >> $O/ab.txt; timeout 300 python tools/layer_probe.py 2>&1 | tail -1 | cut -c1-300 >> $O/ab.txt
done
done
cat $O/ab.txt
