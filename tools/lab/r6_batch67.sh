#!/bin/bash
# round 6 (third session): Q4 lone calls, pairs, triples over shapes (beside FP16's of r6_batch60)
export TMPDIR=/tmp
O=gpurun_out/b67; mkdir -p $O; rm -f $O/scan.txt
for shape in 4096x4096 4096x1024 4096x11008 11008x4096 4096x14336 14336x4096 8192x8192 8192x1024 5120x5120 5120x13824 13824x5120; do
  timeout 300 python tools/lab/nscan.py --q4 1 --shape $shape --ns 1,2,3,4 --mats 24 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
cat $O/scan.txt
