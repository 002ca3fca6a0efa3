#!/bin/bash
# round 6 (third session): the n-scan for Q4 (bucketMulQ4 groups), and FP16 n = 12 .. 32 on the remaining shapes
export TMPDIR=/tmp
O=gpurun_out/b45; mkdir -p $O; rm -f $O/scan.txt
for shape in 4096x11008 4096x14336 14336x4096 4096x4096; do
  timeout 600 python tools/lab/nscan.py --q4 1 --shape $shape --ns 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,20,24,28,32 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --effort 0.5 --ns 1,2,3,4,5,6,7,8,9,10,12,16,20,24,32 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
cat $O/scan.txt
