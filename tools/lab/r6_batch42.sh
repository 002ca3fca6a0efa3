#!/bin/bash
# round 6 (third session): scan n = 1 .. 32 calls per launch x shapes for holes in the geometry heuristics (time per launch should grow smoothly with n)
export TMPDIR=/tmp
O=gpurun_out/b42; mkdir -p $O; rm -f $O/scan.txt
for shape in 4096x1024 4096x2048 4096x4096 4096x11008 4096x14336 14336x4096 11008x4096; do
  timeout 600 python tools/lab/nscan.py --shape $shape 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
timeout 600 python tools/lab/nscan.py --shape 4096x4096 --effort 0.5 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
timeout 600 python tools/lab/nscan.py --shape 4096x11008 --effort 0.5 --ns 1,2,3,4,5,6,7,8,9,10,11,12 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
cat $O/scan.txt
