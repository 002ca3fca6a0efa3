#!/bin/bash
# round 6 (third session): the n-scan with FOUR launches in flight (effort_set_overlap(4)): independent n-call launches, us per launch = replay time / launches
export TMPDIR=/tmp
O=gpurun_out/b55; mkdir -p $O; rm -f $O/scan.txt
for shape in 4096x11008 4096x14336 14336x4096 4096x4096; do
  timeout 600 python tools/lab/nscan.py --overlap 4 --mats 96 --shape $shape --ns 1,2,3,4,5,6,7,8,9,10,11,12,14,16,20,24,28,32 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
timeout 600 python tools/lab/nscan.py --overlap 4 --mats 96 --q4 1 --shape 4096x11008 --ns 1,2,3,4,5,6,7,8,9,10,11,12,14,16,20,24,28,32 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
cat $O/scan.txt
