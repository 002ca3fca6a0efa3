#!/bin/bash
# round 6 (third session): MIXED groups -- k sequences' Wq | Wk | Wv (4096x4096, 4096x1024, 4096x1024) and W1 | W3 pairs in one launch, with and without lanes
export TMPDIR=/tmp
O=gpurun_out/b66; mkdir -p $O; rm -f $O/scan.txt
timeout 600 python tools/lab/nscan.py --mix 4096x4096,4096x1024,4096x1024 --ns 3,6,9,12,15,18,24,30 --mats 60 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
timeout 600 python tools/lab/nscan.py --mix 4096x4096,4096x1024,4096x1024 --ns 3,6,9,12,15,18,24,30 --mats 60 --overlap 4 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
timeout 600 python tools/lab/nscan.py --mix 4096x14336,4096x14336 --ns 2,4,6,8,10,12,16 --mats 48 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
timeout 600 python tools/lab/nscan.py --mix 4096x4096,4096x1024,4096x1024 --ns 3,6,9,12 --mats 60 --effort 0.5 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
cat $O/scan.txt
