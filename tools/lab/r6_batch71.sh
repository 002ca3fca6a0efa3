#!/bin/bash
# round 6 (third session): the column shards of a world of 8: lone calls / pairs / triples on the shard shapes -- 11008 / 8 = 1376, 14336 / 8 = 1792, 4096 / 8 = 512, 1024 / 8 = 128 outputs
export TMPDIR=/tmp
O=gpurun_out/b71; mkdir -p $O; rm -f $O/scan.txt
for shape in 4096x1376 4096x1792 4096x512 4096x128 14336x512 11008x512; do
  timeout 300 python tools/lab/nscan.py --shape $shape --ns 1,2,3,4,6,8 --mats 48 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
timeout 300 python tools/lab/nscan.py --mix 4096x512,4096x128,4096x128 --ns 3,6,12 --mats 48 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
cat $O/scan.txt
