#!/bin/bash
# round 6, second batch: the decode loop's launches over 8-wave geometries with round 5's kernels; Q4 per-item timeline
export TMPDIR=/tmp
O=gpurun_out/b2; mkdir -p $O
CFG="0,0,0;8,1,16;8,1,24;8,1,32;8,1,48;8,1,56;8,1,64;8,2,16;8,2,24;8,2,28;8,2,32;8,2,48;8,2,56;8,2,64;8,4,16;8,4,32;8,4,64"
for e in 0.25 0.5; do
for L in 4096x4096 "4096x4096,4096x1024,4096x1024" "4096x14336,4096x14336" 14336x4096 4096x14336 4096x11008; do
  timeout 300 python tools/lab/geosweep.py --launch $L --effort $e --configs "$CFG" >> $O/geosweep.txt 2>&1
done
done
cat $O/geosweep.txt
timeout 300 python tools/timeline.py --q4 1 --groups 16 --replay 10 --out $O/timeline_q4_16.json > $O/timeline_q4_16.txt 2>&1; cat $O/timeline_q4_16.txt | head -80
