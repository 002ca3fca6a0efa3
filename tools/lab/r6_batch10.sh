#!/bin/bash
# round 6: FP16 groups of 8 / 16 / 24 calls per launch: the heuristic's geometry against balanced / persistent ones (one launch in flight and four)
export TMPDIR=/tmp
O=gpurun_out/b10; mkdir -p $O; rm -f $O/sweep.txt
CFG="0,0,0:-1;8,4,16:-1;8,4,12:-1;8,2,8:-1;8,2,16:-1;8,2,12:-1;8,4,8:1;8,2,8:0;8,4,16:0"
for g in 8 16 24; do
  timeout 400 python tools/qbench.py --group $g --reps 2 --configs "$CFG" --tag fp16x$g 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
done
timeout 400 python tools/qbench.py --group 16 --reps 2 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,4,16:-1;8,2,8:-1" --tag fp16x16x4 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --shape 4096x4096 --group 16 --reps 2 --configs "0,0,0:-1;8,2,16:-1;8,1,16:-1;8,2,32:-1;8,4,16:-1;8,1,32:-1" --tag sq16 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --shape 4096x4096 --group 32 --reps 2 --configs "0,0,0:-1;8,2,16:-1;8,1,16:-1;8,2,32:-1;8,4,16:-1;8,4,32:-1" --tag sq32 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
cat $O/sweep.txt
