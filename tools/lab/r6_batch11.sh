#!/bin/bash
# round 6: ONE full round of workgroups -- the slices chosen so that calls x tiles x slices lands just under 2 x numCU items (480-504 of 512)
export TMPDIR=/tmp
O=gpurun_out/b11; mkdir -p $O; rm -f $O/sweep.txt
timeout 400 python tools/qbench.py --q4 1 --group 16 --reps 2 --configs "0,0,0:-1;8,2,10:0;8,2,9:0;8,2,11:0;8,1,5:0;8,2,10:-1" --tag q4x16 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --q4 1 --group 16 --reps 2 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,2,10:0" --tag q4x16x4 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --q4 1 --group 8 --reps 2 --configs "0,0,0:-1;8,2,21:0;8,2,20:0;8,2,16:0;8,1,10:0" --tag q4x8 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --group 16 --reps 2 --configs "0,0,0:-1;8,4,10:0;8,4,9:0;8,2,5:0;8,4,11:0" --tag fp16x16 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --group 8 --reps 2 --configs "0,0,0:-1;8,4,21:0;8,4,16:0;8,2,10:0;8,4,20:0" --tag fp16x8 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --group 16 --reps 2 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,4,10:0" --tag fp16x16x4 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
timeout 400 python tools/qbench.py --shape 4096x4096 --group 16 --reps 2 --configs "0,0,0:-1;8,4,32:0;8,2,16:0;8,2,15:0;8,4,30:0" --tag sq16 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt
cat $O/sweep.txt
