#!/bin/bash
# round 6 (third session): Q4 groups of 3 .. 9 calls: one round of narrow, tall items (the rule of the 10 .. 16-call groups) by hand
export TMPDIR=/tmp
O=gpurun_out/b46; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --q4 1 --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 3 --mats 33 --configs "0,0,0:-1;8,1,10:-1;8,1,12:-1;8,1,14:-1;8,1,16:-1" --tag q3
q --group 4 --mats 32 --configs "0,0,0:-1;8,1,9:-1;8,1,10:-1;8,1,11:-1" --tag q4
q --group 5 --mats 35 --configs "0,0,0:-1;8,1,7:-1;8,1,9:-1" --tag q5
q --group 6 --mats 36 --configs "0,0,0:-1;8,1,6:-1;8,1,7:-1" --tag q6
q --group 7 --mats 35 --configs "0,0,0:-1;8,1,5:-1;8,1,6:-1" --tag q7
q --group 8 --mats 32 --configs "0,0,0:-1;8,1,5:-1;8,1,8:-1;8,2,10:-1" --tag q8
q --group 9 --mats 36 --configs "0,0,0:-1;8,1,5:-1;8,2,8:-1" --tag q9
q --group 8 --mats 32 --shape 4096x4096 --configs "0,0,0:-1;8,1,8:-1;8,1,12:-1;8,1,15:-1;8,1,16:-1;8,2,16:-1;8,2,30:-1" --tag sq8
q --group 9 --mats 36 --shape 4096x4096 --configs "0,0,0:-1;8,1,13:-1;8,1,12:-1;8,2,16:-1" --tag sq9
q --group 5 --mats 35 --shape 14336x4096 --configs "0,0,0:-1;8,1,24:-1;8,1,22:-1;8,1,18:-1" --tag w2q5
q --group 9 --mats 36 --shape 4096x14336 --configs "0,0,0:-1;8,1,5:-1;8,2,8:-1" --tag w1q9
cat $O/sweep.txt
