#!/bin/bash
# round 6 (third session): Q4 groups that do not fit one E = 1 item per CU: one E = 2 item per CU by hand
export TMPDIR=/tmp
O=gpurun_out/b53; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --q4 1 --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 9 --mats 36 --shape 4096x14336 --configs "0,0,0:-1;8,2,6:-1;8,2,7:-1;8,1,5:-1" --tag w1q9
q --group 7 --mats 35 --shape 14336x4096 --configs "0,0,0:-1;8,2,32:-1;8,2,28:-1;8,2,36:-1;8,2,24:-1" --tag w2q7
q --group 9 --mats 36 --shape 14336x4096 --configs "0,0,0:-1;8,2,24:-1;8,2,28:-1;8,2,20:-1" --tag w2q9
q --group 9 --mats 36 --configs "0,0,0:-1;8,1,8:-1;8,2,7:-1" --tag q9
q --group 7 --mats 35 --shape 4096x14336 --configs "0,0,0:-1;8,2,8:-1;8,2,9:-1;8,1,5:-1" --tag w1q7
q --group 8 --mats 32 --shape 4096x14336 --configs "0,0,0:-1;8,2,8:-1;8,2,7:-1" --tag w1q8
cat $O/sweep.txt
