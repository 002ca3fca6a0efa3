#!/bin/bash
# round 6 (third session): Q4 LONE calls on tall narrow matrices (w2: 14336 -> 4096, 11008 -> 4096): a lone call slower than a pair -- slice counts by hand
export TMPDIR=/tmp
O=gpurun_out/b68; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --q4 1 --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 1 --mats 24 --shape 14336x4096 --configs "0,0,0:-1;8,1,32:-1;8,1,48:-1;8,1,64:-1;8,1,80:-1;8,2,64:-1;8,2,96:-1" --tag w2q1
q --group 1 --mats 24 --shape 11008x4096 --configs "0,0,0:-1;8,1,32:-1;8,1,44:-1;8,1,64:-1;8,2,64:-1;8,2,88:-1" --tag w2bq1
q --group 1 --mats 24 --shape 14336x4096 --effort 0.5 --configs "0,0,0:-1;8,1,48:-1;8,1,64:-1" --tag w2q1e50
q --group 2 --mats 24 --shape 14336x4096 --configs "0,0,0:-1;8,1,32:-1;8,1,24:-1;8,1,64:-1" --tag w2q2
q --group 1 --mats 24 --shape 4096x4096 --configs "0,0,0:-1;8,1,16:-1;8,1,64:-1;8,2,32:-1" --tag sqq1
cat $O/sweep.txt
