#!/bin/bash
# round 6 (third session): 3..10 calls of a BIG matrix per launch: slice counts that fill one round of CUs exactly (any count, not only powers of two)
export TMPDIR=/tmp
O=gpurun_out/b43; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 3 --mats 33 --configs "0,0,0:-1;8,2,10:-1;8,2,12:-1;8,2,14:-1;8,2,16:-1;8,4,24:-1;8,4,28:-1;8,4,16:-1" --tag n3
q --group 4 --mats 32 --configs "0,0,0:-1;8,2,10:-1;8,2,12:-1;8,2,16:-1;8,4,20:-1;8,4,16:-1" --tag n4
q --group 5 --mats 35 --configs "0,0,0:-1;8,2,10:-1;8,4,16:-1;8,4,17:-1" --tag n5
q --group 6 --mats 36 --configs "0,0,0:-1;8,4,8:-1;8,4,12:-1;8,4,14:-1;8,4,16:-1;8,2,16:-1" --tag n6
q --group 7 --mats 35 --configs "0,0,0:-1;8,4,8:-1;8,4,10:-1;8,4,12:-1;8,4,16:-1" --tag n7
q --group 8 --mats 32 --configs "0,0,0:-1;8,4,9:-1;8,4,10:-1;8,4,16:-1;8,2,8:-1" --tag n8
q --group 10 --mats 30 --configs "0,0,0:-1;8,4,16:-1;8,2,8:-1" --tag n10
q --group 3 --mats 33 --effort 0.5 --configs "0,0,0:-1;8,2,14:-1;8,4,28:-1" --tag n3e50
q --group 6 --mats 36 --effort 0.5 --configs "0,0,0:-1;8,4,14:-1" --tag n6e50
q --group 8 --mats 32 --effort 0.5 --configs "0,0,0:-1;8,4,10:-1" --tag n8e50
cat $O/sweep.txt
