#!/bin/bash
# round 6 (second session): E = 2 against E = 4 tiles on plain grids from 20 calls per launch on (one box)
export TMPDIR=/tmp
O=gpurun_out/b28; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/sweep.txt; }
q --group 32 --configs "0,0,0:-1;8,4,8:0;8,2,8:0;8,4,8:2;8,2,8:2;8,4,8:0;8,2,8:0" --tag g32
q --group 32 --effort 0.5 --configs "0,0,0:-1;8,4,8:0;8,2,8:0" --tag g32e50
q --group 32 --effort 0.1 --configs "0,0,0:-1;8,4,8:0;8,2,8:0;8,4,8:2" --tag g32e10
q --group 32 --effort 1.0 --configs "0,0,0:-1;8,4,8:0;8,2,8:0" --tag g32e100
q --group 24 --mats 24 --configs "0,0,0:-1;8,4,8:0;8,2,8:0" --tag g24
q --group 20 --mats 20 --configs "0,0,0:-1;8,4,8:0;8,2,8:0" --tag g20
q --group 32 --shape 4096x14336 --mats 32 --configs "0,0,0:-1;8,4,8:0;8,2,8:0" --tag w1x32
q --group 32 --shape 14336x4096 --mats 32 --configs "0,0,0:-1;0,0,0:0;0,0,0:2" --tag w2x32
q --group 32 --shape 4096x4096 --mats 64 --configs "0,0,0:-1;8,4,8:0;8,2,8:0;8,4,16:0" --tag sq32
cat $O/sweep.txt
