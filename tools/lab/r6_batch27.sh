#!/bin/bash
# round 6 (second session): finer items on PLAIN grids (the dispatcher balances; every workgroup pays its own cutoff), 32 and 16 calls per launch
export TMPDIR=/tmp
O=gpurun_out/b27; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/sweep.txt; }
q --group 32 --configs "0,0,0:-1;8,4,8:0;8,4,12:0;8,4,16:0;8,2,8:0;8,4,10:0;8,4,9:0" --tag g32
q --group 16 --configs "0,0,0:-1;8,2,8:0;8,2,12:0;8,2,16:0;8,4,16:0;8,4,12:0;8,2,10:0" --tag g16
q --group 32 --shape 4096x14336 --mats 32 --configs "0,0,0:-1;0,0,0:0;8,4,8:0;8,4,6:0;8,4,6:2" --tag w1x32
q --group 32 --q4 1 --configs "0,0,0:-1;0,0,0:0" --tag q4x32
q --group 24 --q4 1 --mats 24 --configs "0,0,0:-1;0,0,0:0" --tag q4x24
q --group 8 --configs "0,0,0:-1;8,2,16:0;8,2,12:0;8,4,8:0;8,1,8:0" --tag g8
cat $O/sweep.txt
