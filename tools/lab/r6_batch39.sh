#!/bin/bash
# round 6 (third session): 4096 x 4096 matrices, 16 / 32 calls per launch (0.46 / 0.53 of the roofline): geometry re-sweep under nt + plain grids
export TMPDIR=/tmp
O=gpurun_out/b39; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-100 >> $O/sweep.txt; }
C="0,0,0:-1;8,2,16:-1;8,2,32:-1;8,1,8:-1;8,1,16:-1;8,4,8:-1;8,4,16:-1;8,4,32:-1;8,2,8:2;8,2,16:2"
q --group 16 --mats 32 --shape 4096x4096 --configs "$C" --tag sq16
q --group 32 --mats 64 --shape 4096x4096 --configs "$C" --tag sq32
q --group 16 --mats 32 --shape 4096x4096 --effort 0.5 --configs "$C" --tag sq16e50
q --group 8 --mats 32 --shape 4096x4096 --configs "$C" --tag sq8
cat $O/sweep.txt
