#!/bin/bash
# round 6 (third session): where does the thin tail stop paying?  last-round fills between an eighth and a quarter (lab library, EFFORT_TAIL_CALLS 0 / 2)
export TMPDIR=/tmp EFFORT_HIP_LIB=lab EFFORT_TAIL_MULT=2
O=gpurun_out/b35; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 23 --mats 23 --tails 0,2,0,2 --tag n23-r80
q --group 13 --mats 13 --tails 0,2,0,2 --tag n13-r112
q --group 20 --mats 20 --shape 4096x14336 --tails 0,2,0,2 --tag w1n20-r96
q --group 11 --mats 11 --shape 4096x14336 --tails 0,2,0,2 --tag w1n11-r104
q --group 18 --mats 18 --shape 14336x4096 --tails 0,2,0,2 --tag w2n18-r128
q --group 23 --mats 23 --effort 0.5 --tails 0,2,0,2 --tag n23e50-r80
q --group 13 --mats 13 --effort 0.5 --tails 0,2,0,2 --tag n13e50-r112
cat $O/sweep.txt
