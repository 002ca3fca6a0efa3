#!/bin/bash
# round 6 (third session): which launches gain from thinner slices on their LAST calls?  n calls per launch x the last tc at twice the slices, one launch in flight
export TMPDIR=/tmp EFFORT_HIP_LIB=lab EFFORT_TAIL_MULT=2
O=gpurun_out/b32; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-130 >> $O/sweep.txt; }
q --group 9  --mats 9  --tails 0,3,6,9 --tag n9
q --group 10 --mats 10 --tails 0,2,4,6,8,10 --tag n10
q --group 12 --mats 12 --tails 0,2,4,6,8,10,12 --tag n12
q --group 14 --mats 14 --tails 0,2,4,6,8 --tag n14
q --group 16 --mats 16 --tails 0,2,3,4,5,6 --tag n16
q --group 18 --mats 18 --tails 0,1,2,3,4 --tag n18
q --group 20 --mats 20 --tails 0,1,2 --tag n20
q --group 22 --mats 22 --tails 0,2,6,8,10 --tag n22
q --group 24 --mats 24 --tails 0,4,6,7,8,10 --tag n24
q --group 28 --mats 28 --tails 0,2,4,8 --tag n28
q --group 8  --mats 8  --tails 0,2,4,8 --tag n8
# other shapes / efforts at 16 and 12
q --group 16 --mats 16 --effort 0.5 --tails 0,2,4,6 --tag n16e50
q --group 16 --mats 16 --effort 0.1 --tails 0,2,4,6 --tag n16e10
q --group 12 --mats 12 --effort 0.5 --tails 0,4,8 --tag n12e50
q --group 16 --mats 16 --shape 4096x14336 --tails 0,2,4,6,8 --tag w1n16
q --group 16 --mats 16 --shape 14336x4096 --tails 0,2,4,8,16 --tag w2n16
q --group 16 --mats 32 --shape 4096x4096 --tails 0,4,8,16 --tag sqn16
q --group 32 --mats 64 --shape 4096x4096 --tails 0,8,16,32 --tag sqn32
cat $O/sweep.txt
