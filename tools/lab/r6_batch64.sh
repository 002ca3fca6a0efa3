#!/bin/bash
# round 6 (third session): Q4 launches just past one round of two workgroups per CU (persistent grids): thin last calls by the lab knob
export TMPDIR=/tmp EFFORT_HIP_LIB=lab EFFORT_TAIL_MULT=2
O=gpurun_out/b64; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --q4 1 --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 22 --mats 22 --tails 0,2,4 --tag q22
q --group 24 --mats 24 --tails 0,2,4,8 --tag q24
q --group 26 --mats 26 --tails 0,2,4 --tag q26
q --group 17 --mats 17 --shape 4096x14336 --tails 0,2,4 --tag w1q17
q --group 18 --mats 18 --shape 4096x14336 --tails 0,2,4 --tag w1q18
q --group 17 --mats 17 --shape 14336x4096 --tails 0,2,4 --tag w2q17
q --group 24 --mats 24 --effort 0.5 --tails 0,2,4 --tag q24e50
cat $O/sweep.txt
