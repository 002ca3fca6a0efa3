#!/bin/bash
# round 6 (third session): where should E = 4 begin?  launches of 640..760 E = 4 items (today: E = 4 from 768 = three per CU on)
export TMPDIR=/tmp
O=gpurun_out/b38; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 27 28 29; do q --group $n --mats $n --configs "0,0,0:-1;8,4,0:-1" --tag n$n; done
for n in 20 21 22 23; do q --group $n --mats $n --shape 4096x14336 --configs "0,0,0:-1;8,4,0:-1" --tag w1n$n; done
for n in 20 21 22 23; do q --group $n --mats $n --shape 14336x4096 --configs "0,0,0:-1;8,4,0:-1" --tag w2n$n; done
q --group 28 --mats 28 --effort 1.0 --configs "0,0,0:-1;8,4,0:-1" --tag n28e100
q --group 22 --mats 22 --shape 4096x14336 --effort 0.5 --configs "0,0,0:-1;8,4,0:-1" --tag w1n22e50
cat $O/sweep.txt
