#!/bin/bash
# round 6 (third session): E = 4 against E = 2 for 20..32 calls per launch under plain grids + nt (the E rule dates from persistent grids)
export TMPDIR=/tmp
O=gpurun_out/b37; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 16 20 22 24 26 28 30 31 32; do q --group $n --mats $n --configs "0,0,0:-1;8,4,0:-1;8,2,0:-1" --tag n$n; done
for n in 24 28 32; do q --group $n --mats $n --effort 0.5 --configs "0,0,0:-1;8,4,0:-1;8,2,0:-1" --tag n${n}e50; done
for n in 24 28 32; do q --group $n --mats $n --effort 0.1 --configs "0,0,0:-1;8,4,0:-1;8,2,0:-1" --tag n${n}e10; done
cat $O/sweep.txt
