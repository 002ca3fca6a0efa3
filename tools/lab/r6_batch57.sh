#!/bin/bash
# round 6 (third session): WITHOUT lanes, narrow matrices, 2 .. 7 calls per launch: the 64-column tiles (E = 1) against E = 2 (the heuristic's slices for that E)
export TMPDIR=/tmp
O=gpurun_out/b57; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 2 3 4 5 6 7; do q --group $n --mats $((n*8)) --shape 14336x4096 --configs "0,0,0:-1;8,2,0:-1" --tag w2n$n; done
for n in 2 3 4 5 6 7; do q --group $n --mats $((n*12)) --shape 4096x4096 --configs "0,0,0:-1;8,2,0:-1" --tag sqn$n; done
cat $O/sweep.txt
