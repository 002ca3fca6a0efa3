#!/bin/bash
# round 6 (third session): Q4, four launches in flight, 1 .. 7 calls per launch: E = 1 (the rule for < 8 calls) against E = 2
export TMPDIR=/tmp
O=gpurun_out/b59; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --q4 1 --reps 3 --overlap 4 --steps-per-graph 4 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 1 2 3 4 6 7; do q --group $n --mats $((n*12)) --configs "0,0,0:-1;8,2,0:-1" --tag LQn$n; done
for n in 2 3 4 6; do q --group $n --mats $((n*12)) --shape 4096x4096 --configs "0,0,0:-1;8,2,0:-1" --tag LQsq$n; done
for n in 3 6; do q --group $n --mats $((n*12)) --shape 14336x4096 --configs "0,0,0:-1;8,2,0:-1" --tag LQw2n$n; done
for n in 3 6; do q --group $n --mats $((n*12)) --effort 0.5 --configs "0,0,0:-1;8,2,0:-1" --tag LQe50n$n; done
cat $O/sweep.txt
