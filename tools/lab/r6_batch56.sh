#!/bin/bash
# round 6 (third session): four launches in flight, narrow matrices (<= 256 bucket columns), 1 .. 7 calls per launch: the 64-column tiles of the lone-call rule against E = 2 / E = 4
export TMPDIR=/tmp
O=gpurun_out/b56; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 --overlap 4 --steps-per-graph 4 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 1 2 3 4 6; do q --group $n --mats $((n*16)) --shape 14336x4096 --configs "0,0,0:-1;8,2,0:-1;8,4,0:-1" --tag Lw2n$n; done
for n in 1 2 3 4 6; do q --group $n --mats $((n*16)) --shape 4096x4096 --configs "0,0,0:-1;8,2,0:-1;8,4,0:-1" --tag Lsqn$n; done
for n in 3 6; do q --group $n --mats $((n*8)) --shape 4096x11008 --configs "0,0,0:-1;8,4,0:-1" --tag Ln$n; done
cat $O/sweep.txt
