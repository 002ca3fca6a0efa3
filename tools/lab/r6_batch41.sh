#!/bin/bash
# round 6 (third session): 9..15 calls on small matrices: 8 slices (under one item per CU) against 16
export TMPDIR=/tmp
O=gpurun_out/b41; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-100 >> $O/sweep.txt; }
for n in 9 10 11 12 13 14 15; do q --group $n --mats $((n*3)) --shape 4096x4096 --configs "0,0,0:-1;8,2,16:-1" --tag sq$n; done
for n in 9 10 12 14; do q --group $n --mats $((n*3)) --shape 4096x4096 --effort 0.5 --configs "0,0,0:-1;8,2,16:-1" --tag sq${n}e50; done
for n in 10 12 14 20 24; do q --group $n --mats $((n*2)) --shape 4096x1024 --configs "0,0,0:-1;8,1,16:-1;8,1,32:-1" --tag kv$n; done
for n in 10 12 14; do q --group $n --mats $((n*2)) --shape 4096x2048 --configs "0,0,0:-1;8,1,8:-1;8,1,16:-1;8,2,16:-1;8,2,32:-1" --tag h$n; done
cat $O/sweep.txt
