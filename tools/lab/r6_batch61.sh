#!/bin/bash
# round 6 (third session): triples (and 4 .. 6 calls) of NARROW tall matrices without lanes: the 64-column tiles overflow one round at the fewest slices; 128-column tiles fit
export TMPDIR=/tmp
O=gpurun_out/b61; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 3 --mats 24 --shape 11008x4096 --configs "0,0,0:-1;8,2,24:-1;8,2,32:-1;8,2,40:-1" --tag w2b3
q --group 4 --mats 24 --shape 11008x4096 --configs "0,0,0:-1;8,2,24:-1;8,2,32:-1" --tag w2b4
q --group 5 --mats 25 --shape 11008x4096 --configs "0,0,0:-1;8,2,24:-1" --tag w2b5
q --group 3 --mats 24 --shape 14336x4096 --configs "0,0,0:-1;8,2,32:-1;8,2,40:-1" --tag w2n3
q --group 4 --mats 24 --shape 14336x4096 --configs "0,0,0:-1;8,2,32:-1" --tag w2n4
q --group 3 --mats 24 --shape 13824x5120 --configs "0,0,0:-1;8,4,32:-1;8,4,40:-1;8,2,28:-1" --tag l13w2n3
q --group 3 --mats 24 --shape 11008x4096 --effort 0.5 --configs "0,0,0:-1;8,2,40:-1" --tag w2b3e50
q --group 3 --mats 24 --shape 14336x4096 --effort 0.5 --configs "0,0,0:-1;8,2,40:-1" --tag w2n3e50
cat $O/sweep.txt
