#!/bin/bash
# round 6 (third session): pairs and lone calls with slice counts between the powers of two (one item per CU on the padded ranges)
export TMPDIR=/tmp
O=gpurun_out/b52; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 2 --mats 32 --configs "0,0,0:-1;8,2,18:-1;8,2,20:-1;8,2,21:-1;8,2,24:-1" --tag p11008
q --group 2 --mats 32 --shape 4096x14336 --configs "0,0,0:-1;8,2,17:-1;8,2,18:-1;8,2,20:-1" --tag p14336
q --group 2 --mats 32 --effort 0.5 --configs "0,0,0:-1;8,2,20:-1;8,2,21:-1" --tag p11008e50
q --group 2 --mats 32 --shape 4096x14336 --effort 0.5 --configs "0,0,0:-1;8,2,18:-1" --tag p14336e50
q --group 1 --mats 32 --configs "0,0,0:-1;8,2,36:-1;8,2,40:-1;8,2,42:-1" --tag l11008
q --group 1 --mats 32 --effort 0.5 --configs "0,0,0:-1;8,2,36:-1;8,2,40:-1;8,2,42:-1" --tag l11008e50
q --group 1 --mats 32 --effort 1.0 --configs "0,0,0:-1;8,2,40:-1;8,2,42:-1" --tag l11008e100
q --group 1 --mats 32 --shape 4096x14336 --effort 0.5 --configs "0,0,0:-1;8,2,36:-1" --tag l14336e50
q --group 2 --mats 32 --shape 14336x4096 --configs "0,0,0:-1;8,1,28:-1;8,1,24:-1" --tag p14336x4096
cat $O/sweep.txt
