#!/bin/bash
# bucketMulQ4 4096x11008 at 25 %, 16 calls per launch: launch geometries (waves,elems,slices:workgroups-per-CU; -1 = heuristic grid,
# 0 = plain grid, R = R persistent workgroups per CU) -- how many items, of which size, on how many resident workgroups.
export TMPDIR=/tmp
O=gpurun_out/q4geo; mkdir -p $O; rm -f $O/sweep.txt
CFG="0,0,0:-1;8,2,8:2;8,2,16:-1;8,2,16:0;8,2,12:0;8,2,24:0;8,2,32:-1;8,1,8:-1;8,1,8:0;8,1,16:-1;8,1,16:0;8,4,8:0;8,4,16:0;8,4,16:-1;16,2,8:0;16,4,8:0;16,4,16:0;4,1,16:0;4,1,16:-1;4,2,16:0;4,2,8:0;4,4,8:0"
timeout 900 python tools/qbench.py --q4 1 --group ${GROUP:-16} --reps 2 --configs "$CFG" --tag q4geo 2>&1 | grep -E "rep|Error|error" >> $O/sweep.txt
timeout 600 python tools/qbench.py --q4 1 --group ${GROUP:-16} --reps 1 --no-outliers 1 --configs "0,0,0:-1;8,2,16:-1;8,1,8:-1;8,1,16:0;4,1,16:0;8,4,16:0" --tag q4geo-noOL 2>&1 | grep -E "rep|Error|error" >> $O/sweep.txt
cat $O/sweep.txt
