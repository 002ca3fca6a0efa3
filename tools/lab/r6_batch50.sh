#!/bin/bash
# round 6 (third session): launches the power-of-two snap (or the `hi` bound) pushes OVER one item per CU: fewer slices that fit, by hand
export TMPDIR=/tmp
O=gpurun_out/b50; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [2]|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 9 --mats 36 --shape 8192x4096 --configs "0,0,0:-1;8,4,24:-1;8,4,28:-1;8,4,16:-1" --tag w8n9
q --group 3 --mats 48 --shape 4096x4096 --configs "0,0,0:-1;8,1,20:-1;8,1,16:-1;8,1,24:-1" --tag sq3
q --group 5 --mats 50 --shape 4096x4096 --configs "0,0,0:-1;8,1,12:-1;8,1,8:-1" --tag sq5
q --group 9 --mats 63 --shape 4096x1024 --configs "0,0,0:-1;8,1,24:-1;8,1,16:-1" --tag kv9
q --group 10 --mats 60 --shape 4096x1024 --configs "0,0,0:-1;8,1,24:-1;8,1,16:-1" --tag kv10
q --group 12 --mats 60 --shape 4096x1024 --configs "0,0,0:-1;8,1,16:-1;8,1,20:-1" --tag kv12
q --group 3 --mats 48 --shape 4096x2048 --configs "0,0,0:-1;8,1,32:-1;8,1,24:-1;8,1,16:-1" --tag h3
q --group 5 --mats 50 --shape 4096x2048 --configs "0,0,0:-1;8,1,24:-1;8,1,16:-1" --tag h5
cat $O/sweep.txt
