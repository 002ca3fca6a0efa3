#!/bin/bash
# round 6: FP16 groups as ONE round of workgroups now that slice counts that are not multiples of 8 cost no padding blocks
export TMPDIR=/tmp
O=gpurun_out/b15; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 400 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 16 --configs "0,0,0:-1;8,4,10:0;8,4,9:0;8,4,11:0;8,2,8:0;8,4,12:0" --tag fp16x16
q --group 12 --configs "0,0,0:-1;8,4,14:0;8,4,13:0;8,4,12:0;8,4,10:0" --tag fp16x12
q --group 10 --configs "0,0,0:-1;8,4,16:0;8,4,17:0;8,4,12:0" --tag fp16x10
q --group 8 --configs "0,0,0:-1;8,4,21:0;8,4,20:0;8,4,16:0;8,4,12:0;8,4,10:0" --tag fp16x8
q --group 16 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,4,10:0" --tag fp16x16x4
q --group 16 --effort 0.5 --configs "0,0,0:-1;8,4,10:0" --tag fp16x16e50
q --shape 4096x4096 --group 16 --configs "0,0,0:-1;8,2,16:0;8,2,15:0;8,2,14:0;8,1,8:0;8,4,32:0" --tag sq16
q --shape 4096x4096 --group 32 --configs "0,0,0:-1;8,2,8:0;8,2,7:0;8,4,16:0;8,4,15:0" --tag sq32
cat $O/sweep.txt
