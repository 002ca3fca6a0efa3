#!/bin/bash
# round 6: Q4 around "one round of workgroups, narrow tiles, tall slices" (E = 1, 5 slices: 58.5 us against the heuristic's 65.5 at 16 per launch)
export TMPDIR=/tmp
O=gpurun_out/b12; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 400 python tools/qbench.py --q4 1 --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt; }
q --group 16 --configs "0,0,0:-1;8,1,5:0;8,1,4:0;8,2,5:0;16,2,5:0;16,1,5:0;8,1,6:0;8,1,5:2" --tag q4x16
q --group 16 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,1,5:0" --tag q4x16x4
q --group 12 --configs "0,0,0:-1;8,1,7:0;8,1,6:0;8,1,5:0;8,2,7:0" --tag q4x12
q --group 10 --configs "0,0,0:-1;8,1,8:0;8,1,7:0" --tag q4x10
q --group 8 --configs "0,0,0:-1;8,1,5:0;8,1,4:0;8,2,5:0" --tag q4x8
q --group 4 --configs "0,0,0:-1;8,1,8:0;8,1,10:0;8,2,8:0" --tag q4x4
q --group 2 --configs "0,0,0:-1;8,1,16:0;8,1,20:0;8,1,8:0" --tag q4x2
q --group 1 --configs "0,0,0:-1;8,1,32:0;8,1,40:0;8,1,16:0;8,2,32:0" --tag q4x1
q --group 32 --configs "0,0,0:-1;8,1,5:-1;8,1,5:0;8,1,4:-1" --tag q4x32
q --group 24 --configs "0,0,0:-1;8,1,5:-1;8,1,5:0" --tag q4x24
cat $O/sweep.txt
