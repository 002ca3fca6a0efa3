#!/bin/bash
# round 6 (second session): launch geometries re-swept with the row stream nt (waves,elems,slices:workgroups-per-CU against the heuristic 0,0,0:-1),
# and what a workgroup ALONE on its CU pulls (persistent, one per CU)
export TMPDIR=/tmp
O=gpurun_out/b23; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/sweep.txt; }
q --group 32 --configs "0,0,0:-1;8,4,8:2;8,4,8:1;16,4,8:1;16,4,16:1;8,2,8:2;8,4,12:2;8,4,16:2;8,4,10:2;8,4,8:0" --tag g32
q --group 16 --configs "0,0,0:-1;8,4,8:2;8,4,8:0;8,2,8:0;8,2,8:2;8,4,16:2;8,4,12:0;16,4,8:1" --tag g16
q --group 1 --configs "0,0,0:-1;8,2,32:0;8,4,32:0;8,2,40:0;8,1,32:0;8,2,24:0;8,4,64:0" --tag lone
q --group 3 --configs "0,0,0:-1;8,2,16:0;8,4,16:0;8,2,32:0;8,4,32:0;8,2,24:0" --tag three
q --group 32 --effort 0.5 --configs "0,0,0:-1;8,4,16:2;16,4,8:1" --tag g32e50
cat $O/sweep.txt
