#!/bin/bash
# round 6 (second session): launches of up to three items per CU as PLAIN grids (lean kernel, the dispatcher refills the CUs) against persistent
# workgroups pulling from the queues (the heuristic from 513 items on), one launch at a time and four in flight on disjoint matrices
export TMPDIR=/tmp
O=gpurun_out/b24; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/sweep.txt; }
q --group 16 --configs "0,0,0:-1;0,0,0:0;8,2,8:0;8,2,8:2" --tag g16
q --group 16 --mats 64 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;0,0,0:0" --tag g16x4
q --group 32 --configs "0,0,0:-1;0,0,0:0" --tag g32
q --group 32 --mats 128 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;0,0,0:0" --tag g32x4
q --group 24 --mats 24 --configs "0,0,0:-1;0,0,0:0;8,2,8:0;8,4,8:0" --tag g24
q --group 12 --mats 24 --configs "0,0,0:-1;0,0,0:0" --tag g12
q --group 20 --mats 20 --configs "0,0,0:-1;0,0,0:0;8,2,8:0;8,4,8:0" --tag g20
q --group 32 --effort 0.5 --configs "0,0,0:-1;0,0,0:0" --tag g32e50
q --group 32 --effort 1.0 --configs "0,0,0:-1;0,0,0:0" --tag g32e100
q --group 32 --effort 0.1 --configs "0,0,0:-1;0,0,0:0" --tag g32e10
q --group 16 --effort 0.5 --configs "0,0,0:-1;0,0,0:0" --tag g16e50
q --group 16 --shape 4096x4096 --mats 64 --configs "0,0,0:-1;0,0,0:0;8,2,8:0;8,1,8:0" --tag sq16
q --group 32 --shape 4096x4096 --mats 64 --configs "0,0,0:-1;0,0,0:0;8,2,8:0;8,1,8:0" --tag sq32
q --group 16 --shape 4096x14336 --mats 32 --configs "0,0,0:-1;0,0,0:0;8,2,8:0" --tag w1x16
cat $O/sweep.txt
