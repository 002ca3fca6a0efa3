#!/bin/bash
# round 6 (third session): thinner slices for the LAST calls of a launch (EFFORT_TAIL_CALLS / EFFORT_TAIL_MULT, lab library), ONE launch in flight --
# round 2 had measured it with several launches in flight only (no gain: a launch's tail runs under the next one's head).
export TMPDIR=/tmp EFFORT_HIP_LIB=lab
O=gpurun_out/b30; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-130 >> $O/sweep.txt; }
for pass in 1 2; do
for tc in 0 4 8 12 16 32; do
for tm in 2 4; do
  if [ $tc = 0 ] && [ $tm = 4 ]; then continue; fi
  export EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=$tm
  q --group 32 --configs "0,0,0:-1;0,0,0:2" --tag g32-tc$tc-x$tm
done
done
done
for tc in 0 4 8; do
  export EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=2
  q --group 16 --mats 16 --configs "0,0,0:-1" --tag g16-tc$tc-x2
  q --group 32 --effort 0.5 --configs "0,0,0:-1" --tag g32e50-tc$tc-x2
done
cat $O/sweep.txt
