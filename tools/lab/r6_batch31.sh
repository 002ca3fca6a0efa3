#!/bin/bash
# round 6 (third session): the one point of r6_batch30 that was not slower -- 16 calls per launch with the last 4 at twice the slices -- looked at again
export TMPDIR=/tmp EFFORT_HIP_LIB=lab
O=gpurun_out/b31; mkdir -p $O; rm -f $O/sweep.txt
q() { timeout 300 python tools/qbench.py --reps 3 "$@" 2>&1 | grep -E "rep [12]|rror" | cut -c1-130 >> $O/sweep.txt; }
for pass in 1 2; do
for tc in 0 2 4 6 8; do
  export EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=2
  q --group 16 --mats 16 --tag g16-tc$tc
done
for tc in 0 4; do
  export EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=2
  q --group 12 --mats 12 --tag g12-tc$tc
  q --group 20 --mats 20 --tag g20-tc$tc
  q --group 24 --mats 24 --tag g24-tc$tc
  q --group 16 --mats 32 --tag g16m32-tc$tc
done
done
cat $O/sweep.txt
