#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
for tc in 0 4 8 12 16; do for tm in 2 4; do
  [ $tc = 0 ] && [ $tm = 4 ] && continue
  EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=$tm timeout 120 python tools/qbench.py --group 32 --reps 1 --steps-per-graph 8 --tag "tail $tc x$tm lanes1" 2>&1 | grep -v "amdgpu\|Warn"
  EFFORT_TAIL_CALLS=$tc EFFORT_TAIL_MULT=$tm timeout 120 python tools/qbench.py --group 32 --reps 1 --steps-per-graph 8 --overlap 2 --tag "tail $tc x$tm lanes2" 2>&1 | grep -v "amdgpu\|Warn"
done; done | tee $O/tail.log
