#!/bin/bash
# round 6 (third session): effort_set_row_reuse as TWO copies of the streaming loop -- does the default (nt) path keep its speed against HEAD
# (build/variants/head.so: the one-policy kernels), and does reuse = 1 give the shared-matrices caller the ordinary policy's 117 us back?
export TMPDIR=/tmp
O=gpurun_out/b29; mkdir -p $O; rm -f $O/ab.txt $O/pytest.log
timeout 600 python -m pytest tests -m gpu -x -q -k "row_reuse or plain_and_persistent or group_launch" 2>&1 | tail -3 > $O/pytest.log
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
for rep in 1 2; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q --group 1 --tag lone-$v
  q --group 32 --tag big1-$v
  q --group 32 --mats 128 --overlap 4 --steps-per-graph 4 --tag disjoint4-$v
  q --group 32 --overlap 4 --steps-per-graph 8 --tag shared4-$v
  q --shape 4096x4096 --group 32 --tag sq32-$v
  q --q4 1 --group 16 --tag q4x16-$v
done
unset EFFORT_HIP_LIB
q --group 32 --overlap 4 --steps-per-graph 8 --row-reuse 1 --tag shared4-new-reuse
q --shape 4096x4096 --group 32 --row-reuse 1 --tag sq32-new-reuse
q --group 32 --mats 128 --overlap 4 --steps-per-graph 4 --row-reuse 1 --tag disjoint4-new-reuse
q --group 32 --row-reuse 1 --tag big1-new-reuse
q --group 1 --row-reuse 1 --tag lone-new-reuse
q --q4 1 --group 16 --overlap 4 --steps-per-graph 8 --tag q4shared4-new
q --q4 1 --group 16 --overlap 4 --steps-per-graph 8 --row-reuse 1 --tag q4shared4-new-reuse
done
cat $O/pytest.log $O/ab.txt
