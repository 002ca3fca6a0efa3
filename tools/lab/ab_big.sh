#!/bin/bash
# tools/lab/ab_big.sh A.so : the throughput launches (32 calls per launch, persistent) of build/variants/A.so against the in-tree library, alternating
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O; rm -f $O/ab.txt
A=build/variants/${1:-head}.so
for rep in 1 2 3; do
for v in base new; do
  if [ $v = base ]; then export EFFORT_HIP_LIB=$A; else unset EFFORT_HIP_LIB; fi
  timeout 200 python tools/qbench.py --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag big4-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 32 --reps 2 --tag big1-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  timeout 200 python tools/qbench.py --shape 4096x4096 --group 32 --reps 2 --tag sq32-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --tag q4x16-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline-$v', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
done
