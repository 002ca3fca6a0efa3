#!/bin/bash
# round 6 (second session): nt on the row stream, the cache-honest overlapped job (4 launches in flight on 4 x 32 DISJOINT matrices) and the bench line itself
export TMPDIR=/tmp
O=gpurun_out/b17; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in base nt; do
  if [ $v = nt ]; then export EFFORT_HIP_LIB=$PWD/build/variants/nt.so; else unset EFFORT_HIP_LIB; fi
  q --mats 128 --group 32 --overlap 4 --steps-per-graph 8 --tag g32x4disjoint-$v
  q --mats 128 --group 32 --overlap 2 --steps-per-graph 8 --tag g32x2disjoint-$v
  q --mats 128 --group 32 --tag g32disjoint-$v
done
done
for v in base nt; do
  if [ $v = nt ]; then export EFFORT_HIP_LIB=$PWD/build/variants/nt.so; else unset EFFORT_HIP_LIB; fi
  timeout 300 python bench.py > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY >> $O/ab.txt
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("bench $v", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["single_stream"], d["extra"])
PY
done
cat $O/ab.txt
