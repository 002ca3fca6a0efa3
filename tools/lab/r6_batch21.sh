#!/bin/bash
# round 6 (second session): the tree (rows nt, dense baseline nt) against build/variants/aux0.so (the library before) on ONE box: the -m gpu suite,
# quick launches alternating, then bench.py with each
export TMPDIR=/tmp
O=gpurun_out/b21; mkdir -p $O; rm -f $O/ab.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)" >> $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for rep in 1 2; do
for v in aux0 tree; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  q --group 32 --tag g32-$v
  q --group 1 --tag lone-$v
  q --group 16 --tag g16-$v
  q --group 32 --effort 0.1 --tag g32e10-$v
  q --group 32 --effort 1.0 --tag g32e100-$v
  q --group 16 --shape 4096x4096 --mats 64 --tag sq16-$v
done
done
for v in aux0 tree aux0 tree; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  timeout 300 python bench.py > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY >> $O/ab.txt
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("bench $v", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["single_stream"], d["extra"])
PY
done
cat $O/ab.txt
