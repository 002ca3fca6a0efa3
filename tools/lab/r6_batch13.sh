#!/bin/bash
# round 6: the shipped Q4 one-round rule: suite, then heuristic (0,0,0:-1) numbers per group size, one launch in flight and four
export TMPDIR=/tmp
O=gpurun_out/b13; mkdir -p $O; rm -f $O/sweep.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
q() { timeout 400 python tools/qbench.py --q4 1 --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt; }
for g in 1 2 3 4 8 10 12 16 32; do q --group $g --tag q4x$g; done
q --group 16 --overlap 4 --steps-per-graph 8 --tag q4x16x4
q --group 12 --overlap 4 --steps-per-graph 8 --tag q4x12x4
q --group 16 --configs "8,1,5:0;8,2,8:0" --tag q4x16-forced
cat $O/sweep.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -c 700 $O/bench_line.json
