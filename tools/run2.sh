#!/bin/bash
# round-3 GPU run 2: no-scratch kernels, lanes (in-library overlap)
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for shape in 4096x11008 4096x4096 14336x4096; do for g in 1 3; do
  timeout 120 python tools/qbench.py --shape $shape --group $g --reps 2 --tag "noscratch $shape"
done; done > $O/qb_lone.log 2>&1
grep -v "Warn\|amdgpu" $O/qb_lone.log
timeout 200 python tools/qbench.py --group 32 --reps 2 --configs "0,0,0:-1" --steps-per-graph 8 > $O/qb_32.log 2>&1
timeout 200 python tools/qbench.py --group 32 --reps 2 --configs "0,0,0:-1" --steps-per-graph 8 --overlap 4 --tag overlap4 >> $O/qb_32.log 2>&1
timeout 200 python tools/qbench.py --group 32 --reps 2 --configs "0,0,0:-1" --steps-per-graph 8 --overlap 2 --tag overlap2 >> $O/qb_32.log 2>&1
timeout 200 python tools/qbench.py --group 32 --reps 2 --configs "0,0,0:-1" --steps-per-graph 8 --streams 4 --tag streams4 >> $O/qb_32.log 2>&1
grep -v "Warn\|amdgpu" $O/qb_32.log
timeout 300 python tools/decode_ab.py --efforts 0.25 > $O/decode_ab.json 2> $O/decode_ab.log; cat $O/decode_ab.json
timeout 100 python tools/timeline.py --groups 1 --out $O/tl_1.json > $O/tl_1.log 2>&1; head -40 $O/tl_1.log
