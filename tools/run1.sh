#!/bin/bash
# round-3 GPU run 1: parity of the tree as it is + A/B timings (prefetch under the cutoff, kernel-argument size, stamps, persistent geometries)
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python tools/cutprof.py > $O/cutprof.log 2>&1; cat $O/cutprof.log
for shape in 4096x11008 4096x4096 14336x4096 4096x14336; do for g in 1 2 3; do for pf in 0 1; do
  EFFORT_PREFETCH=$pf timeout 120 python tools/qbench.py --shape $shape --group $g --reps 2 --tag "pf$pf $shape"
done; done; done > $O/qb_pref.log 2>&1
cat $O/qb_pref.log | grep -v Warn
for v in nostamps maxgroup4; do for g in 1 3; do for pf in 0 1; do
  EFFORT_PREFETCH=$pf EFFORT_HIP_LIB=build/variants/$v.so timeout 120 python tools/qbench.py --group $g --reps 2 --tag "$v pf$pf"
done; done; done > $O/qb_variants.log 2>&1
cat $O/qb_variants.log | grep -v Warn
timeout 200 python tools/qbench.py --group 32 --reps 2 --configs "0,0,0:-1;4,4,16:4;4,4,0:4;8,4,16:2;4,2,16:4;16,4,0:1" > $O/qb_persist.log 2>&1
cat $O/qb_persist.log | grep -v Warn
timeout 300 python tools/decode_ab.py > $O/decode_ab.json 2> $O/decode_ab.log; cat $O/decode_ab.json
