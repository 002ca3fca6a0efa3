#!/bin/bash
# round 6, third batch: Q4 per-item timelines at other geometries; split cutoffs; the power-of-two slice rule on the timeit shape
export TMPDIR=/tmp
O=gpurun_out/b3; mkdir -p $O
for t in "8,2,16" "8,1,8"; do
  timeout 300 python tools/timeline.py --q4 1 --groups 16 --replay 10 --tune $t --persistent 0 --out $O/tl.json 2>&1 | grep -v amdgpu | head -24 > $O/timeline_q4_$(echo $t | tr , _).txt; cat $O/timeline_q4_$(echo $t | tr , _).txt
done
for s in 0 1; do
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --split $s --tag q4x16-split$s 2>&1 | grep "rep 1"
  timeout 200 python tools/qbench.py --group 16 --reps 2 --split $s --tag fp16x16-split$s 2>&1 | grep "rep 1"
done | tee $O/split_cutoff.txt
timeout 300 python tools/lab/geosweep.py --launch 4096x14336 --effort 0.25 --configs "0,0,0;8,2,32;8,2,24" 2>&1 | grep -v amdgpu | tee $O/geo_14336.txt
timeout 300 python tools/lab/geosweep.py --launch 4096x14336 --effort 0.5 --configs "0,0,0;8,2,32;8,2,24" 2>&1 | grep -v amdgpu | tee -a $O/geo_14336.txt
