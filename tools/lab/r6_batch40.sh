#!/bin/bash
# round 6 (third session): groups of >= 8 calls on SMALL matrices take enough slices to give every CU an item (pick_slices: fill) -- against the library without it
export TMPDIR=/tmp
O=gpurun_out/b40; mkdir -p $O; rm -f $O/ab.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or experts" 2>&1 | tail -3 > $O/pytest.log
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
ab() { tag=$1; shift; for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q "$@" --tag $tag-$v; done; }
ab sq8 --group 8 --mats 32 --shape 4096x4096
ab sq9 --group 9 --mats 36 --shape 4096x4096
ab sq10 --group 10 --mats 30 --shape 4096x4096
ab sq12 --group 12 --mats 36 --shape 4096x4096
ab sq8e50 --group 8 --mats 32 --shape 4096x4096 --effort 0.5
ab sq8e100 --group 8 --mats 32 --shape 4096x4096 --effort 1.0
ab sq8e10 --group 8 --mats 32 --shape 4096x4096 --effort 0.1
ab kv8 --group 8 --mats 64 --shape 4096x1024
ab kv16 --group 16 --mats 64 --shape 4096x1024
ab kv32 --group 32 --mats 64 --shape 4096x1024
ab kv8e50 --group 8 --mats 64 --shape 4096x1024 --effort 0.5
ab n8 --group 8 --mats 16
ab w2n8 --group 8 --mats 16 --shape 14336x4096
ab h8 --group 8 --mats 32 --shape 4096x2048
ab h16 --group 16 --mats 32 --shape 4096x2048
cat $O/pytest.log $O/ab.txt
