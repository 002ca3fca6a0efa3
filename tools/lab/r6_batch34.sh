#!/bin/bash
# round 6 (third session): the NARROWED thin-tail rule (last round at most an eighth full) against the library without it, case by case, alternating
export TMPDIR=/tmp
O=gpurun_out/b34; mkdir -p $O; rm -f $O/ab.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak" 2>&1 | tail -3 > $O/pytest.log
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
ab() { tag=$1; shift; for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q "$@" --tag $tag-$v; done; }
ab n11 --group 11 --mats 11
ab n12 --group 12 --mats 12
ab n22 --group 22 --mats 22
ab n23 --group 23 --mats 23
ab n12e10 --group 12 --mats 12 --effort 0.1
ab n12e50 --group 12 --mats 12 --effort 0.5
ab n12e100 --group 12 --mats 12 --effort 1.0
ab n22e50 --group 22 --mats 22 --effort 0.5
ab n11e100 --group 11 --mats 11 --effort 1.0
ab w1n19 --group 19 --mats 19 --shape 4096x14336
ab w2n17 --group 17 --mats 17 --shape 14336x4096
ab n12m32 --group 12 --mats 36
ab n16 --group 16 --mats 16
cat $O/pytest.log $O/ab.txt
