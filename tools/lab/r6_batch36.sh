#!/bin/bash
# round 6 (third session): the shipped thin-tail rule (last round at most 7/32 full) -- the whole GPU suite, then the cases at the rule's edge against the library without it
export TMPDIR=/tmp
O=gpurun_out/b36; mkdir -p $O; rm -f $O/ab.txt $O/pytest.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -5 > $O/pytest.log
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
ab() { tag=$1; shift; for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q "$@" --tag $tag-$v; done; }
ab n13 --group 13 --mats 13
ab n23 --group 23 --mats 23
ab n24 --group 24 --mats 24
ab w1n20 --group 20 --mats 20 --shape 4096x14336
ab w1n11 --group 11 --mats 11 --shape 4096x14336
ab n13e50 --group 13 --mats 13 --effort 0.5
ab n23e100 --group 23 --mats 23 --effort 1.0
cat $O/pytest.log $O/ab.txt
