#!/bin/bash
# round 6 (third session): the thin-tail rule in the product library (api.hip do_group: thinFrom) against the library without it (build/variants/head.so), one launch in flight
export TMPDIR=/tmp
O=gpurun_out/b33; mkdir -p $O; rm -f $O/ab.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or plain_and_persistent or bench_geometry" 2>&1 | tail -3 > $O/pytest.log
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  for n in 11 12 13 14 15 16 17 18 19 20 21 22 24 26 28 30 32; do q --group $n --mats $n --tag n$n-$v; done
  q --group 16 --mats 16 --effort 0.5 --tag n16e50-$v
  q --group 16 --mats 16 --effort 0.1 --tag n16e10-$v
  q --group 12 --mats 12 --effort 0.5 --tag n12e50-$v
  q --group 16 --mats 16 --effort 1.0 --tag n16e100-$v
  q --group 16 --mats 16 --shape 4096x14336 --tag w1n16-$v
  q --group 16 --mats 16 --shape 14336x4096 --tag w2n16-$v
  q --group 24 --mats 24 --shape 14336x4096 --tag w2n24-$v
  q --group 24 --mats 24 --shape 4096x14336 --tag w1n24-$v
  q --group 16 --mats 32 --shape 4096x4096 --tag sqn16-$v
  q --group 32 --mats 64 --shape 4096x4096 --tag sqn32-$v
  q --group 16 --mats 64 --overlap 4 --steps-per-graph 4 --tag n16x4lanes-$v
done
cat $O/pytest.log $O/ab.txt
