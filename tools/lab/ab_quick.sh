#!/bin/bash
# tools/lab/ab_quick.sh A.so : lone / group-of-three / layer / decode timings of build/variants/A.so against the in-tree library, alternating
export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O; rm -f $O/ab.txt
A=build/variants/${1:-prev}.so
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export EFFORT_HIP_LIB=$A; else unset EFFORT_HIP_LIB; fi
  timeout 200 python tools/qbench.py --group 1 --reps 2 --tag lone-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 3 --reps 2 --tag three-$v 2>&1 | grep "rep 1" >> $O/ab.txt
  echo "layer_probe $v" >> $O/ab.txt; timeout 300 python tools/layer_probe.py 2>&1 | tail -1 >> $O/ab.txt
done
done
for v in base new base new; do
  if [ $v = base ]; then export EFFORT_HIP_LIB=$A; else unset EFFORT_HIP_LIB; fi
  echo "== decode $v" >> $O/ab.txt
  timeout 400 python tools/decode_bench.py --efforts 0.25 --tokens 64 2>&1 | tail -1 | cut -c1-420 >> $O/ab.txt
done
