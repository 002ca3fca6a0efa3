#!/bin/bash
# a lone FP16 call at effort >= 0.5 streams for 11-20 us on 192 of 256 CUs (6 tiles x 32 slices at E = 2): does an
# effort-aware geometry with >= 256 items pay for the FFN shapes?  us per launch, lone calls back to back in one hipGraph over 32 matrices.
export TMPDIR=/tmp
O=gpurun_out/lonegeo; mkdir -p $O; rm -f $O/sweep.txt
CFG="0,0,0:-1;8,2,40:0;8,2,48:0;8,2,64:0;8,1,24:0;8,1,32:0;8,4,64:0;8,4,88:0"
for shape in 4096x11008 4096x14336; do
for e in 0.25 0.5 1.0; do
  timeout 600 python tools/qbench.py --shape $shape --group 1 --effort $e --reps 2 --configs "$CFG" --tag lone-$shape 2>&1 | grep -E "rep|rror" >> $O/sweep.txt
done
done
cat $O/sweep.txt
