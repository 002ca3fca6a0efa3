#!/bin/bash
# Where a 16-call Q4 launch spends its time (DESIGN 4.1 "Q4: where a launch's time goes"): the same launch with the LDS scatter
# compiled out, with the row loads skipped (EFFORT_ABLATE=4), without the outlier tables, and with neither.  Run on a GPU box from the
# repo root AFTER building the LAB variants here (hipcc cross-compiles; the shipped library has the EFFORT_ABLATE switches compiled out):
#   tools/build_variant.sh noscatter "-DEFFORT_LAB -DEFFORT_ABLATE_NOSCATTER=1"      (the run-time switches are in the in-tree lab library)
#   gpurun --timeout 600 -- 'bash tools/q4_ablate.sh'
export TMPDIR=/tmp
export EFFORT_HIP_LIB=lab
Q="timeout 200 python tools/qbench.py --q4 1 --group ${GROUP:-16} --reps 1"
echo "== whole";                      $Q --tag whole
echo "== without outliers";           $Q --no-outliers 1 --tag no-outliers
echo "== without streaming";          EFFORT_ABLATE=4 $Q --tag no-stream
echo "== neither";                    EFFORT_ABLATE=4 $Q --no-outliers 1 --tag neither
[ -f build/variants/noscatter.so ] && { echo "== loads without the LDS scatter"; EFFORT_HIP_LIB=build/variants/noscatter.so $Q --tag no-scatter; }
