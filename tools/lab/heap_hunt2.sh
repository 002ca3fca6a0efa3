#!/bin/bash
# second hunt: the dense-baseline capture / replay / destroy loop, one variant per process, under the checking allocator + native backtrace
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/hunt2; mkdir -p $OUT
gcc -O1 -g -shared -fPIC -o $OUT/abrt_bt.so tools/lab/abrt_bt.c || exit 1
PRE="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0:$PWD/$OUT/abrt_bt.so"
for rep in 1 2; do
for v in "rocblas lane" "rocblas four" "hip lane" "hip four"; do
  set -- $v
  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$PRE timeout 400 python -X faulthandler tools/lab/dense_graph_repro.py --backend $1 --job $2 --iters ${ITERS:-400} > $OUT/$1_$2_$rep.out 2> $OUT/$1_$2_$rep.err
  echo "rep $rep $1 $2 rc=$? : $(tail -1 $OUT/$1_$2_$rep.out)" | tee -a $OUT/summary.txt
done
done
grep -l "invalid pointer\|signal 6\|signal 11" $OUT/*.err 2>/dev/null | while read f; do echo "=== $f"; grep -v "^MAP" $f | tail -60; done | tee $OUT/failures.txt | cut -c1-250 | tail -150
