#!/bin/bash
# third hunt: fork / join captures, torch work or the library's, temporary events (torch wait_stream) or events that live for the process
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/hunt3; mkdir -p $OUT
gcc -O1 -g -shared -fPIC -o $OUT/abrt_bt.so tools/lab/abrt_bt.c || exit 1
PRE="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0:$PWD/$OUT/abrt_bt.so"
for rep in 1 2 3; do
for v in "torch wait_stream" "torch events" "effort wait_stream" "effort events"; do
  set -- $v
  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$PRE timeout 600 python -X faulthandler tools/lab/graph_event_repro.py --work $1 --sync $2 --iters ${ITERS:-3000} > $OUT/$1_$2_$rep.out 2> $OUT/$1_$2_$rep.err
  echo "rep $rep $1 $2 rc=$? : $(tail -1 $OUT/$1_$2_$rep.out)" | tee -a $OUT/summary.txt
done
done
grep -l "invalid pointer\|signal 6\|signal 11\|corrupt" $OUT/*.err 2>/dev/null | while read f; do echo "=== $f"; grep -v "^MAP" $f | head -40; done > $OUT/failures.txt
cut -c1-200 $OUT/failures.txt | head -120
