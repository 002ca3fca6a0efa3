#!/bin/bash
# Hunt for the once-in-40-runs `free(): invalid pointer` of round 5's full bench run: the measurement process itself
# (BENCH_CHILD=1: no guard) N times under glibc's checking allocator (MALLOC_CHECK_=3 via libc_malloc_debug, MALLOC_PERTURB_) with a
# native backtrace on SIGABRT / SIGSEGV (tools/lab/abrt_bt.c) and Python's faulthandler; stops at the first run that dies.
#   bash tools/lab/heap_hunt.sh [runs=30] [mode=bench|soak|both] [extra bench flags...]
set -u
cd "$(dirname "$0")/../.."
RUNS=${1:-30}; MODE=${2:-bench}; shift 2 2>/dev/null || true
OUT=gpurun_out/hunt; mkdir -p $OUT
gcc -O1 -g -shared -fPIC -o $OUT/abrt_bt.so tools/lab/abrt_bt.c || exit 1
PRE="$PWD/$OUT/abrt_bt.so"
[ "${HUNT_MALLOC_DEBUG:-1}" = 1 ] && PRE="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0:$PRE"
fails=0
for i in $(seq 1 $RUNS); do
  t0=$(date +%s)
  if [ $MODE = bench ] || [ $MODE = both ]; then
    MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$PRE BENCH_CHILD=1 PYTHONFAULTHANDLER=1 timeout 600 \
      python -X faulthandler bench.py --steps 20 --warmup 5 "$@" > $OUT/bench_$i.out 2> $OUT/bench_$i.err
    rc=$?
    echo "run $i bench rc=$rc $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
    if [ $rc -ne 0 ]; then fails=$((fails+1)); cp $OUT/bench_$i.err $OUT/FAILED_bench_$i.err; [ "${HUNT_KEEP_GOING:-0}" = 1 ] || break; else rm -f $OUT/bench_$i.err $OUT/bench_$i.out; fi
  fi
  if [ $MODE = soak ] || [ $MODE = both ]; then
    t0=$(date +%s)
    MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$PRE PYTHONFAULTHANDLER=1 EFFORT_SOAK_SECONDS=${SOAK_SECONDS:-30} EFFORT_SOAK_SEED=$((1000 + i)) timeout 900 \
      python -X faulthandler -m pytest tests/test_gpu_soak.py tests/test_gpu_layer.py tests/test_gpu_overlap.py -m gpu -x -q > $OUT/soak_$i.out 2> $OUT/soak_$i.err
    rc=$?
    echo "run $i soak rc=$rc $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
    if [ $rc -ne 0 ]; then fails=$((fails+1)); cp $OUT/soak_$i.out $OUT/FAILED_soak_$i.out; cp $OUT/soak_$i.err $OUT/FAILED_soak_$i.err; [ "${HUNT_KEEP_GOING:-0}" = 1 ] || break; else rm -f $OUT/soak_$i.err $OUT/soak_$i.out; fi
  fi
done
echo "done: $fails failing run(s)" | tee -a $OUT/summary.txt
