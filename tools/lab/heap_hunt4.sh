#!/bin/bash
# fourth hunt: what about the fork / join capture loop is the trigger?  torch work only (no effort library in the process) unless said.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/hunt4; mkdir -p $OUT
gcc -O1 -g -shared -fPIC -o $OUT/abrt_bt.so tools/lab/abrt_bt.c || exit 1
PRE="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0:$PWD/$OUT/abrt_bt.so"
run() {   # name, env prefix ("perturb" or "plain"), args...
  name=$1; mode=$2; shift 2
  for rep in 1 2 3 4; do
    if [ $mode = perturb ]; then
      MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$PRE timeout 300 python -X faulthandler tools/lab/graph_event_repro.py "$@" > $OUT/${name}_$rep.out 2> $OUT/${name}_$rep.err
    else
      LD_PRELOAD=$OUT/abrt_bt.so timeout 300 python -X faulthandler tools/lab/graph_event_repro.py "$@" > $OUT/${name}_$rep.out 2> $OUT/${name}_$rep.err
    fi
    echo "$name ($mode) rep $rep rc=$? : $(tail -1 $OUT/${name}_$rep.out)" | tee -a $OUT/summary.txt
  done
}
run torch_destroy        perturb --work torch --sync events --iters 4000
run torch_keep           perturb --work torch --sync events --iters 4000 --keep 1
run torch_1stream        perturb --work torch --sync events --iters 4000 --streams 1
run torch_nograph        perturb --work torch --sync events --iters 4000 --nograph 1
run torch_destroy_plain  plain   --work torch --sync events --iters 8000
run effort_keep          perturb --work effort --sync events --iters 2000 --keep 1
run effort_1stream       perturb --work effort --sync events --iters 2000 --streams 1
run effort_1stream_keep  perturb --work effort --sync events --iters 2000 --streams 1 --keep 1
run effort_nograph       perturb --work effort --sync events --iters 2000 --nograph 1
