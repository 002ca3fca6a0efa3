#!/bin/bash
# round-3 GPU run 3: full parity suite on the new tree (lanes, pitched layout, VALU bisection, blocked selection, exact fused rmsNorm), lone-call timings, bench
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 120 python tools/cutprof.py > $O/cutprof.log 2>&1; grep -v amdgpu $O/cutprof.log
for shape in 4096x11008 4096x4096 14336x4096; do for g in 1 3; do
  timeout 120 python tools/qbench.py --shape $shape --group $g --reps 2 --tag "r3c $shape"
done; done > $O/qb_lone.log 2>&1
grep -v "Warn\|amdgpu" $O/qb_lone.log
HIP_FORCE_DEV_KERNARG=1 timeout 120 python tools/qbench.py --group 1 --reps 2 --tag "devkernarg1" 2>&1 | grep -v "Warn\|amdgpu"
HIP_FORCE_DEV_KERNARG=0 timeout 120 python tools/qbench.py --group 1 --reps 2 --tag "devkernarg0" 2>&1 | grep -v "Warn\|amdgpu"
timeout 300 python tools/decode_ab.py --efforts 0.25 > $O/decode_ab.json 2> $O/decode_ab.log; cat $O/decode_ab.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; tail -5 $O/bench.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','timed_region_ms','timed_replays','roofline','by_group_size','by_streams','shared_matrices','four_contexts','shard_projection','decode','cpu_baseline','dense_hip_kernel'):
    print(k, json.dumps(d.get(k))[:1500])
PY
