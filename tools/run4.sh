#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_prologues or column_shards_of_the_baseline" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; grep -v amdgpu $O/bench.log | tail -40; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','timed_region_ms','timed_replays','roofline','by_group_size','by_streams','shared_matrices','four_contexts','shard_projection','decode','cpu_baseline','dense_hip_kernel'):
    print(k, json.dumps(d.get(k))[:1800])
PY
