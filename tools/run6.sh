#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; grep -v amdgpu $O/bench.log | tail -12; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3e/bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','timed_region_ms','timed_replays','roofline','by_group_size','by_streams','shared_matrices','four_contexts','shard_projection','decode','cpu_baseline','dense_hip_kernel','other_configs'):
    print(k, json.dumps(d.get(k))[:2200])
PY
