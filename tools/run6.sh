#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; grep -v amdgpu $O/bench.log | tail -12; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3e/bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','roofline','by_streams','shared_matrices','shard_projection','cpu_baseline','other_configs'):
    print(k, json.dumps(d.get(k))[:3000])
PY
