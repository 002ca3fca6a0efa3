#!/bin/bash
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cfs $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" 
python - <<'PY'
import numpy as np, sys, os, subprocess, json
sys.path.insert(0,'/root/repo')
from oracle import cpu
rng=np.random.default_rng(0)
W=(rng.standard_normal((11008,4096),dtype=np.float32)*0.02).astype(np.float16)
b,s,p,_=cpu.convert_fp16(W)
os.makedirs('/dev/shm/cb',exist_ok=True)
np.save('/dev/shm/cb/b0.npy',b); np.save('/dev/shm/cb/s0.npy',s); np.save('/dev/shm/cb/p0.npy',p); np.save('/dev/shm/cb/v.npy',rng.standard_normal(4096,dtype=np.float32))
for th in (4,8,16,32,64,128):
    for bind in ("spread","close","false"):
        env=dict(os.environ, OMP_NUM_THREADS=str(th), OMP_WAIT_POLICY="active", OMP_PROC_BIND=bind, OMP_PLACES="cores")
        if bind=="false": env.pop("OMP_PLACES")
        r=subprocess.run([sys.executable,'/root/repo/oracle/cpu_bench.py','/dev/shm/cb','4096','11008','0.25','1.5','1'],env=env,capture_output=True,text=True)
        try: d=json.loads(r.stdout.strip().split('\n')[-1]); print(th, bind, round(d['seconds_per_call']*1e3,3),'ms')
        except Exception as e: print(th,bind,'fail',r.stderr[-200:])
PY
