#!/bin/bash
python - <<'PY'
import numpy as np, sys, os, subprocess, json
sys.path.insert(0,'/root/repo')
from oracle import cpu
rng=np.random.default_rng(0)
os.makedirs('/dev/shm/cb',exist_ok=True)
for k in range(4):
    W=(rng.standard_normal((11008,4096),dtype=np.float32)*0.02).astype(np.float16)
    b,s,p,_=cpu.convert_fp16(W)
    np.save(f'/dev/shm/cb/b{k}.npy',b); np.save(f'/dev/shm/cb/s{k}.npy',s); np.save(f'/dev/shm/cb/p{k}.npy',p)
np.save('/dev/shm/cb/v.npy',rng.standard_normal(4096,dtype=np.float32))
for th in (8,16,32):
    for bind in ("close","spread"):
        env=dict(os.environ, OMP_NUM_THREADS=str(th), OMP_WAIT_POLICY="active", OMP_PROC_BIND=bind, OMP_PLACES="cores")
        r=subprocess.run([sys.executable,'/root/repo/oracle/cpu_bench.py','/dev/shm/cb','4096','11008','0.25','2','4'],env=env,capture_output=True,text=True)
        try: d=json.loads(r.stdout.strip().split('\n')[-1]); print(th, bind, round(d['seconds_per_call']*1e3,3),'ms')
        except Exception as e: print(th,bind,'fail',r.stderr[-300:])
PY
