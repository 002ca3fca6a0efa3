#!/usr/bin/env python
"""The reference's timing loop (dependent lone calls into ONE output vector) on each of the context's four lanes in turn
(effort_set_overlap(4); a join moves the next chain to the next lane): does a lane cost more than another?"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effort_amd as ea
from bench import make_weights
dev = torch.device("cuda", 0)
g = ea.gpu(0)
lib = ea.lib()
extra = []
for _ in range(int(os.environ.get("EXTRA_CONTEXTS", "0"))):      # other contexts' lanes shift which hardware queue a lane's stream lands on
    x = ea.Gpu(0); x.set_overlap(4); extra.append(x)
side = [torch.cuda.Stream() for _ in range(int(os.environ.get("EXTRA_STREAMS", "0")))]
ws = make_weights(ea, 32, 4096, 14336, 777, dev, keep_core=False)
hs = [C.c_void_p(ew.handle) if not isinstance(ew.handle, C.c_void_p) else ew.handle for ew in ws]
v = torch.randn(4096, device=dev); t = torch.zeros(14336, device=dev)
vp, tp = C.c_void_p(v.data_ptr()), C.c_void_p(t.data_ptr())
fn, ctx = lib.effort_bucketmul, g.ctx
g._bind_stream()
def loop(n, s):
    rc = 0
    for i in range(n):
        rc |= fn(ctx, hs[i & 31], vp, None, tp, s)
    return rc
for lanes in (4,):
    g.set_overlap(lanes)
    for s in (0.5,):
        row = []
        for k in range(8):
            assert loop(300, s) == 0 if k == 0 else True
            g.eval()
            t0 = time.perf_counter()
            assert loop(1500, s) == 0
            t1 = time.perf_counter()
            g.eval()
            t2 = time.perf_counter()
            row.append((g.last_lane() if hasattr(g, "last_lane") else -1, round((t2 - t0) / 1500 * 1e6, 2), round((t1 - t0) / 1500 * 1e6, 2)))
        print(f"lanes {lanes} effort {s}: (lane, us per call, host us per call) " + " ".join(str(r) for r in row))
g.set_overlap(1)
