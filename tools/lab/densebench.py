#!/usr/bin/env python
"""Dense baseline (basicMul = effort_dense_gemv, the package's own HIP GEMV) timed alone: N distinct f16 matrices rotated inside one
hipGraph, us per call and TB/s.  Library under test: $EFFORT_HIP_LIB or the in-tree build.   python tools/lab/densebench.py [--shape 4096x11008]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--mats", type=int, default=32)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    inDim, outDim = (int(x) for x in a.shape.split("x"))
    import effort_amd as ea
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    Ws = [(torch.randn(outDim, inDim, generator=gen, device=dev) * 0.02).to(torch.float16) for _ in range(a.mats)]
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outDim, device=dev) for _ in Ws]

    def run():
        for W, o in zip(Ws, outs):
            ea.basicMul(v, W, o)
    run(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100 / a.mats
        print(f"{a.tag} dense {a.shape} rep {rep}: {dt * 1e6:7.2f} us/call  {inDim * outDim * 2 / dt / 1e12:.2f} TB/s", flush=True)
    ref = (Ws[0].float() @ v.to(torch.float16).float())
    print(f"{a.tag} max rel err vs torch {float((outs[0] - ref).abs().max() / ref.abs().max()):.2e}")
    os._exit(0)      # (graphs are never destroyed: DESIGN.md 5)


if __name__ == "__main__":
    main()
