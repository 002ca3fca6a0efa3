#!/usr/bin/env python
"""Round 6: what takes a process down after hipGraph captures that fork to other streams (the `free(): invalid pointer` of
round 5's bench, reproduced by round 6's hunts in bench.py's four-stream sections -- SIGSEGV inside hipGraphLaunch, or the heap found
corrupted a few frees later).  The loop here is the smallest thing that has the pattern: capture a graph in which the capturing stream
forks to three other streams and joins them again, replay it, destroy it; thousands of times, one variant per process.

    --work effort|torch      what runs on the streams: the library's dense GEMV (effort_dense_gemv, four contexts) or a plain torch op
    --sync wait_stream|events
          wait_stream: torch's Stream.wait_stream -- each call creates a temporary event, records it, makes the other stream wait for
                       it and DESTROYS it at once, while the capture it took part in is still open (what bench.Job did until round 6)
          events:      the same edges through events that live as long as the process (what the library's own lanes have always
                       done, api.hip's pools, after tools/lab/lane_crash.py found the same crash for destroyed lane events in round 3)
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="torch")
    ap.add_argument("--sync", default="wait_stream")
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--keep", type=int, default=0, help="1: never destroy a graph (every one is kept until the process exits)")
    ap.add_argument("--streams", type=int, default=4, help="1: no fork / join at all (everything on the capturing stream)")
    ap.add_argument("--nograph", type=int, default=0, help="1: the same work eagerly, no capture at all")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    S = max(1, a.streams)
    streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    fork_ev = torch.cuda.Event()
    join_ev = [torch.cuda.Event() for _ in range(S)]
    x = [torch.randn(1 << 16, device=dev) for _ in range(S)]
    y = [torch.zeros(1 << 16, device=dev) for _ in range(S)]
    if a.work == "effort":
        import effort_amd as ea
        ctxs = [ea.gpu(0)] + [ea.Gpu(0) for _ in range(S - 1)]
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        cores = [(torch.randn((4096, 4096), generator=gen, device=dev) * 0.02).to(torch.float16) for _ in range(4)]
        v = torch.randn(4096, generator=gen, device=dev)
        outs = [torch.zeros((4, 4096), device=dev) for _ in range(S)]

    def work(k):
        if a.work == "effort":
            for j, W in enumerate(cores):
                ea.basicMul(v, W, outs[k][j], gpu=ctxs[k])
        else:
            y[k].add_(x[k])
            y[k].mul_(0.5)

    def enqueue(n):
        s0 = torch.cuda.current_stream()
        used = min(S, n)
        if a.sync == "wait_stream":
            for k in range(1, used):
                streams[k].wait_stream(s0)
        else:
            fork_ev.record(s0)
            for k in range(1, used):
                streams[k].wait_event(fork_ev)
        for i in range(n):
            k = i % S
            if k == 0:
                work(0)
            else:
                with torch.cuda.stream(streams[k]):
                    work(k)
        for k in range(1, used):
            if a.sync == "wait_stream":
                s0.wait_stream(streams[k])
            else:
                join_ev[k].record(streams[k])
                s0.wait_event(join_ev[k])
    t0 = time.time()
    kept = []
    tag = f"{a.work} {a.sync} streams={S} keep={a.keep} nograph={a.nograph}"
    for it in range(a.iters):
        enqueue(S)
        torch.cuda.synchronize()
        if a.nograph:
            for _ in range(5):
                enqueue(a.steps)
            torch.cuda.synchronize()
            continue
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            enqueue(a.steps)
        if a.work == "effort":
            for c in ctxs:
                c._bind_stream()
        for _ in range(4):
            g.replay()
        torch.cuda.synchronize()
        if a.keep:
            kept.append(g)
        del g
        if it % 500 == 499:
            print(f"{tag}: {it + 1} captures + replays{'' if a.keep else ' + destroys'}, {time.time() - t0:.0f} s", flush=True)
    print(f"{tag}: done, {a.iters} iterations clean", flush=True)
    sys.stdout.flush()
    if a.keep:
        os._exit(0)          # (the kept graphs are not destroyed at interpreter exit either)


if __name__ == "__main__":
    main()
