#!/usr/bin/env python
"""Timing leg of bench.py's cpu_baseline (TEST INFRASTRUCTURE, like everything under oracle/): the CPU port of bucketMul on
the host cores, in a process of its own so that its OpenMP runtime can be configured for throughput -- threads bound to
cores and spinning between the port's parallel regions -- without disturbing the GPU timings of the parent (whose
OpenMP workers must sleep).

    python oracle/cpu_bench.py <dir with b{k}.npy s{k}.npy p{k}.npy v.npy> <inDim> <outDim> <effort> <budget seconds> <nmat>
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    d, inDim, outDim, effort, budget, nmat = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    from oracle import cpu
    mats = [(np.load(os.path.join(d, f"b{k}.npy")), np.load(os.path.join(d, f"s{k}.npy")), np.load(os.path.join(d, f"p{k}.npy"))) for k in range(nmat)]
    v = np.load(os.path.join(d, "v.npy"))
    sc = cpu.Scratch(inDim * 16)
    for k in range(nmat):
        cpu.bucket_mul(v, *mats[k], inDim, outDim, effort, scratch=sc)            # warm: pages, threads, the half table
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        cpu.bucket_mul(v, *mats[n % nmat], inDim, outDim, effort, scratch=sc)
        n += 1
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"calls": n, "seconds_per_call": dt, "omp_threads": int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count(),
                      "omp_wait_policy": os.environ.get("OMP_WAIT_POLICY"), "omp_proc_bind": os.environ.get("OMP_PROC_BIND")}))


if __name__ == "__main__":
    main()
