#!/usr/bin/env python
"""Turn rocprofv3 --pmc passes of `bench.py` into profiles/<round>_pmc_traffic.json (read back by bench.py as
roofline.traffic).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 50 --warmup 10 --headline-only
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 50 --warmup 10 --headline-only
    python tools/pmc_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --group 32 --effort 0.25 --out profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch, derived
from the L2's fabric-side request counters; on gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes, so it is
DOUBLED here.  WRITE_SIZE is uncalibrated in the guide and is added as reported (it is < 3 % of the traffic of this
kernel).  Only bucket_mul_kernel dispatches of the grouped launches are averaged (the largest grid of that name).
"""
import argparse
import csv
import glob
import json
import os
import statistics


def per_dispatch(dirname, counter):
    vals = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter or "bucket_mul_kernel" not in row["Kernel_Name"]:
                    continue
                key = (path, row["Dispatch_Id"])
                vals[key] = (vals.get(key, (0.0, 0))[0] + float(row["Counter_Value"]), int(row["Grid_Size"]))
    return list(vals.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", default=None)
    ap.add_argument("--group", type=int, required=True)
    ap.add_argument("--effort", type=float, required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f = per_dispatch(a.fetch, "FETCH_SIZE")
    if not f:
        raise SystemExit("no FETCH_SIZE rows for bucket_mul_kernel under " + a.fetch)
    grid = statistics.mode(g for _, g in f)                     # --headline-only: all launches are the timed configuration
    fk = [v for v, g in f if g == grid]
    fetch_kib = statistics.mean(fk)
    write_kib = None
    if a.write:
        w = [v for v, g in per_dispatch(a.write, "WRITE_SIZE") if g == grid]
        write_kib = statistics.mean(w) if w else None
    out = {
        "calls_per_launch": a.group, "effort": a.effort, "grid_size_threads": grid, "dispatches_averaged": len(fk),
        "FETCH_SIZE_KiB_reported": round(fetch_kib, 1), "WRITE_SIZE_KiB_reported": None if write_kib is None else round(write_kib, 1),
        "read_bytes_per_launch": int(fetch_kib * 1024 * 2),
        "write_bytes_per_launch": None if write_kib is None else int(write_kib * 1024),
        "hbm_bytes_per_launch": int(fetch_kib * 1024 * 2 + (write_kib or 0.0) * 1024),
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as reported (uncalibrated)",
    }
    with open(a.out, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
