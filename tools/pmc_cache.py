#!/usr/bin/env python
"""Fold a rocprofv3 --pmc pass with L2 (TCC) hit / miss counters into per-launch averages for the multiply kernel.

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -- python bench.py --steps 48 --warmup 8 --headline-only
    python tools/pmc_cache.py --dir gpurun_out/pmc_tcc --out profiles/r03_pmc_tcc_effort25.json --label "effort 0.25"
"""
import argparse
import csv
import glob
import json
import os
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    per = {}
    for path in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if "bucket_mul_kernel" not in r["Kernel_Name"]:
                    continue
                key = (path, r["Dispatch_Id"])
                d = per.setdefault(key, {"grid": int(r["Grid_Size"])})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not per:
        raise SystemExit("no bucket_mul_kernel counter rows under " + a.dir)
    grid = statistics.mode(d["grid"] for d in per.values())
    sel = [d for d in per.values() if d["grid"] == grid]
    names = sorted({k for d in sel for k in d if k != "grid"})
    res = {"label": a.label, "dispatches_averaged": len(sel), "grid_size_threads": grid}
    for n in names:
        res[n + "_per_launch"] = round(statistics.mean(d.get(n, 0.0) for d in sel), 1)
    if "TCC_HIT_sum" in names and "TCC_MISS_sum" in names:
        h, m = res["TCC_HIT_sum_per_launch"], res["TCC_MISS_sum_per_launch"]
        res["L2_hit_rate"] = round(h / (h + m), 4) if h + m else None
        res["L2_miss_bytes_at_128B_per_launch"] = int(m * 128)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
