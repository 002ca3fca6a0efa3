#!/usr/bin/env python
"""Fold a rocprofv3 --kernel-trace of `bench.py --headline-only` into SPAN numbers: with several launches in flight a launch's
own duration says nothing about bandwidth (each lasts about as many times longer as there are launches in flight), but
first start -> last end of a burst of back-to-back launches does.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 200 --warmup 50 --headline-only
    python tools/span.py --trace gpurun_out/trace --bytes-per-launch 748142592 --out profiles/r03_span.json

A burst = consecutive bucket_mul_kernel dispatches whose starts are < --gap-us apart (a graph replay's launches; replays
issued back to back merge into one burst).  Per burst: launches, span, union of the busy intervals, the mean launch duration
and the overlap factor (sum of durations / union).  The summary takes the bursts with at least --min-launches launches.
"""
import argparse
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--bytes-per-launch", type=float, required=True)
    ap.add_argument("--kernel", default="bucket_mul_kernel")
    ap.add_argument("--gap-us", type=float, default=2000.0)
    ap.add_argument("--min-launches", type=int, default=100)
    ap.add_argument("--peak-gbps", type=float, default=8000.0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    rows = []
    for path in glob.glob(os.path.join(a.trace, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if a.kernel in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    if not rows:
        raise SystemExit("no " + a.kernel + " dispatches under " + a.trace)
    rows.sort()
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][0] > a.gap_us * 1e3:
            bursts.append(cur)
            cur = []
        cur.append(r)
    bursts.append(cur)
    out_b = []
    for b in bursts:
        if len(b) < a.min_launches:
            continue
        s0, e1 = min(x[0] for x in b), max(x[1] for x in b)
        union, ce = 0, None
        cs = None
        for s, e, _ in b:
            if ce is None or s > ce:
                if ce is not None:
                    union += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        dur = sum(e - s for s, e, _ in b)
        # launches in flight, time-weighted
        span = e1 - s0
        out_b.append({"launches": len(b), "span_us": round(span / 1e3, 2), "union_busy_us": round(union / 1e3, 2),
                      "mean_launch_duration_us": round(dur / len(b) / 1e3, 2), "launches_in_flight_avg": round(dur / union, 3),
                      "us_per_launch_from_span": round(span / len(b) / 1e3, 3),
                      "achieved_GBps_from_span": round(a.bytes_per_launch * len(b) / span, 1),
                      "frac_of_hbm_peak_from_span": round(a.bytes_per_launch * len(b) / span / a.peak_gbps, 4),
                      "kernel": b[0][2][:90]})
    if not out_b:
        raise SystemExit("no burst with >= %d launches" % a.min_launches)
    tot_l = sum(x["launches"] for x in out_b)
    tot_s = sum(x["span_us"] for x in out_b)
    res = {"label": a.label, "source": "rocprofv3 --kernel-trace (Start_Timestamp / End_Timestamp of every dispatch)", "bytes_per_launch": a.bytes_per_launch,
           "bursts": out_b, "all_bursts": {"launches": tot_l, "us_per_launch_from_span": round(tot_s / tot_l, 3),
                                           "frac_of_hbm_peak_from_span": round(a.bytes_per_launch * tot_l / (tot_s * 1e3) / a.peak_gbps, 4)}}
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["all_bursts"]), json.dumps(out_b[-1]))


if __name__ == "__main__":
    main()
