#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O/r3i
rocprofv3 --kernel-trace --output-format csv -d $O/prof_bench -- python bench.py --steps 96 --warmup 16 --headline-only > $O/prof_bench.json 2> $O/prof_bench.log
f=$(ls -t $O/prof_bench/*/*kernel_trace.csv | head -1); head -3 $f; grep -c bucket_mul $f; grep bucket_mul $f | head -400 | tail -3
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "bucket_mul" in r["Kernel_Name"]]
st=sorted(int(r["Start_Timestamp"]) for r in rows)
gaps=[b-a for a,b in zip(st,st[1:])]
import statistics
print(len(rows), "gaps us: median", statistics.median(gaps)/1e3, "max", max(gaps)/1e3, "n>300us", sum(g>300e3 for g in gaps))
print(sorted(gaps)[-12:])
PY
cp $f $O/r3i/kernel_trace_4lanes.csv
