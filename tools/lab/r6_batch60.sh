#!/bin/bash
# round 6 (third session): lone calls, pairs and triples over the matrix shapes of other model families (Llama-13B 5120 / 13824, Llama-70B 8192 / 28672 / 1024, 2048-wide): us per launch
export TMPDIR=/tmp
O=gpurun_out/b60; mkdir -p $O; rm -f $O/scan.txt
for shape in 2048x2048 2048x5632 5632x2048 4096x4096 4096x1024 4096x11008 11008x4096 4096x14336 14336x4096 5120x5120 5120x13824 13824x5120 8192x8192 8192x1024 8192x28672 28672x8192 4096x32000; do
  timeout 300 python tools/lab/nscan.py --shape $shape --ns 1,2,3 --mats 24 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
python - <<'PY'
import re
for l in open('gpurun_out/b60/scan.txt'):
    m=re.match(r"(\d+)x(\d+) effort (\S+) q4 0 n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m:
        i,o,e,n,t,s=int(m.group(1)),int(m.group(2)),float(m.group(3)),int(m.group(4)),float(m.group(5)),m.group(6)
        by=n*(e*i*o*2*1.02)   # ~streamed bytes
        print("%6dx%-6d n%d %8.2f us  slices %-7s  %6.0f GB/s streamed (%.3f of 8 TB/s)"%(i,o,n,t,s,by/t/1e3,by/t/1e3/8000))
    elif 'rror' in l: print(l.strip()[:150])
PY
