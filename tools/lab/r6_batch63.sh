#!/bin/bash
# round 6 (third session): the EFFORT axis: us per launch against effort for 1 / 3 / 8 / 32 calls per launch (should be smooth and close to linear above the fixed cost)
export TMPDIR=/tmp
O=gpurun_out/b63; mkdir -p $O; rm -f $O/scan.txt
for e in 0.02 0.05 0.1 0.15 0.2 0.25 0.3 0.4 0.5 0.6 0.75 0.9 1.0; do
  timeout 300 python tools/lab/nscan.py --shape 4096x11008 --ns 1,3,8,32 --effort $e 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
for e in 0.05 0.1 0.25 0.5 0.75 1.0; do
  timeout 300 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --ns 1,16 --effort $e 2>&1 | grep -E "us/launch|rror" >> $O/scan.txt
done
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b63/scan.txt'):
    m=re.match(r"(\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(3),int(m.group(4))),[]).append((float(m.group(2)),float(m.group(5)),m.group(6)))
for k,v in d.items():
    print("q4=%s n=%-2d "%k+"  ".join("%g:%.1f"%(e,t) for e,t,s in v))
    # marginal cost per unit effort between consecutive points
    print("        d(us)/d(effort): "+"  ".join("%.0f"%((v[i+1][1]-v[i][1])/(v[i+1][0]-v[i][0])) for i in range(len(v)-1)))
PY
