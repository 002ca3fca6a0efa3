#!/bin/bash
# round 6 (third session): the thin last calls from 8 calls on (E = 2 launches of 8 .. 10 calls: narrow or mid-size matrices), and a re-scan n = 1 .. 32 with the session's rules
export TMPDIR=/tmp
O=gpurun_out/b49; mkdir -p $O; rm -f $O/scan.txt $O/rescan.txt
for shape in 4096x14336 14336x4096 11008x4096 4096x8192 8192x4096; do
for v in cur new cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --ns 8,9,10 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
unset EFFORT_HIP_LIB
for shape in 4096x11008 4096x14336 14336x4096 4096x4096 4096x1024; do
  timeout 600 python tools/lab/nscan.py --shape $shape 2>&1 | grep -E "us/launch|rror" >> $O/rescan.txt
done
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b49/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),int(m.group(5))),{}).setdefault(m.group(1),[]).append((float(m.group(6)),m.group(7)))
for k,v in d.items():
    print("%-12s n%-2d cur %s  new %s"%(k[0],k[1]," ".join("%6.2f(%s)"%x for x in v['cur'])," ".join("%6.2f(%s)"%x for x in v['new'])))
PY
cat $O/rescan.txt
