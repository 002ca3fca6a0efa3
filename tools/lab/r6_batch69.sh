#!/bin/bash
# round 6 (third session): the session's rules at the EXTREMES of effort (0.02 and 1.0) against the library before the session (build/variants/head.so)
export TMPDIR=/tmp
O=gpurun_out/b69; mkdir -p $O; rm -f $O/scan.txt
for e in 0.02 1.0; do
for shape in 4096x11008 4096x4096 14336x4096; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --effort $e --ns 1,2,3,4,6,7,8,9,11,12,22 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
done
for e in 0.02 1.0; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --effort $e --ns 1,3,4,6,8,16,22,24 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b69/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(4),m.group(2),m.group(3),int(m.group(5))),{})[m.group(1)]=(float(m.group(6)),m.group(7))
for k,v in d.items():
    if 'head' in v and 'new' in v:
        print("q4=%s %-12s e%-4s n%-2d head %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],k[3],v['head'][0],v['head'][1],v['new'][0],v['new'][1],(v['new'][0]/v['head'][0]-1)*100))
PY
