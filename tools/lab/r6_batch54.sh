#!/bin/bash
# round 6 (third session): Q4 8 / 9 calls at E = 2: one item per CU -- against the library of commit f0924e7 (build/variants/cur.so)
export TMPDIR=/tmp
O=gpurun_out/b54; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "q4 or geometry_rules or soak" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
for shape in 4096x11008 4096x14336 14336x4096 4096x4096 8192x4096 4096x8192; do
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape $shape --ns 7,8,9,10 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 14336x4096 --ns 8,9 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x14336 --ns 8,9 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b54/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),int(m.group(5))),{})[m.group(1)]=(float(m.group(6)),m.group(7))
for k,v in d.items():
    if 'cur' in v and 'new' in v:
        print("Q4 %-12s e%-4s n%-2d cur %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],v['cur'][0],v['cur'][1],v['new'][0],v['new'][1],(v['new'][0]/v['cur'][0]-1)*100))
PY
