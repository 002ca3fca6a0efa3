#!/bin/bash
# round 6 (third session): groups of narrow matrices on a context WITH lanes at E = 2 -- against the library of commit 57a248f (build/variants/cur.so), four launches in flight
export TMPDIR=/tmp
O=gpurun_out/b58; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "overlap or lanes or in_flight or soak or timed_configuration or decode" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
for shape in 14336x4096 4096x4096 4096x2048 8192x4096 4096x1024 4096x11008; do
for v in cur new cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --overlap 4 --mats 96 --shape $shape --ns 1,2,3,4,5,6,7,8 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --overlap 4 --mats 96 --shape 14336x4096 --ns 2,3,4,6 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --overlap 4 --mats 96 --shape 4096x4096 --ns 2,3,4,6 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b58/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) lanes 4 n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),int(m.group(5))),{}).setdefault(m.group(1),[]).append((float(m.group(6)),m.group(7)))
for k,v in d.items():
    if 'cur' in v and 'new' in v:
        c=sum(x[0] for x in v['cur'])/len(v['cur']); n=sum(x[0] for x in v['new'])/len(v['new'])
        print("L4 %-12s e%-4s n%-2d cur %s (%s)  new %s (%s)  %+5.1f %%"%(k[0],k[1],k[2]," ".join("%6.2f"%x[0] for x in v['cur']),v['cur'][0][1]," ".join("%6.2f"%x[0] for x in v['new']),v['new'][0][1],(n/c-1)*100))
PY
