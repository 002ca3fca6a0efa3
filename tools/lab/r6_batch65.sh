#!/bin/bash
# round 6 (third session): the thin last calls counted by the stragglers, FP16 and Q4 -- against the library of commit d7d760b (build/variants/cur.so)
export TMPDIR=/tmp
O=gpurun_out/b65; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or q4 or aligned_row" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -6 > $O/pytest.log
run() { for v in cur new cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py "$@" 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done; }
run --q4 1 --shape 4096x11008 --ns 17,18,20,22,23,24,25,26,28,32
run --q4 1 --shape 4096x14336 --ns 14,16,17,18,19,20,24
run --q4 1 --shape 14336x4096 --ns 17,18,20,24
run --q4 1 --shape 4096x11008 --ns 22,24 --effort 0.5
run --shape 4096x11008 --ns 11,12,13,22,23,24
run --shape 4096x14336 --ns 10,11,19,20
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b65/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(4),m.group(2),m.group(3),int(m.group(5))),{}).setdefault(m.group(1),[]).append((float(m.group(6)),m.group(7)))
for k,v in d.items():
    if 'cur' in v and 'new' in v:
        c=sum(x[0] for x in v['cur'])/len(v['cur']); n=sum(x[0] for x in v['new'])/len(v['new'])
        print("q4=%s %-12s e%-4s n%-2d cur %s (%s)  new %s (%s)  %+5.1f %%"%(k[0],k[1],k[2],k[3]," ".join("%6.2f"%x[0] for x in v['cur']),v['cur'][0][1]," ".join("%6.2f"%x[0] for x in v['new']),v['new'][0][1],(n/c-1)*100))
PY
