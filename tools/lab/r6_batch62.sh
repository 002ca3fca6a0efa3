#!/bin/bash
# round 6 (third session): 3 .. 5 calls of tall narrow matrices at E = 2 when E = 1 overflows one round -- against the library of commit d593c1d (build/variants/cur.so)
export TMPDIR=/tmp
O=gpurun_out/b62; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or experts or layer or fused or decode or column" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
for shape in 11008x4096 14336x4096 8192x4096 4096x4096 13824x5120 14336x1024; do
for v in cur new cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --mats 30 --ns 2,3,4,5,6,7 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape 14336x4096 --mats 30 --ns 3,4,5 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --shape 14336x4096 --mats 30 --ns 3,4,5 --effort 0.1 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b62/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),int(m.group(5))),{}).setdefault(m.group(1),[]).append((float(m.group(6)),m.group(7)))
for k,v in d.items():
    if 'cur' in v and 'new' in v:
        c=sum(x[0] for x in v['cur'])/len(v['cur']); n=sum(x[0] for x in v['new'])/len(v['new'])
        print("%-12s e%-4s n%-2d cur %s (%s)  new %s (%s)  %+5.1f %%"%(k[0],k[1],k[2]," ".join("%6.2f"%x[0] for x in v['cur']),v['cur'][0][1]," ".join("%6.2f"%x[0] for x in v['new']),v['new'][0][1],(n/c-1)*100))
PY
