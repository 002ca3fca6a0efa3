#!/bin/bash
# round 6 (third session): "the most slices that fit" also when the old rule went over one item per CU -- against the library of commit d9d45e9 (build/variants/cur.so)
export TMPDIR=/tmp
O=gpurun_out/b51; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or experts or layer or fused or decode" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
for shape in 4096x4096 4096x1024 4096x2048 8192x4096 4096x8192 14336x4096 4096x11008; do
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --ns 3,4,5,6,7,8,9,10,11,12 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for shape in 4096x4096 4096x1024; do
for v in cur new; do
  if [ $v = cur ]; then export EFFORT_HIP_LIB=build/variants/cur.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --ns 3,5,9,10,12 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b51/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),int(m.group(5))),{})[m.group(1)]=(float(m.group(6)),m.group(7))
for k,v in d.items():
    if 'cur' in v and 'new' in v:
        print("%-12s e%-4s n%-2d cur %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],v['cur'][0],v['cur'][1],v['new'][0],v['new'][1],(v['new'][0]/v['cur'][0]-1)*100))
PY
