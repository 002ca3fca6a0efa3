#!/bin/bash
# round 6 (third session): the slice-raising rules stand down under 8 % mean effort -- against the library before the session (head.so) at 2 / 5 / 10 % effort
export TMPDIR=/tmp
O=gpurun_out/b70; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or q4 or selection" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -3 > $O/pytest.log
for e in 0.02 0.05 0.1; do
for shape in 4096x11008 4096x4096 14336x4096; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --effort $e --ns 3,4,6,7,8,9,11,12,22 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --effort $e --ns 3,4,6,8,22,24 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b70/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(4),m.group(2),m.group(3),int(m.group(5))),{})[m.group(1)]=(float(m.group(6)),m.group(7))
for k,v in d.items():
    if 'head' in v and 'new' in v:
        print("q4=%s %-12s e%-4s n%-2d head %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],k[3],v['head'][0],v['head'][1],v['new'][0],v['new'][1],(v['new'][0]/v['head'][0]-1)*100))
PY
