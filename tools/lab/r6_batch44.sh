#!/bin/bash
# round 6 (third session): the one-round rules (pick_slices: 15/16 of the CUs for groups of >= 3; pick_elems: E = 4 for 3..7 calls whose E = 2 items overflow a round) against
# the library without them: n = 1 .. 12 calls per launch x shapes, alternating
export TMPDIR=/tmp
O=gpurun_out/b44; mkdir -p $O; rm -f $O/scan.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "geometry_rules or group_launch or randomized_groups or launch_geometries or soak or experts or layer or fused" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
NS=1,2,3,4,5,6,7,8,9,10,11,12
for shape in 4096x11008 4096x14336 14336x4096 11008x4096 4096x4096 4096x1024; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --ns $NS 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for shape in 4096x11008 14336x4096; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape $shape --ns 3,4,5,6,7,8 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --shape $shape --ns 3,4,5,6,7,8 --effort 0.1 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b44/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 0 n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),int(m.group(4))),{})[m.group(1)]=(float(m.group(5)),m.group(6))
for k,v in d.items():
    if 'head' in v and 'new' in v:
        print("%-12s e%-4s n%-2d head %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],v['head'][0],v['head'][1],v['new'][0],v['new'][1],(v['new'][0]/v['head'][0]-1)*100))
PY
