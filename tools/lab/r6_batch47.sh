#!/bin/bash
# round 6 (third session): Q4 one-item-per-CU rule (3 .. 9 calls) against the library before; the new FP16 / Q4 rules with four launches IN FLIGHT (lanes); order check
export TMPDIR=/tmp
O=gpurun_out/b47; mkdir -p $O; rm -f $O/scan.txt $O/lanes.txt $O/pytest.log
timeout 900 python -m pytest tests -m gpu -x -q -k "q4 or geometry_rules or soak or randomized_groups" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 > $O/pytest.log
NS=1,2,3,4,5,6,7,8,9,10
for shape in 4096x11008 4096x14336 14336x4096 4096x4096; do
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape $shape --ns $NS 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
done
for v in head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --ns 3,4,5,6,7,8,9 --effort 0.5 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --q4 1 --shape 4096x11008 --ns 3,4,5,6,7,8,9 --effort 0.1 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
# order check: the small FP16 shapes with `new` FIRST
for v in new head; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  timeout 600 python tools/lab/nscan.py --shape 4096x1024 --ns 1,2,3,4 --effort 0.3 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  timeout 600 python tools/lab/nscan.py --shape 4096x4096 --ns 1,2,3,4 --effort 0.3 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
done
# lanes: four launches in flight
q() { timeout 300 python tools/qbench.py --reps 2 --overlap 4 --steps-per-graph 4 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/lanes.txt; }
ab() { tag=$1; shift; for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q "$@" --tag $tag-$v; done; }
ab L-n3 --group 3 --mats 48
ab L-n6 --group 6 --mats 48
ab L-n8 --group 8 --mats 64
ab L-sq8 --group 8 --mats 64 --shape 4096x4096
ab L-kv8 --group 8 --mats 64 --shape 4096x1024
ab L-w2n6 --group 6 --mats 48 --shape 14336x4096
cat $O/pytest.log
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b47/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch.*slices (\S+)",l)
    if m: d.setdefault((m.group(2),m.group(3),m.group(4),int(m.group(5))),{})[m.group(1)]=(float(m.group(6)),m.group(7))
for k,v in d.items():
    if 'head' in v and 'new' in v:
        print("%-12s e%-4s q4=%s n%-2d head %7.2f (%s)  new %7.2f (%s)  %+5.1f %%"%(k[0],k[1],k[2],k[3],v['head'][0],v['head'][1],v['new'][0],v['new'][1],(v['new'][0]/v['head'][0]-1)*100))
PY
cat $O/lanes.txt
