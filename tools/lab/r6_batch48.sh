#!/bin/bash
# round 6 (third session): the row switch only in the group kernels (persistent, E = 4 plain): do lone calls get head's speed back?  lanes: the new rules off; reuse still works where it lives
export TMPDIR=/tmp
O=gpurun_out/b48; mkdir -p $O; rm -f $O/scan.txt $O/ab.txt
for rep in 1 2 3; do
for v in new head; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  for shape in 4096x1024 4096x4096 4096x11008 4096x14336 14336x4096; do
    timeout 600 python tools/lab/nscan.py --shape $shape --ns 1,2 2>&1 | grep -E "us/launch|rror" | sed "s/^/$v /" >> $O/scan.txt
  done
done
done
q() { timeout 300 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/ab.txt; }
ab() { tag=$1; shift; for v in head new head new; do
  if [ $v = head ]; then export EFFORT_HIP_LIB=build/variants/head.so; else unset EFFORT_HIP_LIB; fi
  q "$@" --tag $tag-$v; done; }
ab L-n3 --group 3 --mats 48 --overlap 4 --steps-per-graph 4
ab L-n6 --group 6 --mats 48 --overlap 4 --steps-per-graph 4
ab L-sq8 --group 8 --mats 64 --shape 4096x4096 --overlap 4 --steps-per-graph 4
ab L-w2n6 --group 6 --mats 48 --shape 14336x4096 --overlap 4 --steps-per-graph 4
unset EFFORT_HIP_LIB
q --group 32 --overlap 4 --steps-per-graph 8 --tag shared4-new
q --group 32 --overlap 4 --steps-per-graph 8 --row-reuse 1 --tag shared4-new-reuse
q --shape 4096x4096 --group 32 --tag sq32-new
q --shape 4096x4096 --group 32 --row-reuse 1 --tag sq32-new-reuse
q --group 32 --tag big1-new
q --group 32 --mats 128 --overlap 4 --steps-per-graph 4 --tag disjoint4-new
python - <<'PY'
import re,collections
d=collections.OrderedDict()
for l in open('gpurun_out/b48/scan.txt'):
    m=re.match(r"(\w+) (\S+) effort (\S+) q4 (\d) n\s+(\d+):\s+([\d.]+) us/launch",l)
    if m: d.setdefault((m.group(2),int(m.group(5))),{}).setdefault(m.group(1),[]).append(float(m.group(6)))
for k,v in d.items():
    h=sum(v['head'])/len(v['head']); n=sum(v['new'])/len(v['new'])
    print("%-12s n%-2d head %s  new %s  %+5.2f us"%(k[0],k[1]," ".join("%6.2f"%x for x in v['head'])," ".join("%6.2f"%x for x in v['new']),n-h))
PY
cat $O/ab.txt
