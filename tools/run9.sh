#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
for cfg in "0,0,0:-1" "16,2,16:1" "16,2,8:1" "16,1,8:1" "16,2,24:1" "8,2,16:2" "16,2,16:-1" "8,2,12:-1"; do
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 1 --configs "$cfg" --tag "q4 geo"
done 2>&1 | grep -v "amdgpu\|Warn" | tee $O/q4_c.log
for cfg in "0,0,0:-1" "16,2,8:1" "16,2,16:1"; do
  timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 1 --configs "$cfg" --tag "q4 geo"
done 2>&1 | grep -v "amdgpu\|Warn" | tee -a $O/q4_c.log
