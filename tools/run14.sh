#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 120 python tools/cutprof.py 2>&1 | grep -v amdgpu | head -2
for i in 1 2; do
EFFORT_HIP_LIB=build/variants/before.so timeout 300 python tools/qbench.py --group 1 --reps 1 --tag before 2>&1 | grep -v "amdgpu\|Warn"
timeout 300 python tools/qbench.py --group 1 --reps 1 --tag after 2>&1 | grep -v "amdgpu\|Warn"
done
timeout 300 python tools/decode_ab.py --efforts 0.25 2>/dev/null
