#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "launch_geometries or bench_geometry" 2>&1 | tail -3
timeout 300 python tools/qbench.py --group 32 --reps 2 --steps-per-graph 8 --configs "0,0,0:-1;8,6,8:-1;8,6,8:0;8,6,16:-1;8,4,8:0" 2>&1 | grep -v "amdgpu\|Warn"
timeout 300 python tools/qbench.py --group 32 --reps 1 --steps-per-graph 8 --overlap 2 --configs "0,0,0:-1;8,6,8:-1" --tag lanes2 2>&1 | grep -v "amdgpu\|Warn"
timeout 300 python tools/qbench.py --group 16 --reps 1 --steps-per-graph 8 --configs "0,0,0:-1;8,6,16:-1;8,6,8:-1" --tag g16 2>&1 | grep -v "amdgpu\|Warn"
