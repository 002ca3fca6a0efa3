#!/bin/bash
# round 6: Q4 with the outliers merged into the PERSISTENT kernel's streaming loop (shipped library): geometries again, 16 and 32 per launch
export TMPDIR=/tmp
O=gpurun_out/b5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_c_client.py -m gpu -x -q -k "q4 or Q4 or soak or outlier" > $O/pytest_q4.log 2>&1; echo "pytest q4 rc=$?"; grep -E "passed|failed" $O/pytest_q4.log | tail -1
CFG="0,0,0:-1;8,2,8:1;8,2,16:-1;8,2,12:-1;8,2,24:-1;8,1,8:-1;8,1,16:-1;8,1,8:0;8,2,16:0"
timeout 900 python tools/qbench.py --q4 1 --group 16 --reps 2 --configs "$CFG" --tag q4x16 2>&1 | grep -E "rep 1|rror" | tee $O/q4_sweep.txt
timeout 900 python tools/qbench.py --q4 1 --group 32 --reps 2 --configs "0,0,0:-1;8,2,16:-1;8,2,8:-1;8,1,8:-1" --tag q4x32 2>&1 | grep -E "rep 1|rror" | tee -a $O/q4_sweep.txt
timeout 900 python tools/qbench.py --q4 1 --group 16 --reps 2 --overlap 4 --steps-per-graph 8 --configs "0,0,0:-1;8,2,16:-1;8,1,8:-1" --tag q4x16-4inflight 2>&1 | grep -E "rep 1|rror" | tee -a $O/q4_sweep.txt
timeout 900 python tools/qbench.py --q4 1 --group 8 --reps 2 --configs "0,0,0:-1;8,2,16:-1;8,1,16:-1;8,1,32:-1" --tag q4x8 2>&1 | grep -E "rep 1|rror" | tee -a $O/q4_sweep.txt
# the LEAN kernel's prologue in shader-clock stamps (a -DEFFORT_LAB -DEFFORT_CUT_FINE -DEFFORT_LEAN_STAMPS variant): what a kernel-argument preload could take off
EFFORT_HIP_LIB=build/variants/leanstamps.so timeout 300 python tools/lab/cutfine.py 2>&1 | grep -v amdgpu | tee $O/cutfine_lean.txt
