#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do LD_PRELOAD=build/segv_bt.so timeout 200 python tools/lane_crash.py benchjob 2>&1 | grep -v amdgpu | tail -8 | cut -c1-200; done
