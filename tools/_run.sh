export EFFORT_TAIL_CALLS=0
for cfg in "8,4,0 2" "16,4,0 1" "8,4,0 1" "16,2,0 1" "8,2,0 2" "8,4,16 2" "16,4,16 1"; do set -- $cfg; echo "#### tune=$1 persistent=$2"; python tools/timeline.py --groups 32 --tune $1 --persistent $2 --out gpurun_out/tl_x.json 2>&1 | grep -v amdgpu.ids; done
