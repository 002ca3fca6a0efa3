#!/bin/bash
# Q4 evidence for profiles/ (VERDICT round 3, item 5): run on a GPU box from the repo root AFTER building the lab variants HERE
#   tools/build_variant.sh noscatter "-DEFFORT_LAB -DEFFORT_ABLATE_NOSCATTER=1"     (the EFFORT_ABLATE switches live in the in-tree lab library: EFFORT_HIP_LIB=lab)
#   gpurun --timeout 900 -- 'ROUND=r04 bash tools/q4_profile.sh'      then copy gpurun_out/${ROUND}_q4_* into profiles/
set -u
R=${ROUND:-r06}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
Q="timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 1"
us() { grep "us/launch" | sed -E 's/.*: +([0-9.]+) us\/launch.*/\1/' | tail -1; }
# 1. rocprofv3 kernel stats of the 16-per-launch Q4 job at 25 % effort (BASELINE config 3, 16 calls per launch)
rm -rf $O/prof_q4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q4 -- $Q --tag prof > $O/prof_q4.log 2>&1
cp "$(ls -t $O/prof_q4/*/*kernel_stats.csv | head -1)" $O/${R}_q4_rocprofv3_kernel_stats_16_per_launch.csv
rm -rf $O/prof_q4
# 2. the ablation table (LAB builds read EFFORT_ABLATE; the shipped library does not)
WHOLE=$($Q --tag whole 2>&1 | us)
LONE=$(timeout 200 python tools/qbench.py --q4 1 --group 1 --reps 1 --tag lone 2>&1 | us)
G32=$(timeout 200 python tools/qbench.py --q4 1 --group 32 --reps 1 --tag g32 2>&1 | us)
NOOL=$($Q --no-outliers 1 --tag no-outliers 2>&1 | us)
NOSTREAM=$(EFFORT_HIP_LIB=lab EFFORT_ABLATE=4 $Q --tag no-stream 2>&1 | us)
NEITHER=$(EFFORT_HIP_LIB=lab EFFORT_ABLATE=4 $Q --no-outliers 1 --tag neither 2>&1 | us)
NOSCATTER=$(EFFORT_HIP_LIB=build/variants/noscatter.so $Q --tag no-scatter 2>&1 | us)
# 3. the LDS atomic rate the streaming phase is bound by
[ -x tools/microbench ] && tools/microbench 2>&1 | grep -E "^device|^scatter" > $O/${R}_q4_microbench_scatter.txt
DSADD=$(grep "ds_add_u32" $O/${R}_q4_microbench_scatter.txt | head -1 | sed -E 's/.* ([0-9.]+) elem\/ns\/CU.*/\1/')
python - <<PY > $O/${R}_q4_ablation.json
import json
w, lone, g32, nool, nostream, neither, noscatter, dsadd = [float(x) if x else None for x in "$WHOLE|$LONE|$G32|$NOOL|$NOSTREAM|$NEITHER|$NOSCATTER|$DSADD".split("|")]
print(json.dumps({
  "what": "bucketMulQ4 4096x11008 at 25 % effort with the converter's 2 % outlier tables, 16 calls per launch unless said; us per LAUNCH (hipGraph replays over 32 rotating matrices, tools/qbench.py)",
  "whole": w, "lone_call": lone, "us_per_call_16_per_launch": None if w is None else round(w / 16, 3), "us_per_launch_32_per_launch": g32,
  "without_outliers": nool, "without_streaming(EFFORT_ABLATE=4, lab build)": nostream, "neither": neither,
  "loads_without_the_lds_scatter(EFFORT_ABLATE_NOSCATTER build)": noscatter,
  "ds_add_u32_elements_per_ns_per_CU(tools/microbench)": dsadd,
  "reading": "whole - loads_without_scatter = what the four ds_add per 16-bit word cost beyond the loads; whole - without_outliers = the outlier phase; neither = staging, cutoff, selection, hand-off, reduction"}, indent=1))
PY
cat $O/${R}_q4_ablation.json
ls -la $O | grep ${R}_q4
