#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun; outputs land in gpurun_out/).
#   gpurun --timeout 2400 -- 'ROUND=r03 bash tools/collect_profiles.sh'   then copy gpurun_out/${ROUND}_* into profiles/
set -u
R=${ROUND:-r06}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
H="python bench.py --steps 96 --warmup 16 --headline-only"
rm -rf $O/prof_*
# 1. kernel trace of the timed job (4 launches in flight through one context, every step on its own 32 matrices): per-kernel
#    stats AND the span fold (first start -> last end of the back-to-back launches: what the chip did per launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- $H > $O/prof_bench.json 2> $O/prof_bench.log
cp "$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_bench.csv
BPL=$(python -c "import json;print(json.loads(open('$O/prof_bench.json').read().strip().split('\n')[-1])['bytes_per_launch'])")
python tools/span.py --trace $O/prof_bench --bytes-per-launch $BPL --out $O/${R}_span.json --label "bench.py headline: 4 launches in flight (effort_set_overlap), 4 x 32 disjoint matrices"
# 2. the same with ONE launch in flight: kernels do not overlap, the stats' average duration is the launch duration
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench1 -- $H --streams 1 > $O/prof_bench1.json 2> $O/prof_bench1.log
cp "$(ls -t $O/prof_bench1/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_bench_single_stream.csv
python tools/span.py --trace $O/prof_bench1 --bytes-per-launch $BPL --out $O/${R}_span_single_stream.json --label "one launch in flight"
# 3. round 2's job (every step in flight on the SAME 32 matrices) for the comparison
rocprofv3 --kernel-trace --output-format csv -d $O/prof_shared -- $H --headline-shared > $O/prof_shared.json 2> $O/prof_shared.log
python tools/span.py --trace $O/prof_shared --bytes-per-launch $BPL --out $O/${R}_span_shared_matrices.json --label "4 launches in flight on the SAME 32 matrices (round 2's job)"
rocprofv3 --kernel-trace --output-format csv -d $O/prof_shared_r -- $H --headline-shared --row-reuse > $O/prof_shared_r.json 2> $O/prof_shared_r.log
python tools/span.py --trace $O/prof_shared_r --bytes-per-launch $BPL --out $O/${R}_span_shared_matrices_row_reuse.json --label "4 launches in flight on the SAME 32 matrices, effort_set_row_reuse(1): the ordinary cache policy on the row stream"
rm -rf $O/prof_shared_r
# 4. HBM-side traffic and L2 hit / miss counters (separate --pmc passes, kernel-trace only: MI355X_MICROARCH.md)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- $H > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- $H > /dev/null 2> $O/pmc_write.log
python tools/pmc_traffic.py --fetch $O/prof_fetch --write $O/prof_write --group 32 --effort 0.25 --out $O/${R}_pmc_traffic.json
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_tcc25 -- $H > /dev/null 2> $O/pmc_tcc25.log
python tools/pmc_cache.py --dir $O/prof_tcc25 --out $O/${R}_pmc_tcc_effort25.json --label "effort 0.25, 4 in flight, disjoint matrices"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_tcc10 -- $H --effort 0.1 > /dev/null 2> $O/pmc_tcc10.log
python tools/pmc_cache.py --dir $O/prof_tcc10 --out $O/${R}_pmc_tcc_effort10.json --label "effort 0.10, 4 in flight, disjoint matrices"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_tcc25s -- $H --headline-shared > /dev/null 2> $O/pmc_tcc25s.log
python tools/pmc_cache.py --dir $O/prof_tcc25s --out $O/${R}_pmc_tcc_effort25_shared_matrices.json --label "effort 0.25, 4 in flight on the SAME 32 matrices"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetchs -- $H --headline-shared > /dev/null 2> $O/pmc_fetchs.log
python tools/pmc_traffic.py --fetch $O/prof_fetchs --group 32 --effort 0.25 --out $O/${R}_pmc_traffic_shared_matrices.json
rm -rf $O/prof_bench $O/prof_bench1 $O/prof_shared $O/prof_fetch $O/prof_write $O/prof_tcc25 $O/prof_tcc10 $O/prof_tcc25s $O/prof_fetchs
# 5. the bench line as the driver runs it, and at the default length
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_driver_style.json 2> $O/bench_driver_style.log
tail -c 400 $O/${R}_bench_driver_style.json
cp $O/bench_full.json $O/${R}_bench_full.json
# 5b. what the driver's command no longer runs (tools/bench_extra.py), and the N > 1 leg as a world of one through RCCL
python tools/bench_extra.py --steps 20 --warmup 5 > /dev/null 2> $O/bench_extra.log; cp $O/bench_extra.json $O/${R}_bench_extra.json
BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 > $O/${R}_bench_forced_dist_world1.json 2> $O/bench_forced_dist.log
# 6. decode loop: tokens/s and per-kernel summary
python tools/decode_bench.py --tokens 64 > $O/${R}_decode_bench.json 2> $O/decode.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -- python tools/decode_bench.py --tokens 64 --efforts 0.25 > $O/prof_dec.log 2>&1
cp "$(ls -t $O/prof_dec/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_decode.csv
rm -rf $O/prof_dec
# 7. the cutoff's phases, a layer's multiplies as launches of their own
python tools/cutprof.py 2>&1 | grep "effort" > $O/${R}_cutprof.txt
python tools/layer_probe.py > $O/${R}_layer_probe_effort25.json 2>> $O/probe.log
python tools/layer_probe.py --effort 0.5 > $O/${R}_layer_probe_effort50.json 2>> $O/probe.log
[ -x build/rowbench ] && build/rowbench | grep "^mode" > $O/${R}_rowbench.txt
[ -x build/rampbench ] && build/rampbench > $O/${R}_rampbench.txt
# 8. Q4: kernel stats, ablation table (tools/q4_profile.sh)
ROUND=$R bash tools/q4_profile.sh > $O/q4_profile.log 2>&1
ls -la $O | grep ${R}_
