#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun; outputs land in gpurun_out/).
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'   then copy gpurun_out/${ROUND}_* into profiles/
set -u
R=${ROUND:-r02}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof_bench $O/prof_bench1 $O/prof_dec $O/pmc_fetch $O/pmc_write
# HBM-side traffic of the timed configuration (separate --pmc passes, kernel-trace only: MI355X_MICROARCH.md)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 48 --warmup 8 --headline-only > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 48 --warmup 8 --headline-only > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py --fetch $O/pmc_fetch --write $O/pmc_write --group 32 --effort 0.25 --out $O/${R}_pmc_traffic.json
cp $O/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json          # bench.py reads it back as roofline.traffic
# kernel durations: the timed job (4 launches in flight: each launch lasts ~4x the chip's time per launch) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --steps 200 --warmup 50 --headline-only > $O/prof_bench.log 2>&1
cp "$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_bench.csv
# ... and the same job with ONE launch in flight (kernels do not overlap: the launch duration of roofline.single_stream)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench1 -- python bench.py --steps 200 --warmup 50 --headline-only --streams 1 > $O/prof_bench1.log 2>&1
cp "$(ls -t $O/prof_bench1/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_bench_single_stream.csv
python bench.py > $O/${R}_bench.json 2> $O/bench.log
tail -c 600 $O/${R}_bench.json
python tools/decode_bench.py --tokens 64 > $O/${R}_decode_bench.json 2> $O/decode.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -- python tools/decode_bench.py --tokens 64 --efforts 0.25 > $O/prof_dec.log 2>&1
cp "$(ls -t $O/prof_dec/*/*kernel_stats.csv | head -1)" $O/${R}_rocprofv3_kernel_stats_decode.csv
rm -rf $O/prof_bench $O/prof_bench1 $O/prof_dec $O/pmc_fetch $O/pmc_write
ls -la $O
