#!/bin/bash
# round 6, first measurement batch: the GPU suite, the slimmed bench + bench_extra, product-only A/B, the split probe, Q4 / lone geometry sweeps
export TMPDIR=/tmp
O=gpurun_out/b1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt; tail -c 600 $O/bench_line.json; cp gpurun_out/bench_full.json $O/bench_full.json
timeout 1200 python tools/bench_extra.py --steps 20 --warmup 5 > $O/bench_extra_line.json 2> $O/bench_extra.err; echo "bench_extra rc=$?" | tee -a $O/rc.txt; cp gpurun_out/bench_extra.json $O/
BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err; echo "forced dist rc=$?" | tee -a $O/rc.txt; tail -c 300 $O/bench_forced_dist.json
timeout 600 python tools/lab/split_probe.py > $O/split_probe.txt 2>&1; cat $O/split_probe.txt
timeout 300 python tools/lab/split_probe.py --uneven 1 --splits 2 >> $O/split_probe.txt 2>&1
bash tools/lab/q4_geosweep.sh > $O/q4geo.log 2>&1; tail -40 $O/q4geo.log
bash tools/lab/lone_geosweep.sh > $O/lonegeo.log 2>&1; tail -50 $O/lonegeo.log
# product-only kernels (GA_TSTAMP / GA_ABLATE / GA_TRACE compiled out of every instantiation) against the tree's
sed -i 's/for rep in 1 2 3; do/for rep in 1 2; do/' tools/lab/ab_big.sh
bash tools/lab/ab_big.sh product > $O/ab_product.log 2>&1; cp gpurun_out/r5b/ab.txt $O/ab_product.txt; cat $O/ab_product.txt
