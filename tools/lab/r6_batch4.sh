#!/bin/bash
# round 6: the GPU suite with the graph-keeping conftest; batch 3; then the NEW bench looped under the checking allocator (hunt 5)
export TMPDIR=/tmp
O=gpurun_out/b4; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; grep -E "passed|failed" $O/pytest.log | tail -2
bash tools/lab/r6_batch3.sh > $O/batch3.log 2>&1; tail -60 $O/batch3.log
rm -rf gpurun_out/hunt; HUNT_KEEP_GOING=1 bash tools/lab/heap_hunt.sh ${HUNT_RUNS:-24} bench > $O/hunt5.log 2>&1; tail -30 $O/hunt5.log
