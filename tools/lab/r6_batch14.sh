#!/bin/bash
# round 6: the dense item numbering for slice counts that are not multiples of 8 (no padding blocks interleaved): suite + Q4 by group size + the bench's Q4 lines
export TMPDIR=/tmp
O=gpurun_out/b14; mkdir -p $O; rm -f $O/sweep.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
q() { timeout 400 python tools/qbench.py --q4 1 --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-100 >> $O/sweep.txt; }
for g in 10 12 14 16; do q --group $g --tag q4x$g; done
q --group 16 --configs "8,1,5:0;8,1,6:0;8,1,7:0;8,2,10:0;8,2,8:0" --tag q4x16-forced
q --group 12 --configs "8,1,6:0;8,1,7:0" --tag q4x12-forced
cat $O/sweep.txt
export TMPDIR=/tmp
rm -rf $O/prof_q4; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q4 -- timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 1 --tag prof > $O/prof_q4.log 2>&1
head -3 "$(ls -t $O/prof_q4/*/*kernel_stats.csv | head -1)" | cut -c1-160
