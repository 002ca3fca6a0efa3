python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for nj in 1 0; do
  if [ $nj = 1 ]; then export EFFORT_NO_CUTJOBS=1; else unset EFFORT_NO_CUTJOBS; fi
  echo "== no_cutjobs=$nj"
  python tools/tune.py --mats 32 --groups 16,32 --configs "0,0,0" --efforts 0.25,0.5,0.9 2>&1 | grep -o '"effort": [0-9.]*\|"g[0-9]*": [0-9.]*\|g32_kernel_us": [0-9.]*\|g32_wgmean": [^]]*' | paste - - - - -
  python tools/tune.py --q4 1 --mats 32 --groups 32 --configs "0,0,0" --efforts 0.25 2>&1 | grep -o '"effort": [0-9.]*\|"g[0-9]*": [0-9.]*\|g32_kernel_us": [0-9.]*' | paste - - -
  python tools/tune.py --shape 4096x4096 --mats 32 --groups 32 --configs "0,0,0" --efforts 0.25,0.5 2>&1 | grep -o '"effort": [0-9.]*\|"g[0-9]*": [0-9.]*\|g32_kernel_us": [0-9.]*' | paste - - -
done
