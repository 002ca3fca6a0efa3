export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -5 > $O/pytest.log
for rep in 1 2; do
for v in main new; do
  if [ $v = main ]; then export EFFORT_HIP_LIB=build/variants/main.so; else unset EFFORT_HIP_LIB; fi
  echo "== $v rep $rep" >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 1 --reps 3 --tag lone-$v >> $O/ab.txt 2>&1
  timeout 200 python tools/qbench.py --group 3 --reps 3 --tag three-$v >> $O/ab.txt 2>&1
  timeout 200 python tools/cutprof.py 2>&1 | grep "11008) effort 0.25" >> $O/ab.txt
  timeout 200 python tools/qbench.py --group 32 --reps 2 --overlap 4 --steps-per-graph 8 --tag big-$v >> $O/ab.txt 2>&1
  timeout 200 python tools/qbench.py --shape 4096x4096 --group 32 --reps 2 --tag sq32-$v >> $O/ab.txt 2>&1
  timeout 200 python tools/qbench.py --q4 1 --group 16 --reps 2 --tag q4x16-$v >> $O/ab.txt 2>&1
  timeout 200 python tools/qbench.py --q4 1 --group 1 --reps 2 --tag q4lone-$v >> $O/ab.txt 2>&1
done
done
for v in main new main new; do
  if [ $v = main ]; then export EFFORT_HIP_LIB=build/variants/main.so; else unset EFFORT_HIP_LIB; fi
  echo "== decode $v" >> $O/ab.txt
  timeout 400 python tools/decode_bench.py --efforts 0.25 --tokens 64 2>&1 | tail -1 >> $O/ab.txt
done
unset EFFORT_HIP_LIB
echo "== layer_probe new" >> $O/ab.txt; timeout 300 python tools/layer_probe.py 2>&1 | tail -3 >> $O/ab.txt
export EFFORT_HIP_LIB=build/variants/main.so
echo "== layer_probe main" >> $O/ab.txt; timeout 300 python tools/layer_probe.py 2>&1 | tail -3 >> $O/ab.txt
