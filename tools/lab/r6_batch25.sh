#!/bin/bash
# round 6 (second session): what bounds a workgroup ALONE on its CU (a lone call's stream: 192 workgroups at ~16 GB/s each; a big launch's tail)?
# build/variants/noscatter.so = the loads without the LDS scatter (lab build, temporal policy) against aux0.so (the product, temporal policy);
# k16.so = 16 rows per batch at E = 4 (nt) against the tree, one workgroup per CU
export TMPDIR=/tmp
O=gpurun_out/b25; mkdir -p $O; rm -f $O/ab.txt
q() { timeout 600 python tools/qbench.py --reps 2 "$@" 2>&1 | grep -E "rep 1|rror" | cut -c1-110 >> $O/ab.txt; }
for v in aux0 noscatter tree k16; do
  case $v in tree) unset EFFORT_HIP_LIB;; *) export EFFORT_HIP_LIB=$PWD/build/variants/$v.so;; esac
  q --group 1 --tag lone-$v
  q --group 1 --effort 1.0 --tag lone100-$v
  q --group 32 --configs "8,4,8:1;0,0,0:-1" --tag g32-$v
  q --group 1 --configs "8,4,32:0;8,4,16:0" --tag loneE4-$v
done
cat $O/ab.txt
