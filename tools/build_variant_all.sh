#!/bin/bash
# tools/build_variant_all.sh NAME "-DFLAG=1 ..." : an A/B build of the WHOLE library with extra flags -> build/variants/NAME.so (tools only)
set -e
cd "$(dirname "$0")/../effort_amd/csrc"
D=../../build/variants/$1.d
mkdir -p $D
for f in api bucket_mul cutoff dispatch convert convert_q4 decode gemv; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I/opt/rocm/include $2 -c $f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build/variants/$1.so $D/*.o -L/opt/rocm/lib -lrocblas -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $D
