#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." : an A/B build of the multiply kernel -> build/variants/NAME.so (tools only).
# Lab switches (EFFORT_PAD_TEST, EFFORT_CUT_FINE, EFFORT_ABLATE_*, ...) need -DEFFORT_LAB among the flags; without it the variant is a
# product build (no stamps), linked against the tree's product api.o -- with it, against lab_api.o.
set -e
cd "$(dirname "$0")/../effort_amd/csrc"
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I/opt/rocm/include $2 -c bucket_mul.hip -o ../../build/variants/$1_bm.o
API=api.o; case "$2" in *EFFORT_LAB*) API=lab_api.o;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build/variants/$1.so $API ../../build/variants/$1_bm.o cutoff.o dispatch.o convert.o convert_q4.o decode.o gemv.o -L/opt/rocm/lib -lrocblas -ldl -Wl,-rpath,/opt/rocm/lib
rm -f ../../build/variants/$1_bm.o
