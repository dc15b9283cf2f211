#!/bin/bash
# builds tools/kbench/kbench_m24 against the product's objects (run __graft_entry__.build() first)
#   build_m24.sh                      the product's ct_marlin24.hip -> kbench_m24
#   build_m24.sh <source.hip> <name>  an experimental copy of the kernel source -> <name> (e.g. under tools/scratch/)
set -e
cd "$(dirname "$0")"
C=../../compressed_tensors_amd/csrc
SRC=${1:+-DCT_M24_SRC=\"$(realpath "$1")\"}
OUT=${2:-kbench_m24}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-float-cast-overflow -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=off \
  -I../../include -I$C $SRC -c kbench_m24.hip -o $OUT.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OUT.o $C/build/ct_api.o -o $OUT
rm -f $OUT.o
