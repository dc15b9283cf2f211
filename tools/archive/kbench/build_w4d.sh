#!/bin/bash
# builds tools/kbench/kbench_w4d against the product's objects (run __graft_entry__.build() first)
#   build_w4d.sh [<source.hip> <name>]   an experimental copy of ct_quant.hip instead of the product's
set -e
cd "$(dirname "$0")"
C=../../compressed_tensors_amd/csrc
SRC=${1:+-DCT_QUANT_SRC=\"$(realpath "$1")\"}
OUT=${2:-kbench_w4d}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-float-cast-overflow -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=off \
  -I../../include -I$C $SRC -c kbench_w4d.hip -o $OUT.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OUT.o $C/build/ct_quant_g32_lo.o $C/build/ct_quant_g32_hi.o $C/build/ct_api.o -o $OUT
rm -f $OUT.o
