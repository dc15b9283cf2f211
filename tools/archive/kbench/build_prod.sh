#!/bin/bash
# builds tools/kbench/kbench_prod against the product's objects (run __graft_entry__.build() first)
set -e
cd "$(dirname "$0")"
C=../../compressed_tensors_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-float-cast-overflow -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=off \
  -I../../include -I$C -c kbench_prod.hip -o kbench_prod.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 kbench_prod.o $C/build/ct_quant_g32_lo.o $C/build/ct_quant_g32_hi.o $C/build/ct_api.o -o kbench_prod
rm -f kbench_prod.o
