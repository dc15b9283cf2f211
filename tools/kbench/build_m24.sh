#!/bin/bash
# builds tools/kbench/kbench_m24 against the product's objects (run __graft_entry__.build() first)
set -e
cd "$(dirname "$0")"
C=../../compressed_tensors_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-float-cast-overflow -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=off \
  -I../../include -I$C -c kbench_m24.hip -o kbench_m24.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 kbench_m24.o $C/build/ct_api.o -o kbench_m24
rm -f kbench_m24.o
