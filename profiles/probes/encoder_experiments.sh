#!/bin/bash
# Experiment builds of the fused encoder (csrc/encoder.hip, -DENC_X=<n>): variants that REMOVE one ingredient of a loop (results are
# wrong by construction) to see what that ingredient costs.  Builds robotics-rl-srl_amd/csrc/build/libsrlhip_encx<n>.so for every n
# given; run with SRLHIP_LIB=<that file> python profiles/encoder_microbench.py (phase cycles of workgroup 0).  Never the product.
set -e
cd "$(dirname "$0")/../../robotics-rl-srl_amd/csrc"
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DENC_X=$n -c encoder.hip -o build/encoder_x$n.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsrlhip_encx$n.so $(ls build/*.hip.o build/*.cpp.o | grep -v "encoder.hip.o\|encoder_x\|encoder_general_x\|kuka_tree_prof\|_pprof") build/encoder_x$n.hip.o
done
