#!/bin/bash
# Experiment builds of the layered encoder (csrc/encoder_general.hip, -DEG_X=<n>): variants that REMOVE one ingredient of layers 2-3
# (results are wrong by construction) to see what it costs.  1 = no MFMAs in the k-loop, 2 = no pooling / band stores, 3 = no window
# loads.  Builds robotics-rl-srl_amd/csrc/build/libsrlhip_egx<n>.so; run with SRLHIP_LIB=<that file>.  Never the product.
set -e
cd "$(dirname "$0")/../../robotics-rl-srl_amd/csrc"
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DEG_X=$n -c encoder_general.hip -o build/encoder_general_x$n.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsrlhip_egx$n.so $(ls build/*.hip.o build/*.cpp.o | grep -v "encoder_general.hip.o\|encoder_general_x\|encoder_x\|kuka_tree_prof\|_pprof") build/encoder_general_x$n.hip.o
done
