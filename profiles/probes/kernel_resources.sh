#!/bin/bash
# Register / LDS / spill figures of the kernels in one object file of csrc/build (default: kuka_tree.hip.o), from the code object's
# metadata notes.  usage: profiles/probes/kernel_resources.sh [object] [name-regex]
LIB=${1:-robotics-rl-srl_amd/csrc/build/kuka_tree.hip.o}; FILTER=${2:-.}
TMP=$(mktemp -d); trap 'rm -rf $TMP' EXIT
BUNDLER=/opt/rocm/lib/llvm/bin/clang-offload-bundler
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB
$BUNDLER --list --type=o --input=$TMP/fat.bin 2>/dev/null | grep gfx950 | while read t; do
  $BUNDLER --unbundle --type=o --input=$TMP/fat.bin --targets=$t --output=$TMP/co 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/co 2>/dev/null
done | python3 -c '
import sys, re
name = None; rec = {}
keys = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size")
flt = sys.argv[1]
rows = []
for line in sys.stdin:
    m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and not v.endswith(".kd") and "(" not in v: cur = v
    if k == "symbol": rows.append((v.replace(".kd", ""), dict(rec))); rec = {}
    if k in keys: rec[k] = v
import subprocess
for n, r in rows:
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if re.search(flt, d): print(d[:110], {k.replace("_count", "").replace("_fixed_size", ""): r.get(k) for k in keys})
' "$FILTER"
