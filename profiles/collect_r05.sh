#!/bin/bash
# Round 5, run ON THE GPU BOX (via gpurun) from the repo root: the default bench command as the driver runs it, rocprofv3 --kernel-trace
# --stats per workload, PMC passes (separate runs, --pmc only) over the headline kernel and the pixel path, the per-step API probes in
# steady state, the IK-crossing extent, config 5 in its single-device dry-run form, the phase profile of the tree kernel, the encoder's
# phase cycles, the GPU suite and smoke().  Summaries -> gpurun_out/profiles_r05/ (copied to profiles/ as r05_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_k
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/kuka_pmc_${tag}.csv
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_p
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_p -o pmc -- python $R/bench.py --workload kuka_pixels --no-cpu-baseline --no-secondary --no-live-pmc --steps 2 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_p -name "*counter_collection.csv" | head -1) $OUT/kuka_pixels_pmc_${tag}.csv
done
cd $R
timeout 300 python profiles/probes/kuka_launch_cost.py > $OUT/kuka_launch_cost.txt 2>&1
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
timeout 300 python profiles/probes/ik_crossing_extent.py > $OUT/ik_crossing_extent.json 2>$OUT/ik_crossing_extent.err
timeout 300 python profiles/encoder_microbench.py > $OUT/encoder_microbench.txt 2>&1
SRLHIP_SINGLE_DEVICE=1 SRLHIP_DIST_BACKEND=gloo timeout 600 python bench.py --workload kuka_pixels --gpus 8 2>/dev/null | grep "^{" > $OUT/bench_pixels_gpus8_single_device.json
if [ -f robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so ]; then
  SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so timeout 300 python profiles/probes/kuka_tree_phases.py 2048 > $OUT/kuka_tree_phases.txt 2>&1
fi
(time timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6) > $OUT/gputests.log 2>&1
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
ls -la $OUT; cut -c1-300 $OUT/bench_default.json; cat $OUT/gputests.log; tail -3 $OUT/smoke.log
