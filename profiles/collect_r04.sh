#!/bin/bash
# Round 4, run ON THE GPU BOX (via gpurun) from the repo root: the default bench command (headline = KukaButtonGymEnv full model,
# secondary = mobile + kuka_pixels + the lumped model), rocprofv3 --kernel-trace --stats per workload, PMC passes (separate runs,
# --pmc only) over the headline kernel and the pixel path, batch-size sweep, the per-step API probes, config 5 in its single-device
# dry-run form, the phase profile of the tree kernel.  Summaries -> gpurun_out/profiles/ (copied to profiles/ as r04_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) the default command exactly as the driver runs it: one JSON line
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>/dev/null
# (2) kernel stats per workload, each traced alone
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
# (3) PMC: headline kernel
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_k
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/kuka_pmc_${tag}.csv
done
# (4) PMC: pixel path (raster_k, encoder_fwd_k) and MobileRobot
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_p /tmp/pmc_m
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_p -o pmc -- python $R/bench.py --workload kuka_pixels --no-cpu-baseline --no-secondary --no-live-pmc --steps 2 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_p -name "*counter_collection.csv" | head -1) $OUT/kuka_pixels_pmc_${pmc}.csv
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_m -o pmc -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 2 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $OUT/mobile_pmc_${pmc}.csv
done
# (5) N-sweep of the Kuka stepper (full model: the tree lane-group kernel at every size)
for n in 1024 4096 8192 16384 65536; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
# (6) the per-step API: launch cost by rollout length, VecEnv step latency
cd $R
timeout 300 python profiles/probes/kuka_launch_cost.py > $OUT/kuka_launch_cost.txt 2>&1
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
# (7) BASELINE config 5 exactly as stated (raw_pixels, 8 ranks x 4096 envs), dry run of the N > 1 path on this ONE device (gloo)
SRLHIP_SINGLE_DEVICE=1 SRLHIP_DIST_BACKEND=gloo timeout 600 python bench.py --workload kuka_pixels --gpus 8 2>/dev/null | grep "^{" > $OUT/bench_pixels_gpus8_single_device.json
# (8) where a step of the tree kernel goes (profiling build: shader-clock stamps)
if [ -f robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so ]; then
  SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so timeout 300 python profiles/probes/kuka_tree_phases.py 2048 > $OUT/kuka_tree_phases.txt 2>&1
fi
ls -la $OUT
