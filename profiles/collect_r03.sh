#!/bin/bash
# Round 3, run ON THE GPU BOX (via gpurun) from the repo root: the default bench command under rocprofv3 --kernel-trace --stats
# (headline = KukaButtonGymEnv full model, secondary = mobile + kuka_pixels), PMC passes (separate runs, --pmc only) over the
# headline kernel, batch-size sweep of the Kuka stepper.  Summaries -> gpurun_out/profiles/ (copy to profiles/ as r03_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) the default command exactly as the driver runs it: headline + secondary legs + cpu baselines, one JSON line
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>/dev/null
# (2) kernel stats per workload, each traced alone so that a kernel's average launch duration is that of ONE launch shape
#     (the default command mixes 2048-step rollouts and single steps of the same kernel)
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
# the reference-exact MT19937 streams on the same workload
timeout 300 python $R/bench.py --rng mt19937 --no-cpu-baseline --no-secondary --no-live-pmc --steps 5 > $OUT/bench_kuka_mt19937.json 2>/dev/null
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_k
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/kuka_pmc_${tag}.csv
done
# MobileRobot: HBM traffic of the episode-parallel rollout (its spare workgroups now also write the next action plane)
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_m
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_m -o pmc -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 2 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $OUT/mobile_pmc_${pmc}.csv
done
# the lumped-gripper model (rounds 1-2) on its kernels, for the before / after comparison
timeout 300 python $R/bench.py --kuka-model lumped --no-cpu-baseline --no-secondary --no-live-pmc --steps 5 > $OUT/bench_kuka_lumped.json 2>/dev/null
# N-sweep of the Kuka stepper (full model: the tree lane-group kernel at every size)
for n in 1024 4096 8192 16384 65536; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
# the N > 1 launch form on this one device (gloo instead of RCCL): bench.py --gpus 2 starts its own ranks
SRLHIP_SINGLE_DEVICE=1 SRLHIP_DIST_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 3 --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_gpus2_single_device.json
ls -la $OUT
