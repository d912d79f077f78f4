#!/bin/bash
# Round 6, run ON THE GPU BOX (via gpurun) from the repo root: the default bench command as the driver runs it, rocprofv3 --kernel-trace
# --stats per workload (the 224 x 224 pixel leg included), PMC passes (separate runs, --pmc only) over the headline kernel, the MobileRobot
# rollout and the pixel path, the per-step API probes (single handle and sharded), the fused encoder's phase cycles with its int8 and f16
# layer 1 and the experiment builds, the layered encoder with both layer-1 forms, the GPU suite and smoke().
# Summaries -> gpurun_out/profiles_r06/ (copied to profiles/ as r06_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 1200 python $R/bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
rm -rf /tmp/prof_224
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_224 -o p224 -- python $R/bench.py --workload kuka_pixels --img-size 224 --no-cpu-baseline --no-secondary --no-live-pmc --steps 3 > $OUT/bench_kuka_pixels_224.json 2>/dev/null
cp $(find /tmp/prof_224 -name "*kernel_stats.csv" | head -1) $OUT/kuka_pixels_224_kernel_stats.csv
SRLHIP_ENCODER_L1=i8 timeout 600 python $R/bench.py --workload kuka_pixels --img-size 224 --no-cpu-baseline --no-secondary --no-live-pmc --steps 3 > $OUT/bench_kuka_pixels_224_i8_layer1.json 2>/dev/null
SRLHIP_ENCODER_L1=f16 timeout 600 python $R/bench.py --workload kuka_pixels --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_kuka_pixels_f16_layer1.json 2>/dev/null
for w in kuka mobile; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d" " -f1)
    rm -rf /tmp/pmc_k
    timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
    python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/${w}_pmc_${tag}.csv
  done
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_p
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_p -o pmc -- python $R/bench.py --workload kuka_pixels --no-cpu-baseline --no-secondary --no-live-pmc --steps 2 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_p -name "*counter_collection.csv" | head -1) $OUT/kuka_pixels_pmc_${tag}.csv
done
cd $R
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
timeout 300 python bench.py --per-step > $OUT/bench_per_step.json 2>/dev/null
timeout 300 python bench.py --per-step --envs-per-gpu 2048 --device-ids 0,0 > $OUT/bench_per_step_2x2048_one_device.json 2>/dev/null
timeout 300 python bench.py --per-step --envs-per-gpu 1024 --device-ids 0,0,0,0 > $OUT/bench_per_step_4x1024_one_device.json 2>/dev/null
rm -rf /tmp/prof_ps
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ps -o ps -- python $R/bench.py --per-step > /dev/null 2>&1; cp $(find /tmp/prof_ps -name "*kernel_stats.csv" | head -1) $OUT/per_step_kernel_stats.csv)
timeout 300 python bench.py --per-step --persistent > $OUT/bench_per_step_persistent.json 2>/dev/null
timeout 300 python bench.py --per-step --persistent --envs-per-gpu 256 > $OUT/bench_per_step_persistent_256.json 2>/dev/null
timeout 300 python bench.py --per-step --workload mobile --persistent > $OUT/bench_per_step_mobile_persistent.json 2>/dev/null
timeout 300 python bench.py --per-step --workload mobile --persistent --envs-per-gpu 16 > $OUT/bench_per_step_mobile_persistent_16.json 2>/dev/null
[ -f robotics-rl-srl_amd/csrc/build/libsrlhip_pprof.so ] && SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_pprof.so timeout 200 python profiles/probes/persist_timeline.py 4096 2>&1 | grep -v libdrm > $OUT/persist_timeline.txt
timeout 300 python profiles/encoder_microbench.py > $OUT/encoder_microbench.txt 2>&1
SRLHIP_ENCODER_L1=f16 timeout 300 python profiles/encoder_microbench.py > $OUT/encoder_microbench_f16_layer1.txt 2>&1
for n in 4 5 6; do
  [ -f robotics-rl-srl_amd/csrc/build/libsrlhip_encx$n.so ] && SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_encx$n.so timeout 200 python profiles/probes/encoder_phase_probe.py 2>&1 | grep -v libdrm >> $OUT/encoder_experiments.txt
done
timeout 200 python profiles/probes/encoder_phase_probe.py 2>&1 | grep -v libdrm >> $OUT/encoder_experiments.txt
for n in 1 2 3 4 5 6; do
  [ -f robotics-rl-srl_amd/csrc/build/libsrlhip_egx$n.so ] && SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_egx$n.so timeout 200 python profiles/probes/encoder_general_layer_times.py 2>&1 | grep "ms per" >> $OUT/encoder_general_experiments.txt
done
timeout 200 python profiles/probes/encoder_general_layer_times.py 2>&1 | grep "ms per" >> $OUT/encoder_general_experiments.txt
SRLHIP_SINGLE_DEVICE=1 SRLHIP_DIST_BACKEND=gloo timeout 600 python bench.py --workload kuka_pixels --gpus 8 2>/dev/null | grep "^{" > $OUT/bench_pixels_gpus8_single_device.json
for n in 4096 16384 65536; do
  timeout 300 python bench.py --workload kuka --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
(time timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6) > $OUT/gputests.log 2>&1
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
ls -la $OUT; cut -c1-300 $OUT/bench_default.json; cat $OUT/gputests.log; tail -3 $OUT/smoke.log
