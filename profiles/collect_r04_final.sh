#!/bin/bash
# Round 4, final refresh (run ON THE GPU BOX via gpurun from the repo root) after the contact-sweep work: default bench line (with its
# live PMC passes), rocprofv3 kernel stats of the headline and the pixel path, both Kuka kernel variants over the batch sizes, the
# per-step API probes, the phase profile.  Summaries -> gpurun_out/profiles_final/ (copied to profiles/ as r04_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>/dev/null
for w in kuka kuka_pixels; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_k
  timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/kuka_pmc_SQ_WAVES.csv
done
for n in 1024 4096 8192 16384 32768 65536 131072; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
for n in 16384 32768 65536 131072; do for o in 0 1; do
  SRLHIP_KUKA_OCC=$o timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 2>/dev/null | sed "s/^{/{\"occ\": $o, /" >> $OUT/occ_nsweep.jsonl
done; done
cd $R
timeout 300 python profiles/probes/kuka_launch_cost.py > $OUT/kuka_launch_cost.txt 2>&1
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
if [ -f robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so ]; then
  SRLHIP_LIB=$R/robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so timeout 300 python profiles/probes/kuka_tree_phases.py 2048 > $OUT/kuka_tree_phases.txt 2>&1
fi
ls -la $OUT
