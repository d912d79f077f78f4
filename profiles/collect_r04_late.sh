#!/bin/bash
# Round 4, late refresh (run ON THE GPU BOX via gpurun from the repo root) after the MobileRobot kernel work and the two-wavefront
# Kuka variant: the default bench line again, the MobileRobot kernel stats / PMC passes, the Kuka batch-size sweep with the default
# dispatch (two wavefronts per SIMD from 32768 envs).  Summaries -> gpurun_out/profiles_late/ (copied to profiles/ as r04_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_late
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>/dev/null
rm -rf /tmp/prof_mobile
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mobile -o mobile -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_mobile.json 2>/dev/null
cp $(find /tmp/prof_mobile -name "*kernel_stats.csv" | head -1) $OUT/mobile_kernel_stats.csv
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_m
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_m -o pmc -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 2 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $OUT/mobile_pmc_${tag}.csv
done
for n in 1024 4096 8192 16384 32768 65536 131072; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
for n in 4096 65536 1048576; do
  timeout 300 python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --envs-per-gpu $n >> $OUT/nsweep_mobile.jsonl 2>/dev/null
done
cd $R
timeout 200 python profiles/probes/mobile_chain_probe.py > $OUT/mobile_chain_probe.txt 2>&1
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
ls -la $OUT
