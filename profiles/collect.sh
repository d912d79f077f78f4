#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats of the default bench commands and
# PMC passes (separate runs, --pmc only) for the dominant kernels; writes small summaries to gpurun_out/profiles/.
# Copy them to profiles/ as rNN_* afterwards (profiles/README.md).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  # the kernel-stats run is the full default command (cpu_baseline leg included)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
# the layered encoder (224x224x3 and 6-channel frames) and the rasteriser on their own
timeout 300 python $R/profiles/encoder_general_microbench.py > $OUT/encoder_general_microbench.jsonl 2>/dev/null
timeout 100 python $R/profiles/raster_microbench.py > $OUT/raster_microbench.txt 2>/dev/null
rm -rf /tmp/prof_eg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eg -o eg -- python $R/profiles/probes/encoder_general_prof.py > /dev/null 2>&1
cp $(find /tmp/prof_eg -name "*kernel_stats.csv" | head -1) $OUT/encoder_general_kernel_stats.csv
[ -n "$SRLHIP_COLLECT_QUICK" ] && { ls -la $OUT; exit 0; }      # the PMC passes and the N-sweeps below only change with the steppers
for w in kuka mobile; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d" " -f1)
    rm -rf /tmp/pmc_$w
    timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_$w -o pmc -- python $R/bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
    python $R/profiles/summarize_pmc.py $(find /tmp/pmc_$w -name "*counter_collection.csv" | head -1) $OUT/${w}_pmc_${tag}.csv
  done
done
# the lane-per-env Kuka kernel at the same size, for the before / after comparison
SRLHIP_KUKA_KERNEL=lane timeout 300 python $R/bench.py --workload kuka --no-cpu-baseline --steps 5 > $OUT/bench_kuka_lane_kernel.json 2>/dev/null
# N-sweep of the Kuka stepper: default kernel choice per batch size
for n in 1024 4096 16384 65536; do
  timeout 300 python $R/bench.py --workload kuka --no-cpu-baseline --envs-per-gpu $n --steps 3 --inner-steps 1024 >> $OUT/nsweep_kuka.jsonl 2>/dev/null
done
for n in 4096 65536 1048576; do
  timeout 300 python $R/bench.py --workload mobile --no-cpu-baseline --envs-per-gpu $n --steps 5 --inner-steps 1024 >> $OUT/nsweep_mobile.jsonl 2>/dev/null
done
ls -la $OUT
