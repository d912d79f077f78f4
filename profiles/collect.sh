#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats of the default bench commands and
# PMC passes (separate runs, --pmc only) for the dominant kernels; writes small summaries to gpurun_out/profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in kuka mobile kuka_pixels; do
  rm -rf /tmp/prof_$w
  extra=""                                    # the kernel-stats run is the full default command (cpu_baseline leg included)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w $extra > $OUT/bench_$w.json 2>/dev/null
  cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
done
for w in kuka mobile; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d" " -f1)
    rm -rf /tmp/pmc_$w
    timeout 400 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_$w -o pmc -- python $R/bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
    python $R/profiles/summarize_pmc.py $(find /tmp/pmc_$w -name "*counter_collection.csv" | head -1) $OUT/${w}_pmc_${tag}.csv
  done
done
ls -la $OUT
