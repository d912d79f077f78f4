R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mob
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_m
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o m -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --steps 200 --warmup 100 > $OUT/bench_mobile_traced.json 2>/dev/null
cp $(find /tmp/prof_m -name "*kernel_stats.csv" | head -1) $OUT/mobile_kernel_stats.csv
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_m
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_m -o pmc -- python $R/bench.py --workload mobile --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 2 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $OUT/mobile_pmc_${tag}.csv
done
head -5 $OUT/mobile_kernel_stats.csv; cat $OUT/mobile_pmc_*.csv | head -30
