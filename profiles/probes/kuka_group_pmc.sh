# PMC passes over the lane-group Kuka rollout kernel (run on the GPU box from the repo root); summaries -> gpurun_out/pmc_group/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_group
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_g -o pmc -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --inner-steps 256 > /dev/null 2>/tmp/pmc_err.log
  f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f $OUT/pass$i.csv; grep "kuka_group_rollout_k" $OUT/pass$i.csv; else echo "pass $i failed"; tail -3 /tmp/pmc_err.log; fi
done
rm -rf /tmp/prof_g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o kuka -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --inner-steps 256 > $OUT/bench.json 2>/dev/null
cp $(find /tmp/prof_g -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; head -5 $OUT/kernel_stats.csv
