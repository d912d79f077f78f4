R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_last
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_k
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_k -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_k -name "*counter_collection.csv" | head -1) $OUT/kuka_pmc_${tag}.csv
done
grep -h "kuka_tree_rollout_k" $OUT/*.csv
