# PMC passes over the full-model (tree) Kuka rollout kernel (run on the GPU box from the repo root); summaries -> gpurun_out/pmc_tree/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_tree
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_g -o pmc -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --inner-steps 256 > /dev/null 2>/tmp/pmc_err.log
  f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f $OUT/pass$i.csv; grep "kuka_tree_rollout_k" $OUT/pass$i.csv; else echo "pass $i failed"; tail -3 /tmp/pmc_err.log; fi
done
