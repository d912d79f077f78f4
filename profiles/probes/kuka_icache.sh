# Instruction-cache behaviour of the Kuka rollout kernel (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lanes in 64 16; do
rm -rf /tmp/pmc_ic
SRLHIP_KUKA_LANES=$lanes timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d /tmp/pmc_ic -o pmc -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>/tmp/ic_err.log
f=$(find /tmp/pmc_ic -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f /tmp/ic_$lanes.csv; echo "lanes $lanes"; grep kuka_rollout_k /tmp/ic_$lanes.csv; else echo "failed lanes $lanes"; tail -3 /tmp/ic_err.log; fi
done
