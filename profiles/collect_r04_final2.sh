R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_final5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2>/dev/null
rm -rf /tmp/prof_kuka
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kuka -o kuka -- python $R/bench.py --workload kuka --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_kuka.json 2>/dev/null
cp $(find /tmp/prof_kuka -name "*kernel_stats.csv" | head -1) $OUT/kuka_kernel_stats.csv
cd $R
timeout 300 python profiles/probes/kuka_launch_cost.py > $OUT/kuka_launch_cost.txt 2>&1
timeout 300 python profiles/probes/vecenv_latency.py > $OUT/vecenv_latency.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/gputests.txt
cut -c1-200 $OUT/bench_default.json; head -3 $OUT/kuka_kernel_stats.csv | cut -c1-60,300-420; grep tree $OUT/kuka_launch_cost.txt; tail -2 $OUT/vecenv_latency.txt; cat $OUT/gputests.txt
