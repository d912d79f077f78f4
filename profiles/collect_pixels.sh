R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
w=kuka_pixels
rm -rf /tmp/prof_$w
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $R/bench.py --workload $w > $OUT/bench_$w.json 2>/dev/null
cp $(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1) $OUT/${w}_kernel_stats.csv
cut -c1-200 $OUT/bench_$w.json
