cd /tmp && export TMPDIR=/tmp
for args in "mobile mt" "kuka mt"; do
rm -rf /tmp/mt; rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o mt -- python $GRAFT_REPO_ROOT/profiles/probes/mt_step_trace.py $args > /dev/null 2>&1
f=$(find /tmp/mt -name "*kernel_trace.csv" | head -1)
python - "$f" "$args" <<PY
import csv, sys, statistics
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "rollout" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000 for r in rows]
print(sys.argv[2], "n", len(d), "mean %.1f us" % statistics.mean(d), "median %.1f" % statistics.median(d), "min %.1f max %.1f" % (min(d), max(d)), "p90 %.1f" % sorted(d)[int(0.9*len(d))])
PY
done
