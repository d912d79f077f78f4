"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel sums / per-dispatch averages.
usage: summarize_pmc.py <counter_collection.csv> <out.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in rows:
    m = re.search(r"(\w+_k)\b", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"].split("<")[0].split("(")[0][-40:]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
    for name in sorted(agg):
        for c, v in sorted(agg[name].items()):
            n = len(disp[name])
            f.write("{},{},{},{:.6g},{:.6g}\n".format(name, c, n, v, v / max(n, 1)))
