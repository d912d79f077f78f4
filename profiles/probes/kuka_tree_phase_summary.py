import sys, re, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"tprof block (\d+) T (\d+) phase (\d+) cycles (\d+)", l)
    if m: acc[int(m.group(3))].append(int(m.group(4)))
    elif l.startswith("launch"): print(l.strip())
n = sum(acc[17]) or 1
T = 2048 * len(acc[17])
print("contact steps counted:", n, "of", T)
for ph, nm in ((12, "candidates->rows"), (13, "W J"), (14, "own row"), (15, "150 sweeps"), (16, "outputs")):
    print("  phase %d %-16s %8.0f cycles per contact step" % (ph, nm, sum(acc[ph]) / n))
free = sum(sum(acc[p]) for p in (0,1,2,3,4,5,6,7,9,10,11,20,21)) / T
print("  all other phases per step: %.0f (free sweeps %.0f)" % (free, sum(acc[7]) / (T - n)))
