"""Kernel time per kernel name inside the LAST occurrence of a window [first launch of <start>, next launch of <stop>)
of a rocprofv3 --kernel-trace csv, plus the window's span (kernel time against wall time = host gaps).
usage: python scripts/trace_window.py <dir> <start substring> <stop substring>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ipxk::", "") for r in rows]
starts = [i for i, n in enumerate(names) if sys.argv[2] in n]
a = starts[-1]
# the window starts at the first launch of the start kernel in its cluster (walk back while the gap is small)
b = next((i for i in range(a + 1, len(rows)) if sys.argv[3] in names[i]), len(rows) - 1)
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["End_Timestamp"])
tot = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for i in range(a, b + 1):
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    tot[names[i][:70]][0] += 1; tot[names[i][:70]][1] += d; busy += d
print("window %s .. %s: span %.2f ms, kernels busy %.2f ms, %d launches" % (names[a], names[b], (t1 - t0) / 1e6, busy / 1e3, b - a + 1))
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-70s %5d %9.1f us" % (k, c, d))
