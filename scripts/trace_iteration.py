"""Kernel sequence of ONE basis CR iteration (between two split_finish_kernel launches) from a rocprofv3
--kernel-trace csv.  usage: python scripts/trace_iteration.py <dir> [which]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ipxk::", "").replace("ipxk::", "") for r in rows]
idx = [i for i, n in enumerate(names) if "split_finish" in n]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["End_Timestamp"]); prev = t0
for i in range(a + 1, b + 1):
    r = rows[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-45s grid %7s  start +%8.1f us  dur %7.1f us  gap %5.1f us" % (names[i][:45], r["Grid_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
print("iteration span %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
