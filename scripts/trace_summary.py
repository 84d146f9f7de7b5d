"""Per-kernel summary of a rocprofv3 --kernel-trace csv: calls, total, average; and for the sweep kernels the
durations grouped by launch shape (grid size), which identifies the runs of the launch plan.
usage: python scripts/trace_summary.py <dir with *_kernel_trace.csv> [min_calls]"""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
dur = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ipxk::", "").replace("ipxk::", "")
    key = name
    if "sweep_run_kernel" in name:
        key = "%s grid=%s" % (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-90s %7s %10s %9s %9s" % ("kernel", "calls", "total_ms", "avg_us", "median_us"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-90s %7d %10.3f %9.2f %9.2f" % (k[:90], len(v), sum(v) / 1e3, sum(v) / len(v), v2[len(v2) // 2]))
