"""Top kernels of a rocprofv3 results.db (short names).  usage: python scripts/rocprof_top.py <results.db> [n]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
def short(name):
    m = re.search(r"(sp_\w+|lu_\w+|radix_sort_\w+|merge_sort_\w+|scan_impl|init_lookback\w+|__amd_rocclr_\w+|\w+_kernel)", name)
    s = m.group(1) if m else name[:60]
    t = re.search(r"_config<[^,]+, ([\w ]+), ([\w ]+)>", name)
    return s + (" <%s,%s>" % (t.group(1), t.group(2)) if t else "")
agg = {}
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    k = short(name)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += calls; a[1] += total
tot = sum(v[1] for v in agg.values())
print("total kernel time %.1f ms" % (tot / 1e3))
for k, (calls, total) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print("%-60s calls %7d total %9.2f ms avg %8.1f us %5.1f%%" % (k, calls, total / 1e3, total / calls, 100 * total / tot))
