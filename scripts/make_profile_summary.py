"""Condenses rocprofv3 outputs under gpurun_out/ into the tracked summaries in profiles/.
usage: python scripts/make_profile_summary.py <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir> <tag>"""
import collections, csv, glob, json, os, shutil, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench          # source_hashes(): git blob hashes of the kernel sources these counters belong to
trace_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
one = lambda d, pat: glob.glob("%s/*/*%s" % (d, pat))[0]
rows = list(csv.DictReader(open(one(trace_dir, "kernel_trace.csv"))))
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-banded --no-basis --no-newton --no-other-configs --no-lu --no-maxvolume --no-dropin   (MI355X)",
         "# avg over ALL calls includes the early-exit no-op launches after CR termination; 'working' = duration > 20 us",
         "# for the SpMV kernels (launches that did work)",
         "kernel,calls,total_ms,avg_us_all,calls_working,avg_us_working,median_us_working"]
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    thr = 20000 if ("spmv_phased" in k or "sliced_tile" in k or "sorted_tile" in k or "acc_tile" in k) else (8000 if "sliced_combine" in k else 0)
    w = [x for x in v if x > thr]
    lines.append('"%s",%d,%.3f,%.2f,%d,%.2f,%.2f' % (k.replace('"', "'"), len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, len(w),
                                                (sum(w) / len(w) / 1e3 if w else 0), (statistics.median(w) / 1e3 if w else 0)))
open("profiles/%s_bench_kernel_summary.csv" % tag, "w").write("\n".join(lines) + "\n")
shutil.copy(one(trace_dir, "kernel_stats.csv"), "profiles/%s_bench_kernel_stats_rocprofv3.csv" % tag)
def short(k):
    return k.split("(")[0].replace("void ipxk::", "").replace("ipxk::", "")
# the kernels of one NormalMatrix apply, whichever layout each pass uses
apply_kernels = [k for k in dur if "spmv_" in k and ("EpiScale" in k or "EpiNormalRows" in k) and len(dur[k]) > 60]   # not the few build-time tuning launches
apply_kernels.sort(key=lambda k: ("EpiNormalRows" in k, "combine" in k))
def working(k, v):        # launches after CR termination return at once
    return [x for x in v if x > (8000 if "combine" in k else 20000)]
def pmc(d, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
        if r.get("Counter_Name") == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
def pick(agg, k, thr):
    v = [x for x in agg[k] if x > thr]
    return sum(v) / len(v)
rows_txt, total, total_us = [], 0.0, 0.0
for k in apply_kernels:
    w = working(k, dur[k])
    f, wv = pick(fe, k, 1000), pick(wr, k, 100)
    us = sum(w) / len(w) / 1e3
    total += (2 * f + wv) * 1024
    total_us += us
    rows_txt.append("%s,%.1f,%.0f,%.1f,%.0f,%.1f" % (short(k).replace(",", ""), us, f, 2 * f * 1024 / 1e6, wv, wv * 1024 / 1e6))
layouts = ["acc" if any("acc_tile" in k and e in k for k in apply_kernels) else
           "sorted" if any("sorted_tile" in k and e in k for k in apply_kernels) else
           "sliced" if any("sliced" in k and e in k for k in apply_kernels) else "phased" for e in ("EpiScale", "EpiNormalRows")]
txt = """# L2<->fabric traffic of the NormalMatrix apply (C3: m=1M, n=2M, nnz=16M), MI355X, %s
# separate passes:  rocprofv3 --pmc FETCH_SIZE -- python3 bench.py ...   and   rocprofv3 --pmc WRITE_SIZE -- ...
# units KiB as reported.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of a coalesced
# streaming read -- calibrated on THIS access pattern with scripts/bench_gather (k<0>: 16M x (4+8) B = 196,608 KiB
# streamed, FETCH_SIZE = 98,310 KiB), so reads are doubled; WRITE_SIZE is exact.  Infinity-Cache hits are included.
# pass 1 (t = Ws.*(A'y)): the EpiScale kernels; pass 2 (lhs = W_I.*y + A t): the EpiNormalRows kernels
kernel,avg_us_working,FETCH_SIZE_KiB_raw,read_MB_corrected,WRITE_SIZE_KiB,write_MB
%s
# per apply: %.1f us of kernel time, traffic = %.1f MB vs algorithmic 476.0 MB (ratio %.2f)
""" % (tag, "\n".join(rows_txt), total_us, total / 1e6, total / 476e6)
open("profiles/%s_pmc_traffic.txt" % tag, "w").write(txt)
json.dump({"workload": "C3 m=1000000 n=2000000 nnz=16000000", "layouts": layouts, "traffic_bytes_per_apply": total,
           "source_hashes": bench.source_hashes(),
           "source": "profiles/%s_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)" % tag},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print(txt)
