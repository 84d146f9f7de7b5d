"""Condenses rocprofv3 outputs under gpurun_out/ into the tracked summaries in profiles/.
usage: python scripts/make_profile_summary.py <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir> <tag>"""
import collections, csv, glob, json, shutil, statistics, sys
trace_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
one = lambda d, pat: glob.glob("%s/*/*%s" % (d, pat))[0]
rows = list(csv.DictReader(open(one(trace_dir, "kernel_trace.csv"))))
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-banded   (MI355X)",
         "# avg over ALL calls includes the early-exit no-op launches after CR termination; 'working' = duration > 20 us",
         "# for the SpMV kernels (launches that did work)",
         "kernel,calls,total_ms,avg_us_all,calls_working,avg_us_working,median_us_working"]
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    thr = 20000 if "spmv_phased" in k else 0
    w = [x for x in v if x > thr]
    lines.append('"%s",%d,%.3f,%.2f,%d,%.2f,%.2f' % (k.replace('"', "'"), len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, len(w),
                                                (sum(w) / len(w) / 1e3 if w else 0), (statistics.median(w) / 1e3 if w else 0)))
open("profiles/%s_bench_kernel_summary.csv" % tag, "w").write("\n".join(lines) + "\n")
shutil.copy(one(trace_dir, "kernel_stats.csv"), "profiles/%s_bench_kernel_stats_rocprofv3.csv" % tag)
key1 = [k for k in dur if "EpiScale, 8" in k][0]
key2 = [k for k in dur if "EpiNormalRows, 4" in k][0]
p1 = [x for x in dur[key1] if x > 20000]
p2 = [x for x in dur[key2] if x > 20000]
def pmc(d, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
        if r.get("Counter_Name") == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
def pick(agg, key, thr):
    v = [x for x in agg[[k for k in agg if key in k][0]] if x > thr]
    return sum(v) / len(v)
f1, f2 = pick(fe, "EpiScale, 8", 1000), pick(fe, "EpiNormalRows, 4", 1000)
w1, w2 = pick(wr, "EpiScale, 8", 100), pick(wr, "EpiNormalRows, 4", 100)
total = (2 * f1 + w1 + 2 * f2 + w2) * 1024
txt = """# L2<->fabric traffic of the NormalMatrix apply (C3: m=1M, n=2M, nnz=16M), MI355X, %s
# separate passes:  rocprofv3 --pmc FETCH_SIZE -- python3 bench.py ...   and   rocprofv3 --pmc WRITE_SIZE -- ...
# units KiB as reported.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of a coalesced
# streaming read -- calibrated on THIS access pattern with scripts/bench_gather (k<0>: 16M x (4+8) B = 196,608 KiB
# streamed, FETCH_SIZE = 98,310 KiB), so reads are doubled; WRITE_SIZE is exact.  Infinity-Cache hits are included.
kernel,avg_us_working,FETCH_SIZE_KiB_raw,read_MB_corrected,WRITE_SIZE_KiB,write_MB
pass1 spmv_phased_kernel<EpiScale 8> (t = Ws.*(A'y)),%.1f,%.0f,%.1f,%.0f,%.1f
pass2 spmv_phased_kernel<EpiNormalRows 4> (lhs = W_I.*y + A t),%.1f,%.0f,%.1f,%.0f,%.1f
# per apply: %.1f us, traffic = %.1f MB vs algorithmic 476.0 MB (ratio %.2f)
# reads beyond the compulsory bytes are 8-byte gathers that miss the XCD's L2 (each miss moves a whole line)
""" % (tag, sum(p1) / len(p1) / 1e3, f1, 2 * f1 * 1024 / 1e6, w1, w1 * 1024 / 1e6, sum(p2) / len(p2) / 1e3, f2, 2 * f2 * 1024 / 1e6,
       w2, w2 * 1024 / 1e6, (sum(p1) / len(p1) + sum(p2) / len(p2)) / 1e3, total / 1e6, total / 476e6)
open("profiles/%s_pmc_traffic.txt" % tag, "w").write(txt)
json.dump({"workload": "C3 m=1000000 n=2000000 nnz=16000000", "traffic_bytes_per_apply": total,
           "source": "profiles/%s_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)" % tag},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print(txt)
