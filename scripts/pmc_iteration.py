"""Per-operator-application HBM-side traffic of the basis CR iteration from two rocprofv3 --pmc runs (FETCH_SIZE,
WRITE_SIZE) of scripts/gpu_basis_iter.py: sums the counter over the kernels of the CR loop and divides by the number
of operator applications (= calls of split_finish_kernel).  usage: pmc_iteration.py <fetch dir> <write dir>"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
KERNELS = ("sweep_run_kernel", "spmv_sliced", "spmv_sorted", "spmv_phased", "spmv_long", "gather_perm_kernel", "fill_sentinel_kernel",
           "split_finish_kernel", "cr_direction_kernel", "cr_control_update_kernel", "snapshot_done_kernel", "unpack_result_kernel")
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ipxk::", "").replace("ipxk::", "")
        if not any(k in name for k in KERNELS): continue
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return tot, cnt
ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
napply = max(c for k, c in fc.items() if "split_finish" in k)
print("operator applications: %d" % napply)
read_mb = write_mb = 0.0
for k in sorted(ft, key=lambda k: -ft[k]):
    r = 2.0 * ft[k] * 1024 / napply / 1e6          # KiB, doubled (gfx950 correction, MI355X_MICROARCH.md)
    w = wt.get(k, 0.0) * 1024 / napply / 1e6
    read_mb += r; write_mb += w
    print("%-70s calls/apply %6.2f  read %8.1f MB  write %7.1f MB" % (k[:70], fc[k] / napply, r, w))
print("per application: read %.1f MB + write %.1f MB = %.1f MB" % (read_mb, write_mb, read_mb + write_mb))
json.dump({"workload": "C3 basis path, planted factors: one operator application + CR vector kernels", "traffic_bytes_per_iteration": (read_mb + write_mb) * 1e6,
           "layouts": [a for a in sys.argv[3:5]] if len(sys.argv) > 4 else ["sorted", "sorted"], "source_hashes": bench.source_hashes(),
           "source": "profiles/r04_basis_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on scripts/gpu_basis_iter.py, FETCH x2 gfx950 correction)"},
          open("gpurun_out/pmc_traffic_basis.json", "w"), indent=1)
import shutil
shutil.copy("gpurun_out/pmc_traffic_basis.json", "profiles/pmc_traffic_basis.json")   # a bench.py run in the same call reads it from there
