"""Per-operator-application HBM-side traffic of the basis CR iteration from two rocprofv3 --pmc runs (FETCH_SIZE,
WRITE_SIZE) of scripts/gpu_basis_iter.py: sums the counter over EVERY kernel dispatched between the first and the last
split_finish_kernel of the run (the CR loops of its solves: whatever the kernels are called -- round 4 selected them by a list of
names that missed the accumulated-tile products and the block kernels) and divides by the number of operator applications
(= dispatches of split_finish_kernel).  usage: pmc_iteration.py <fetch dir> <write dir> [layout layout]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ipxk::", "").replace("ipxk::", "")
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key: rows.sort(key=lambda r: int(r[key]))
    names = [short(r["Kernel_Name"]) for r in rows]
    marks = [i for i, k in enumerate(names) if "split_finish_kernel" in k]
    if not marks: raise SystemExit("no split_finish_kernel in " + f)
    # the solves of the run are separated by set-up work (Prepare, uploads): keep the dispatches inside windows that begin after a
    # split_finish_kernel and end with the next one no more than 64 dispatches later (one CR iteration is ~25)
    keep = [False] * len(rows)
    for a, b in zip(marks[:-1], marks[1:]):
        if b - a <= 64:
            for i in range(a + 1, b + 1): keep[i] = True
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for i, r in enumerate(rows):
        if keep[i]:
            tot[names[i]] += float(r["Counter_Value"]); cnt[names[i]] += 1
    return tot, cnt
ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
napply = max(c for k, c in fc.items() if "split_finish" in k)
print("operator applications inside the windows: %d" % napply)
read_mb = write_mb = 0.0
for k in sorted(ft, key=lambda k: -ft[k]):
    r = 2.0 * ft[k] * 1024 / napply / 1e6          # KiB, doubled (gfx950 correction, MI355X_MICROARCH.md)
    w = wt.get(k, 0.0) * 1024 / napply / 1e6
    read_mb += r; write_mb += w
    print("%-70s calls/apply %6.2f  read %8.1f MB  write %7.1f MB" % (k[:70], fc[k] / napply, r, w))
print("per application: read %.1f MB + write %.1f MB = %.1f MB" % (read_mb, write_mb, read_mb + write_mb))
tag = os.environ.get("IPXK_PROFILE_TAG", "r05")
json.dump({"workload": "C3 basis path, planted factors: one operator application + CR vector kernels", "traffic_bytes_per_iteration": (read_mb + write_mb) * 1e6,
           "layouts": [a for a in sys.argv[3:5]] if len(sys.argv) > 4 else ["sorted", "sorted"], "source_hashes": bench.source_hashes(),
           "source": "profiles/%s_basis_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on scripts/gpu_basis_iter.py, every kernel between "
                     "consecutive split_finish_kernel dispatches, FETCH x2 gfx950 correction)" % tag},
          open("gpurun_out/pmc_traffic_basis.json", "w"), indent=1)
import shutil
shutil.copy("gpurun_out/pmc_traffic_basis.json", "profiles/pmc_traffic_basis.json")   # a bench.py run in the same call reads it from there
