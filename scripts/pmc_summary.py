"""Sums a rocprofv3 --pmc counter per kernel name: usage pmc_summary.py <dir> <COUNTER> [substring of the kernels to list]"""
import collections, csv, glob, sys
d, counter = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != counter: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ipxk::", "").replace("ipxk::", "")
    if sub and sub not in name: continue
    tot[name] += float(r["Counter_Value"]); cnt[name] += 1
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-80s calls %5d  %s per call %.1f" % (k[:80], cnt[k], counter, tot[k] / cnt[k]))
