"""Summarise one CR iteration of a rocprofv3 kernel trace:
python trace_iteration.py trace.csv [kernel pattern to list] [control kernel pattern, e.g. 'cr_control_update_kernel<2>']"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X'])) for r in rows)
names = [k[2] for k in ks]
ctl_pat = sys.argv[3] if len(sys.argv) > 3 else 'cr_control'
ctl = [i for i, n in enumerate(names) if ctl_pat in n]
a, b = ctl[len(ctl) // 2], ctl[len(ctl) // 2 + 1]
seq = ks[a:b + 1]
print("iteration span %.1f us, %d kernels" % ((seq[-1][1] - seq[0][1]) / 1e3, len(seq) - 1))
agg = collections.OrderedDict(); prev = seq[0][1]
for s, e, n, g in seq[1:]:
    key = n.split('(')[0][-60:]
    v = agg.setdefault(key, [0, 0.0, 0.0]); v[0] += 1; v[1] += (e - s) / 1e3; v[2] += (s - prev) / 1e3; prev = e
for n, v in agg.items():
    print("%-62s n=%3d busy %8.1f us gaps %7.1f us" % (n, v[0], v[1], v[2]))
if len(sys.argv) > 2 and sys.argv[2]:
    for s, e, n, g in seq[1:]:
        if sys.argv[2] in n: print("  %s grid %d: %.1f us" % (n.split('(')[0][-40:], g, (e - s) / 1e3))
