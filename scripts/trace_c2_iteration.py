"""Kernel sequence of ONE preconditioned-CR iteration of a diag solve (between two cr_control_update launches) from a rocprofv3
--kernel-trace csv: start, run time and the gap to the previous kernel's end -- which of the chip-wide dependencies of an
iteration (src/conjugate_residuals.cc:129-211) costs what.  usage: python scripts/trace_c2_iteration.py <dir> [which]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ipxk::", "").replace("ipxk::", "") for r in rows]
idx = [i for i, n in enumerate(names) if "cr_control_update" in n]
for w in (which, which + 1, which + 2):
    a, b = idx[w], idx[w + 1]
    t0 = int(rows[a]["Start_Timestamp"]); prev = None
    run = 0.0
    for i in range(a, b):
        r = rows[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        run += (e - s) / 1e3
        print("%-62s grid %6s x %4s  start +%6.1f us  run %5.1f us  gap before %4.1f us" % (names[i][:62], r["Grid_Size_X"], r["Workgroup_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev = e
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    print("iteration: %.1f us from control kernel to control kernel, %.1f us inside kernels, %.1f us between them\n" % (span, run, span - run))
