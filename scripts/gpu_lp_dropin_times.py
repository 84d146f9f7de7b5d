"""Where the time of KKTSolverBasisHip::Factorize goes under the reference's LpSolver (oracle/_ref/test_lp_hip with
IPXK_VERBOSE=1): the library prints the phases of every Prepare / LU; this script sums them per phase."""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n, seed = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1500, 2500, 34)))
params = dict(crossover=1, dualize=1, switchiter=2) if len(sys.argv) <= 4 else dict(crossover=1)
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp(m, n, seed), **params)
for exe in (T.REF_BIN, T.HIP_BIN):
    os.makedirs(d + "/out", exist_ok=True)
    r = subprocess.run([exe, d + "/in", d + "/out"], capture_output=True, text=True, env=dict(os.environ, IPXK_VERBOSE="1", IPXK_TIME_CPU_PREPARE="1"))
    info = dict(ln.split() for ln in open(d + "/out/info.txt"))
    print(os.path.basename(exe), {k: info[k] for k in ("iter", "kktiter2", "time_ipm2", "time_kkt_factorize", "time_kkt_solve", "time_maxvol", "time_cr2", "time_lu_invert", "time_lu_update", "lu_factorizations", "lu_device_seconds") if k in info},
          {k: info[k] for k in info if k.startswith("cpu_prepare")})
    tot = collections.defaultdict(lambda: [0.0, 0])
    for ln in r.stderr.splitlines():
        if not ln.startswith("ipxk:"): continue
        parts = ln.split(":", 2)
        if len(parts) < 3: continue
        head = re.sub(r"[0-9]+", "#", parts[1].strip())
        for name, val in re.findall(r"([A-Za-z',/ ()]+?) ([0-9.]+) ms", parts[2]):
            t = tot[head + " | " + name.strip(" ,")]; t[0] += float(val); t[1] += 1
    for k, (v, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
        print("   %9.1f ms in %4d  %s" % (v, c, k))
