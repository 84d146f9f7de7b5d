"""The reference's whole LpSolver through its own KKT solver classes and through the Hip classes on a larger synthetic LP
(tests/test_gpu_lp_dropin.py's generator): times of the phases.  usage: python scripts/gpu_lp_dropin_large.py m n [seed]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp(m, n, seed), crossover=0)
for exe in (T.HIP_BIN, T.REF_BIN):
    t0 = time.time()
    info, _, out = T.run(exe, d + "/in", d + "/out_" + os.path.basename(exe), timeout=1100)
    print(out.strip().splitlines()[0] if out.strip() else "")
    print(os.path.basename(exe), "wall %.1f s" % (time.time() - t0), {k: info[k] for k in ("status_ipm", "iter", "kktiter1", "kktiter2", "updates_ipm", "pobjval", "time_ipm1", "time_ipm2", "time_starting_basis", "time_kkt_factorize", "time_kkt_solve", "time_maxvol", "time_cr2", "lu_factorizations", "lu_max_bump", "lu_device_seconds") + (("device_maxvolume_calls", "cpu_maxvolume_calls") if "device_maxvolume_calls" in info else ())}, flush=True)
