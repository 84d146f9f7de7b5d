"""The north star's own sentence at the benchmark size: the reference's LpSolver (presolve, starting point, initial IPM with
the diagonally preconditioned KKT solves) on a 1M x 2M synthetic LP, through the reference's KKTSolverDiag (CPU) and through
KKTSolverDiagHip (MI355X); stop_at_switch = 1 and a small ipm_maxiter keep the CPU run within minutes.
usage: python scripts/gpu_lp_dropin_c3.py [m n ipm_maxiter]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n, iters = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000000, 2000000, 3)
d = tempfile.mkdtemp()
t0 = time.time()
T.write_model(d + "/in", *T.general_lp(m, n, 31, frac_eq=0.0, frac_free=0.0, frac_boxed=0.0), crossover=0, stop_at_switch=1, ipm_maxiter=iters)
print("model written in %.1f s" % (time.time() - t0), flush=True)
keys = ("status_ipm", "iter", "kktiter1", "time_ipm1", "time_kkt_factorize", "time_kkt_solve", "time_cr1", "time_total", "pobjval", "dobjval", "rel_presidual", "rel_dresidual")
res = {}
for exe in (T.HIP_BIN, T.REF_BIN):
    t0 = time.time()
    info, _, out = T.run(exe, d + "/in", d + "/out_" + os.path.basename(exe), timeout=1100)
    res[exe] = info
    print(os.path.basename(exe), "wall %.1f s" % (time.time() - t0), {k: info.get(k) for k in keys}, flush=True)
a, b = res[T.HIP_BIN], res[T.REF_BIN]
print("device models: %d ipxk_create for %d solver objects (HipModel registry: one context per Model)" % (a.get("hip_model_creations", -1), a.get("hip_model_creations", 0) + a.get("hip_model_hits", 0)))
print("KKT solve time (Info::time_kkt_solve): reference %.2f s, Hip %.2f s -> %.1f x;  initial IPM (time_ipm1): %.2f / %.2f s -> %.1f x; kktiter1 %d / %d"
      % (b["time_kkt_solve"], a["time_kkt_solve"], b["time_kkt_solve"] / a["time_kkt_solve"], b["time_ipm1"], a["time_ipm1"], b["time_ipm1"] / a["time_ipm1"], b["kktiter1"], a["kktiter1"]))
