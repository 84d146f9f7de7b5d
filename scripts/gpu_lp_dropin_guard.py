"""The reference's LpSolver through the Hip classes on a synthetic LP with IPXK_VERBOSE=1: phase times and what the guard of the
explicit inverses said about every dense block of the IPM's bases.  usage: python scripts/gpu_lp_dropin_guard.py m n [seed]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp(m, n, seed), crossover=0)
os.makedirs(d + "/out", exist_ok=True)
t0 = time.time()
r = subprocess.run([T.HIP_BIN, d + "/in", d + "/out"], capture_output=True, text=True, timeout=1100, env=dict(os.environ, IPXK_VERBOSE="1", IPXK_SWEEP_STATS="1"))
print(r.stdout.strip().splitlines()[0] if r.stdout.strip() else "", "wall %.1f" % (time.time() - t0))
info = {ln.split()[0]: float(ln.split()[1]) for ln in open(d + "/out/info.txt")}
print({k: info[k] for k in ("iter", "kktiter2", "updates_ipm", "time_ipm2", "time_kkt_factorize", "time_kkt_solve", "time_maxvol", "time_cr2", "lu_device_seconds")})
lines = [l for l in r.stderr.splitlines() if "dense block" in l or "own probe" in l or "before the refinement" in l]
dense = [l for l in lines if "dense block" in l]
print(len(dense), "dense blocks;", sum("refinement" in l for l in dense), "refined;", sum("REJECTED" in l for l in dense), "rejected")
for l in lines: print(l[:190])
sw = [l for l in r.stderr.splitlines() if "ipxk: sweep " in l and "inverted" in l]
print(len(sw), "inverted head / tail blocks of sweeps probed;", sum("REJECTED" in l or "rejected" in l for l in sw), "rejected")
for l in sw[:8]: print(l[:190])
lv = [l for l in r.stderr.splitlines() if "sweep blocks:" in l]
print(len(lv), "sweep plans; the last eight:")
for l in lv[-8:]: print(l[:190])
lu = [l for l in r.stderr.splitlines() if l.startswith("ipxk: LU dim")]
print(len(lu), "device factorizations; every fourth:")
for l in lu[::4]: print(l[:230])
