"""the reference's LpSolver through the Hip classes on general_lp(m, n, seed): last lines of stderr (IPXK_VERBOSE=1) -- what went wrong when a run ends
with IPX_STATUS_internal_error.  usage: python scripts/gpu_lp_stderr.py m n [seed]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp(m, n, seed), crossover=0, debug=1, display=1)
os.makedirs(d + "/out", exist_ok=True)
r = subprocess.run([T.HIP_BIN, d + "/in", d + "/out"], capture_output=True, text=True, env=dict(os.environ, IPXK_VERBOSE="1"), timeout=1000)
print("rc", r.returncode)
print("\n".join(r.stdout.splitlines()[-25:]))
print("---- stderr")
print("\n".join(ln[:400] for ln in r.stderr.splitlines()[-25:]))
