import sys, os, tempfile, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "./tests")
import test_gpu_lp_dropin as T
tmp = pathlib.Path(tempfile.mkdtemp())
ref, hip = T.both(tmp, "lp", *T.general_lp(6000, 15000, 31), crossover=1)
T.compare_runs(ref, hip)
print("REF:", {k: ref[0][k] for k in ("status_ipm", "status_crossover", "iter", "kktiter2", "updates_ipm", "pobjval", "objval", "time_ipm2")})
print("HIP:", {k: hip[0][k] for k in ("status_ipm", "status_crossover", "iter", "kktiter2", "updates_ipm", "pobjval", "objval", "time_ipm2", "kept_eta_calls", "device_maxvolume_calls")})
print("compare_runs: PASS")
