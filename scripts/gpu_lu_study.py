"""The LU of the IPM's bases under the reference's LpSolver (oracle/_ref/test_lp_hip only): phases, fill and the LU lines of
the library for a list of environment variants.
usage: python scripts/gpu_lu_study.py m n seed out_prefix VARIANT [VARIANT ...]     VARIANT = "name:K=V,K=V" (name alone: defaults)"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lp_dropin as T
m, n, seed, prefix = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
d = tempfile.mkdtemp()
T.write_model(d + "/in", *T.general_lp(m, n, seed), crossover=0)
for var in sys.argv[5:]:
    name, _, kv = var.partition(":")
    env = dict(os.environ, IPXK_VERBOSE="1")
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        env[k] = v.replace("@", prefix + "_" + name)
        if k == "IPXK_LU_DUMP": os.makedirs(env[k], exist_ok=True)
    os.makedirs(d + "/out", exist_ok=True)
    t0 = time.time()
    try:
        r = subprocess.run([T.HIP_BIN, d + "/in", d + "/out"], capture_output=True, text=True, env=env, timeout=1000)
    except subprocess.TimeoutExpired:
        print(name, "TIMEOUT", flush=True)
        continue
    wall = time.time() - t0
    with open("%s_%s.log" % (prefix, name), "w") as f:
        f.write(r.stdout[-4000:] + "\n" + "\n".join(ln for ln in r.stderr.splitlines() if "LU dim" in ln or "Error" in ln or "error" in ln))
    if r.returncode != 0 or not os.path.exists(d + "/out/info.txt"):
        print(name, "FAILED rc", r.returncode, r.stdout[-500:], r.stderr[-1500:], flush=True)
        continue
    info = dict(ln.split() for ln in open(d + "/out/info.txt"))
    lus = [ln for ln in r.stderr.splitlines() if "LU dim" in ln]
    fills = []
    for ln in lus:
        mm = re.search(r"nnz (\d+):.*nnz\(L\) (\d+) nnz\(U\) (\d+)", ln)
        if mm and int(mm.group(1)) > m: fills.append((int(mm.group(2)) + int(mm.group(3))) / int(mm.group(1)))
    print(name, "wall %.1f" % wall, {k: info[k] for k in ("status_ipm", "iter", "kktiter2", "updates_ipm", "pobjval", "time_ipm2", "time_kkt_factorize", "time_kkt_solve",
                                                      "time_maxvol", "lu_factorizations", "lu_max_bump", "lu_device_seconds", "device_maxvolume_calls", "cpu_maxvolume_calls") if k in info},
          {k.replace("factorize_", "").replace("_seconds", ""): round(float(v), 2) for k, v in info.items() if k.startswith("factorize_")},
          "LU calls %d, fill mean %.1f max %.1f" % (len(lus), sum(fills) / max(len(fills), 1), max(fills or [0])), flush=True)
    for ln in lus[len(lus) // 2: len(lus) // 2 + 3]: print("   ", ln)
    os.remove(d + "/out/info.txt")
