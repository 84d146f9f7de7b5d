"""Dumps two bases of the 50 000 x 125 000 LP (IPXK_LU_DUMP) in one process, then -- in a fresh process each -- factorizes one of them four
times under the default policy and compares the first call with the later ones bit for bit.   usage: ... dump | check FILE"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if sys.argv[1] == "dump":
    import tempfile, test_gpu_lp_dropin as T
    d = tempfile.mkdtemp()
    T.write_model(d + "/in", *T.general_lp(50000, 125000, 31), crossover=0)
    out = os.path.join(ROOT, "gpurun_out", "bases50k"); os.makedirs(out, exist_ok=True); os.makedirs(d + "/out", exist_ok=True)
    env = dict(os.environ, IPXK_LU_DUMP=out, IPXK_LU_DUMP_EVERY="3")
    r = subprocess.run([T.HIP_BIN, d + "/in", d + "/out"], capture_output=True, text=True, env=env, timeout=1000)
    print(r.stdout[-300:]); print(sorted(os.listdir(out))[:8])
    for f in sorted(os.listdir(out))[3:]: os.remove(os.path.join(out, f))          # keep three
else:
    from ipx_amd import kkt, synth
    raw = open(sys.argv[2], "rb").read()
    dim, nb = np.frombuffer(raw[:16], np.int64)
    o = 16
    Bp = np.frombuffer(raw[o:o + 4 * (dim + 1)], np.int32).astype(np.int64); o += 4 * (dim + 1)
    Bi = np.frombuffer(raw[o:o + 4 * nb], np.int32).astype(np.int64); o += 4 * nb
    Bx = np.frombuffer(raw[o:o + 8 * nb], np.float64).copy()
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    runs = []
    for rep in range(4):
        F = c.lu_factorize(int(dim), Bp[:-1], Bp[1:], Bi, Bx, 0.1, download=True)
        runs.append(F)
        print("call %d: bump %d sparse pivots %d, nnz(L) %d nnz(U) %d, bump phase %.1f ms" % (rep, F["bump"], F.get("sparse_pivots", -1), F["L"].nnz, F["U"].nnz, F["seconds_bump"] * 1e3), flush=True)
    a = runs[0]
    for rep in range(1, 4):
        b = runs[rep]
        pat = a["L"].i.shape == b["L"].i.shape and np.array_equal(a["L"].i, b["L"].i) and np.array_equal(a["U"].i, b["U"].i) and np.array_equal(a["rowperm"], b["rowperm"]) and np.array_equal(a["colperm"], b["colperm"])
        dl = int(np.count_nonzero(a["L"].x != b["L"].x)) if a["L"].x.shape == b["L"].x.shape else -1
        du = int(np.count_nonzero(a["U"].x != b["U"].x)) if a["U"].x.shape == b["U"].x.shape else -1
        print("call 0 vs call %d: pattern and permutations %s, differing values L %d U %d" % (rep, "equal" if pat else "DIFFERENT", dl, du))
