"""Randomized parity run of the LU's elimination rounds against the CPU restatement: bases with misplaced or exchanged columns of
many sizes and seeds, small dense limits so that the rounds run long; permutations, patterns and values must agree bit for bit.
usage: python scripts/gpu_sparse_lu_stress.py [cases] [auto]     (auto: the default policy -- tearing, the rounds where tearing refuses)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
auto = len(sys.argv) > 2 and sys.argv[2] == "auto"
if not auto:
    os.environ["IPXK_LU_SPARSE"] = "1"
os.environ["IPXK_LU_MFMA_MIN"] = "0"
ctx = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
orc = po.Oracle()
rng = np.random.default_rng(2024)
bad = 0
for case in range(ncases):
    dim = int(rng.integers(1500, 30000))
    kind = case % 3
    seed = int(rng.integers(1, 10**6))
    if kind == 0:
        G = synth.misplaced_basis_matrix(dim, int(rng.integers(5, 80)), seed=seed, bump=int(rng.integers(10, 200)))
    elif kind == 1:
        G = synth.disturbed_basis_matrix(seed=seed, dim=dim, num_exchanged=int(rng.integers(4, 40)), bump=int(rng.integers(10, 120)), offdiag=int(rng.integers(2, 4)))
    else:
        G = synth.lp_like_basis_matrix(dim=dim, bump=int(rng.integers(300, 1500)), bump_density=float(rng.uniform(0.005, 0.05)), seed=seed)
    limit = int(rng.integers(4, 60)) if auto else int(rng.integers(16, 400))
    smin = min(512, limit) if auto else int(rng.integers(2, max(3, limit // 2)))
    slow = 256 if auto else int(rng.choice([0, 16, 256]))
    os.environ["IPXK_LU_BUMP_MAX"] = str(limit)
    if not auto:
        os.environ["IPXK_LU_SPARSE_MIN"] = str(smin)
        os.environ["IPXK_LU_SPARSE_SLOW_DEN"] = str(slow)
    try:
        F = ctx.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
    except kkt.KktError as e:
        F = None
        msg = str(e)[:100]
    Fo = orc.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=limit) if auto else None
    if Fo is None:
        Fo = orc.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=limit, sparse_min=smin, slow_den=slow)
    if F is None or Fo is None:
        ok = F is None and Fo is None
        print("case %d dim %d kind %d limit %d: device %s, restatement %s -> %s" % (case, dim, kind, limit, "refused" if F is None else "ok", "refused" if Fo is None else "ok", "agree" if ok else "DISAGREE"), flush=True)
        bad += not ok
        continue
    same = all(np.array_equal(F[k], Fo[k]) for k in ("rowperm", "colperm", "dependent")) and \
        all(np.array_equal(getattr(F[f], a), getattr(Fo[f], a)) for f in ("L", "U") for a in ("p", "i", "x"))
    print("case %d dim %d kind %d limit %d smin %d slow %d: %d spikes, %d rounds, %d pivots, rest %d, %d dependent, fill %.2f -> %s" %
          (case, dim, kind, limit, smin, slow, F["spikes"], F["sparse_rounds"], F["sparse_pivots"], F["bump"], F["num_dependent"], (F["lnz"] + F["unz"]) / len(G["Bi"]),
           "identical" if same else "DIFFERENT"), flush=True)
    bad += not same
print("%d of %d cases disagree" % (bad, ncases))
ctx.close()
sys.exit(1 if bad else 0)
