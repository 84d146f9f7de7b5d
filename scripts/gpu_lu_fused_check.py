"""The fused sub-panel update of the dense LU against the two-launch form, and the look-ahead (late trailing update on a second
stream) against the in-order form: the same factors bit for bit, and timings.
usage: python scripts/gpu_lu_fused_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

ctx = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
for dim, bump, dens in ((6000, 1500, 0.05), (9000, 3000, 0.02), (12000, 5000, 0.01), (20000, 8000, 0.01)):
    G = synth.lp_like_basis_matrix(dim=dim, bump=bump, bump_density=dens, seed=3)
    out = {}
    for mfma in ("0", None):
        for fused in ("0", "1"):
            os.environ["IPXK_LU_FUSED_SUB"] = fused
            if mfma is None:
                os.environ.pop("IPXK_LU_MFMA_MIN", None)
            else:
                os.environ["IPXK_LU_MFMA_MIN"] = mfma
            for rep in range(2):
                t0 = time.perf_counter()
                F = ctx.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=(rep == 1))
                dt = time.perf_counter() - t0
            out[(mfma, fused)] = (F, F["seconds_bump"] * 1e3)
    # look-ahead off / on (matrix cores, fused sub-panel update)
    os.environ.pop("IPXK_LU_MFMA_MIN", None)
    os.environ["IPXK_LU_FUSED_SUB"] = "1"
    la = {}
    for look in ("0", "1"):
        os.environ["IPXK_LU_LOOKAHEAD"] = look
        for rep in range(3):
            F = ctx.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=(rep == 2))
        la[look] = (F, F["seconds_bump"] * 1e3)
    os.environ.pop("IPXK_LU_LOOKAHEAD", None)
    a, b = la["0"][0], la["1"][0]
    same = all(np.array_equal(a[k], b[k]) for k in ("rowperm", "colperm", "dependent")) and \
        all(np.array_equal(getattr(a[f], x), getattr(b[f], x)) for f in ("L", "U") for x in ("p", "i", "x"))
    print("bump %d, look-ahead: off %.1f ms, on %.1f ms; factors equal bit for bit: %s" % (bump, la["0"][1], la["1"][1], same), flush=True)
    for mfma in ("0", None):
        a, b = out[(mfma, "0")][0], out[(mfma, "1")][0]
        same = all(np.array_equal(a[k], b[k]) for k in ("rowperm", "colperm", "dependent")) and \
            all(np.array_equal(getattr(a[f], x), getattr(b[f], x)) for f in ("L", "U") for x in ("p", "i", "x"))
        print("bump %d, trailing update %s: two launches %.1f ms, fused %.1f ms; factors equal bit for bit: %s" %
              (bump, "by FMA loops" if mfma == "0" else "on the matrix cores", out[(mfma, "0")][1], out[(mfma, "1")][1], same), flush=True)
ctx.close()
