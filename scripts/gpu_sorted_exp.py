"""Sorted sub-tiles against the sliced layout at C3 (m = 1M, n = 2M, uniformly random): per-pass and per-apply
times for IPXK_SPMV_LAYOUT=sliced|sorted and the kernel variants (IPXK_SORTED_VARIANT).  One process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import ctypes as C
    import numpy as np
    from ipx_amd import synth, kkt
    m, n = 1000000, 2000000
    A = synth.synthetic_lp(m, n, 8, 12345)
    ctx = kkt.KktContext(A)
    rng = np.random.default_rng(0)
    ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
    ctx.set_pointer_mode(True)
    y = ctx.vector(m, rng.standard_normal(m)); t = ctx.vector(n); lhs = ctx.vector(m)
    ms = C.c_double(0)
    out = []
    for which, x, o in [(1, y, t), (2, t, lhs)]:
        ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), o.as_arg(), 3, C.byref(ms))
        ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), o.as_arg(), 20, C.byref(ms))
        out.append(ms.value / 20 * 1e3)
    ctx.time_normal_apply(y, lhs, 5)
    ap = ctx.time_normal_apply(y, lhs, 50) / 50 * 1e3
    print("layout %s variant %s: pass1 %.1f us pass2 %.1f us apply %.1f us  (%s)" % (os.environ.get("IPXK_SPMV_LAYOUT"), os.environ.get("IPXK_SORTED_VARIANT"), out[0], out[1], ap, ctx.spmv_layout()), flush=True)
else:
    for lay, var in (("sliced", "-"), ("sorted", "-"), ("auto", "-")):
        env = dict(os.environ, IPXK_SORTED_VARIANT=var)
        if lay != "auto": env["IPXK_SPMV_LAYOUT"] = lay
        else: env.pop("IPXK_SPMV_LAYOUT", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=600)
