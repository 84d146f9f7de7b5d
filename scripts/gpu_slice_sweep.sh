#!/bin/bash
# sweeps the time-tiling slice size for the C3 NormalMatrix apply
for kb in 256 512 1024 2048 4096 65536; do
  echo "IPXK_SLICE_KB=$kb"
  IPXK_SLICE_KB=$kb python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ipx_amd import synth, kkt
for (m, n) in [(1000000, 2000000), (50000, 100000)]:
    A = synth.synthetic_lp(m, n, 8, 12345)
    ctx = kkt.KktContext(A)
    rng = np.random.default_rng(0)
    ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
    rhs = ctx.vector(m, rng.standard_normal(m)); lhs = ctx.vector(m)
    ctx.time_normal_apply(rhs, lhs, 3)
    ms = ctx.time_normal_apply(rhs, lhs, 20)
    B = ctx.normal_apply_bytes
    print("  m=%d n=%d: %.1f us/apply %.2f TB/s" % (m, n, ms/20*1e3, B/(ms/20*1e-3)/1e12), flush=True)
    ctx.close()
PY
done
