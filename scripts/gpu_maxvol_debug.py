import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po
from test_maxvolume_oracle import setup
m, n, bump, seed = 300, 700, 20, 4
P, status, colscale, Ao = setup(po, m, n, bump, seed)
o = po.Oracle()
t0 = time.time()
want = o.basis(Ao, P["basis"], status).maxvolume(colscale, rows_per_slice=100)
print("oracle", time.time() - t0, {k: v for k, v in want.items() if k != "exchanges"}, flush=True)
ctx = kkt.KktContext(P["A"])
ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
ctx.split_prepare_lu(status, colscale)
print("prepared", flush=True)
got = ctx.maxvolume(status, colscale, rows_per_slice=100)
print({k: v for k, v in got.items() if k not in ("exchanges", "basis", "status")}, flush=True)
print(np.array_equal(got["exchanges"], want["exchanges"]))
