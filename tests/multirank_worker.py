"""One rank of a partitioned KKTSolverDiag solve, started by tests/test_gpu_multirank.py as a separate
process.  All ranks share GPU 0; the collectives travel through the library's direct exchange
(IPXK_COMM=direct, hipIpc between the processes) because RCCL refuses two ranks on one device.
argv: rank world idfile outprefix partition(rows|columns) m n seed"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ipx_amd import kkt, partition, synth  # noqa: E402


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    idfile, out, part = sys.argv[3], sys.argv[4], sys.argv[5]
    m, n, seed = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    columns = part == "columns"
    A = synth.synthetic_lp(m, n, 8, seed)
    st = synth.synthetic_ipm_state(m, n, 1.0, seed)
    slab = partition.col_slab(A, st, rank, world) if columns else partition.row_slab(A, st, rank, world)
    ctx = kkt.KktContext(slab.A, device=0)
    if rank == 0:
        uid = ctx.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the communicator id")
            time.sleep(0.02)
        uid = open(idfile, "rb").read()
    ctx.comm_init(uid, rank, world, columns=columns)
    assert ctx.kkt_diag_factorize(slab.xl, slab.xu, slab.zl, slab.zu, st["mu"], precond_dense_cols=False) == 0
    tol = 0.3 * np.sqrt(st["mu"])
    u = np.random.default_rng(0).standard_normal(m)
    u_loc = u if columns else u[slab.r0:slab.r1]
    lhs, dot = ctx.normal_apply(u_loc)
    x, y, it, err, _ = ctx.kkt_diag_solve(slab.a, slab.b, tol, 500)
    np.savez(out + ".rank%d.npz" % rank, lhs=lhs, dot=dot, x=x, y=y, it=it, err=err)
    ctx.close()


if __name__ == "__main__":
    main()
