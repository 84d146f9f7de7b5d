"""GPU: drop-in at the reference's own C++ boundary.  oracle/_ref/test_dropin (built in the build
container from tests/dropin/dropin_main.cc + ipx_amd/host/kkt_solver_diag_hip.cc against the
reference's headers and objects) lets the reference's Model / Iterate / Control drive
ipx::KKTSolverDiag (reference, CPU) and ipx::KKTSolverDiagHip (MI355X) through the abstract
ipx::KKTSolver interface and compares errflag, iteration counts and solutions."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "test_dropin")


@pytest.mark.gpu
@pytest.mark.parametrize("m,n", [(3000, 7000), (50000, 100000)])
def test_kkt_solver_diag_hip_is_a_drop_in(m, n):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/test_dropin not built (needs the reference sources at build time)")
    r = subprocess.run([BIN, str(m), str(n)], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    # Factorize(iterate), Factorize(nullptr), LinearOperator level, two models of equal shape back to back (HipModel)
    assert r.stdout.count("PASS") == 4


def test_host_classes_compile_against_reference_headers():
    """CPU: the IPX-side classes are written against the reference's own headers."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")):
        pytest.skip("reference sources not present")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ipx_amd", "host"),
           "-I" + os.path.join(ref, "include"), "-I" + os.path.join(ref, "src")]
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ipx_amd", "host"),
                           os.path.join(ROOT, "tests", "dropin", "handoff_main.cc")])      # the glue needs no reference header
    for f in ("kkt_solver_diag_hip.cc", "lu_kernel_hip.cc"):
        subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only"] + inc + [os.path.join(ROOT, "ipx_amd", "host", f)])
    # kkt_solver_basis_hip.cc calls the reference's private KKTSolverBasis::DropPrimal / DropDual: it compiles against
    # src/kkt_solver_basis.h WITH the one friend line of INTEGRATION.md (applied to the preprocessed unit, as oracle/Makefile does)
    # and is rejected without it
    src = os.path.join(ROOT, "ipx_amd", "host", "kkt_solver_basis_hip.cc")
    pre = subprocess.run(["g++", "-std=c++11", "-E"] + inc + [src], capture_output=True, text=True, check=True).stdout
    line = "class KKTSolverBasis : public KKTSolver {"
    assert pre.count(line) == 1
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++-cpp-output", "-"], input=pre.replace(line, line + " friend class KKTSolverBasisHip;"),
                   text=True, check=True)
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++-cpp-output", "-"], input=pre, text=True, capture_output=True)
    assert r.returncode != 0 and "private" in r.stderr
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++"] + inc + ["-"],
                   input='#include "linear_operators_hip.h"\n', text=True, check=True)


@pytest.mark.gpu
def test_basis_hip_glue_on_golden_fixture(tmp_path):
    """the host-side glue of ipx::KKTSolverBasisHip (ipx_amd/host/device_glue.h) executed on plain arrays:
    tests/dropin/handoff_main.cc runs Factorize's hand-off (full, then scaling only) and Solve on the golden
    basis fixture; results against the fixture's reference-kernel solution and the CPU restatement"""
    import numpy as np
    from oracle import pyoracle as po
    g = np.load(os.path.join(ROOT, "tests", "golden", "basis_200.npz"))
    m, n = int(g["m"]), int(g["n"])
    d = str(tmp_path)
    i64, f64 = np.int64, np.float64
    rng = np.random.default_rng(9)
    colscale2 = g["colscale"] * np.where(np.isfinite(g["colscale"]) & (g["colscale"] > 0), 10.0 ** rng.uniform(-0.3, 0.3, n + m), 1.0)
    a, b = rng.uniform(-0.5, 0.5, n + m), rng.uniform(-0.5, 0.5, m)
    tol = 1e-9
    arrays = dict(dims=np.array([m, n], i64), Ap=g["Ap"][:n + 1], Ai=g["Ai"][:g["Ap"][n]], Ax=g["Ax"][:g["Ap"][n]],
                  Lp=g["Lp"], Li=g["Li"], Lx=g["Lx"], Up=g["Up"], Ui=g["Ui"], Ux=g["Ux"], rowperm=g["rowperm"],
                  colperm=g["colperm"], basis=g["basis"], status=g["status"], colscale=g["colscale"], colscale2=colscale2,
                  a=a, b=b, tol=np.array([tol]))
    for k, v in arrays.items():
        np.ascontiguousarray(v, dtype=i64 if np.issubdtype(np.asarray(v).dtype, np.integer) else f64).tofile(os.path.join(d, k + ".bin"))
    exe = os.path.join(d, "handoff")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ipx_amd", "host"),
                           os.path.join(ROOT, "tests", "dropin", "handoff_main.cc"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ipx_amd", "lib"), "-lipx_kkt_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ipx_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    assert "DEBUG3: CR method not converged in 2 iterations. residual = " in r.stdout, r.stdout
    rd = lambda k, t=f64: np.fromfile(os.path.join(d, k + ".bin"), dtype=t)
    iters = rd("iters", i64)
    assert list(iters[[1, 3, 5, 7]]) == [0, 0, 0, 201] and iters[6] == 2
    # the CPU restatement of KKTSolverBasis::_Solve on the same inputs
    orc = po.Oracle()
    cs = lambda p, i, x, nr, nc: po.Csc(nr, nc, p, i, x)
    Ap = g["Ap"][:n + 1]
    AIp = np.concatenate([Ap, Ap[n] + 1 + np.arange(m)])
    AI = po.Csc(m, n + m, AIp, np.concatenate([arrays["Ai"], np.arange(m)]), np.concatenate([arrays["Ax"], np.ones(m)]))
    L, U = cs(g["Lp"], g["Li"], g["Lx"], m, m), cs(g["Up"], g["Ui"], g["Ux"], m, m)
    for tag, csv in (("1", g["colscale"]), ("2", colscale2)):
        S = orc.split_prepare(AI, n, L, U, g["rowperm"], g["colperm"], g["basis"], g["status"], csv)
        xo, yo, ito, eo, _ = S.kkt_solve(a, b, tol)
        xg, yg = rd("x" + tag), rd("y" + tag)
        assert eo == 0 and abs(int(iters[0 if tag == "1" else 2]) - ito) <= 2
        assert np.abs(xg - xo).max() <= 1e-6 * np.abs(xo).max() and np.abs(yg - yo).max() <= 1e-6 * np.abs(yo).max()
    # scaling-only hand-off == full hand-off, bit for bit
    assert np.array_equal(rd("x2"), rd("x3")) and np.array_equal(rd("y2"), rd("y3")) and iters[2] == iters[4]


@pytest.mark.gpu
def test_example_ipm_loop():
    """examples/ipm_loop.cc: an interior-point loop through the C ABI only (C++, no Python in the path) --
    factorize from the resident iterate, predictor-corrector step, repeat; residuals and mu must fall."""
    import re
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([os.path.join(ROOT, "examples", "ipm_loop"), "2000", "5000", "6"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln.split() for ln in r.stdout.splitlines() if re.match(r"\s*\d+\s", ln)]
    assert len(rows) == 6
    pres = [float(x[1]) for x in rows]
    dres = [float(x[2]) for x in rows]
    mu = [float(x[3]) for x in rows]
    assert all(b < a for a, b in zip(pres, pres[1:])) and all(b < a for a, b in zip(dres, dres[1:]))
    assert mu[-1] < 0.1 * mu[0] and pres[-1] < 0.02 * pres[0]
    # with the fourth argument the example goes on with the basis-preconditioned phase until optimal
    r = subprocess.run([os.path.join(ROOT, "examples", "ipm_loop"), "1500", "3500", "4", "1"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    mo = re.search(r"main IPM: status (\d+) after (\d+) iterations.*pobjective (\S+) dobjective (\S+)", r.stdout)
    assert mo and int(mo.group(1)) == 1, r.stdout                               # IPX_STATUS_optimal
    po_, do_ = float(mo.group(3)), float(mo.group(4))
    assert abs(po_ - do_) <= 1e-8 * (1 + abs(po_))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,bump,nrep", [(400, 30, 12), (30000, 500, 40)])
def test_lu_kernel_hip_under_the_reference_forrest_tomlin(tmp_path, dim, bump, nrep):
    """ipx::LuKernelHip as the LuFactorization of the reference's own ForrestTomlin (oracle/_ref/test_lu_dropin,
    tests/dropin/lu_main.cc): factorize (flag 0, the reference's stability estimate < 1e-12), SolveDense both
    ways, then column replacements through FtranForUpdate / BtranForUpdate / Update and solves with the updated
    basis -- all arithmetic after the factorization is the reference's, on factors from the MI355X"""
    import numpy as np
    from ipx_amd import synth
    exe = os.path.join(ROOT, "oracle", "_ref", "test_lu_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_lu_dropin not built (needs the reference sources at build time)")
    G = synth.lp_like_basis_matrix(dim=dim, bump=bump, seed=21)
    rng = np.random.default_rng(5)
    d = str(tmp_path)
    i64, f64 = np.int64, np.float64
    np.array([dim, nrep], i64).tofile(os.path.join(d, "dims.bin"))
    for k in ("Bp", "Bi"):
        np.ascontiguousarray(G[k], i64).tofile(os.path.join(d, k + ".bin"))
    np.ascontiguousarray(G["Bx"], f64).tofile(os.path.join(d, "Bx.bin"))
    for k, pos in enumerate(rng.choice(dim, nrep, replace=False)):
        # the entering column: the leaving one perturbed plus a few new entries (keeps the basis nonsingular)
        ci, cx = G["Bi"][G["Bp"][pos]:G["Bp"][pos + 1]], G["Bx"][G["Bp"][pos]:G["Bp"][pos + 1]]
        extra = np.setdiff1d(rng.choice(dim, 3, replace=False), ci)
        ni = np.concatenate([ci, extra])
        nx = np.concatenate([cx * rng.uniform(0.7, 1.4, cx.size), rng.uniform(-0.3, 0.3, extra.size)])
        np.array([pos], i64).tofile(os.path.join(d, "pos_%d.bin" % k))
        ni.astype(i64).tofile(os.path.join(d, "ci_%d.bin" % k))
        nx.astype(f64).tofile(os.path.join(d, "cx_%d.bin" % k))
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("PASS") == 6 and "bump %d " % bump in r.stdout        # ... and the fallback delegation
