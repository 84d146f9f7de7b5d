#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): KKT solves/sec + A*D*A' SpMV HBM GB/s on the 1M-row
synthetic LP, 1/2/4/8 GPUs.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one KKTSolver::Solve (reference src/kkt_solver_diag.cc:82-118: right-hand-side
assembly, preconditioned CR to tol = 0.3*sqrt(mu), solution recovery) on the C3 workload of
SURVEY.md section 8(d): m = 1M, n = 2M, 8 nnz/col, IPM scaling spread s = 1, inputs resident in
HBM.  N > 1: the SAME system with the rows of AI partitioned over the N ranks and one RCCL
all-reduce per NormalMatrix apply (strong scaling; value = solves/s of the joint solve); the
alternative column partition of SURVEY.md section 8(e) is measured in the same run and reported
in config.column_partition.

One JSON line on rank 0:  metric/value/unit/...,
  "roofline":     the NormalMatrix apply (its two sparse products, kernels named in the line) against the HBM roof:
                  algorithmic bytes 2*nnz*12 + (n+m+2)*4 + 8*(3n+4m) per apply / HIP-event time;
  "cpu_baseline": the same solve on ONE host core, timed in this run, by the reference's own
                  objects (oracle/_ref, kind "reference") when that build is present, else by
                  this repo's restatement (oracle/, kind "port").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--cols", type=int, default=2000000)
    ap.add_argument("--spread", type=float, default=1.0)
    ap.add_argument("--maxiter", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--basis", action="store_true", help="(default at N = 1; kept for compatibility)")
    ap.add_argument("--no-basis", action="store_true",
                    help="skip the basis-preconditioned solve on planted LU factors (config.basis_path)")
    ap.add_argument("--newton", action="store_true", help="(default at N = 1; kept for compatibility)")
    ap.add_argument("--no-newton", action="store_true",
                    help="skip the resident Newton step / IPM iteration measurements (config.newton_step)")
    ap.add_argument("--no-column-partition", action="store_true",
                    help="N > 1: skip the extra measurement of the column partition (config.column_partition)")
    ap.add_argument("--no-lu", action="store_true", help="skip the basis LU factorization on the device (config.lu_path)")
    ap.add_argument("--no-maxvolume", action="store_true", help="skip Maxvolume on the device (config.maxvolume_path)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference's LpSolver through both KKT solver classes (config.dropin_lp_solver)")
    ap.add_argument("--no-banded", action="store_true", help="skip the banded-matrix probe of the SpMV (extra field of roofline)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 2 and 5 (config.other_configs)")
    ap.add_argument("--no-direct-exchange", action="store_true",
                    help="N > 1: skip the extra measurement over the direct (hipIpc) exchange (config.direct_exchange)")
    return ap.parse_args()


def cpu_baseline(A, st, tol, maxiter):
    """One solve of the same system on one host core (bounded: ~10-16 s at the default size)."""
    return reference_diag_solve(A, st, tol, maxiter, 4)


def kkt_residual_scaled(A, W, a, b, x, y):
    """Residual of the system KKTSolver::Solve must satisfy (reference src/kkt_solver.h:21-27) for G = inv(W),
    recomputed on the CPU: returns (max |D*(G x + AI'y - a)| with D = sqrt(W), max |AI x - b|)."""
    S = A.to_scipy()
    n = A.ncol
    aty = np.concatenate([S.T @ y, y])
    res1 = x / W + aty - a
    res2 = S @ x[:n] + x[n:] - b
    return float(np.abs(res1 * np.sqrt(W)).max()), float(np.abs(res2).max())


def pcr_trajectory_check(ctx, A, a, b, tol, nit=10):
    """First `nit` loop-head residual norms of the preconditioned CR (reference src/conjugate_residuals.cc:129-138)
    on this KKT system: the HIP loop against the CPU restatement in oracle/ (pinned bit for bit to the
    reference's objects by tests/test_oracle_vs_ref.py), same right-hand side, maxiter = nit."""
    from oracle import pyoracle as po
    m, n = A.nrow, A.ncol
    W, resscale = ctx.kkt_diag_get()
    S = A.to_scipy()
    rhs = -b + S @ (W[:n] * a[:n]) + W[n:] * a[n:]          # kkt_solver_diag.cc:90-92
    _, it1, e1, h1, _ = ctx.pcr_solve(rhs, tol, resscale, nit, hist_cap=nit + 1)
    orc = po.Oracle()
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    P, _ = orc.diag_factorize(Ao, W, orc.find_dense_columns(Ao)[1], True)
    _, it2, e2, h2 = orc.pcr_solve(lambda v: orc.normal_apply(Ao, W, v), P.apply, rhs, tol, resscale, nit,
                                   hist_cap=nit + 1)
    k = min(len(h1), len(h2), nit)
    return {"first_resnorms_max_rel_diff": float(np.abs(h1[:k] - h2[:k]).max() / h2[:k].max()), "compared": int(k),
            "iter_errflag_gpu": [int(it1), int(e1)], "iter_errflag_cpu": [int(it2), int(e2)]}


def reference_diag_solve(A, st, tol, maxiter, reps):
    """KKTSolverDiag::Solve of the same system on ONE host core by the reference's own objects (oracle/_ref),
    else by the restatement; returns (cpu_baseline dict, (x, y, iterations))."""
    from oracle import pyoracle as po
    from ipx_amd import synth
    m, n = A.nrow, A.ncol
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    cpu = os.cpu_count()
    if po.ref_available():
        try:
            ref = po.Ref()
            v = synth.lp_vectors(m, n)
            rm = ref.model(Ao, v["rhs"], v["constr_type"], v["obj"], v["lb"], v["ub"])
            if rm.m == m and rm.n == n and not rm.dualized:
                k = rm.kkt_diag(maxiter=maxiter)
                k.factorize(np.ones(n + m), st["xl"], st["xu"], np.zeros(m), st["zl"], st["zu"])
                t0 = time.perf_counter()
                for _ in range(reps):
                    x, y, it, err = k.solve(st["a"], st["b"], tol)
                dt = (time.perf_counter() - t0) / reps
                return dict(value=1.0 / dt, unit="solves/s", cores=1, kind="reference",
                            sample="%d x KKTSolverDiag::Solve of the same system by the reference's objects "
                                   "(%d CR iterations, errflag %d, %.2f s each; host has %d cores)"
                                   % (reps, it, err, dt, cpu)), (x, y, it)
        except Exception as exc:   # reference build unusable on this box: use the port
            sys.stderr.write("reference baseline unavailable (%s); using the port\n" % exc)
    orc = po.Oracle()
    k = orc.kkt_diag(Ao, maxiter=maxiter)
    k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    t0 = time.perf_counter()
    x, y, it, err, _ = k.solve(st["a"], st["b"], tol)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="solves/s", cores=1, kind="port",
                sample="1 solve of the same system by oracle/ipx_oracle.cc (%d CR iterations, errflag %d, "
                       "%.2f s; host has %d cores)" % (it, err, dt, cpu)), (x, y, it)


def bench_diag_config(kkt, synth, label, m, n, num_dense, args, reps_cpu):
    """Another BASELINE.json config of the diag path on this GPU (not the headline value): ms per resident solve,
    the NormalMatrix apply against the HBM roof, parity checks and the reference's own solve as baseline."""
    A = synth.synthetic_lp(m, n, 8, 12345, num_dense=num_dense)
    st = synth.synthetic_ipm_state(m, n, args.spread, 12345)
    tol = 0.3 * np.sqrt(st["mu"])
    ctx = kkt.KktContext(A)
    t0 = time.perf_counter()
    err = ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    t_fact_first = time.perf_counter() - t0          # builds the dense-column structures once per model
    assert err == 0
    t0 = time.perf_counter()
    assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
    t_fact = time.perf_counter() - t0                # what every IPM iteration pays (host vectors over PCIe included)
    traj = pcr_trajectory_check(ctx, A, st["a"], st["b"], tol)
    xg, yg, itg, eg, _ = ctx.kkt_diag_solve(st["a"], st["b"], tol, args.maxiter)
    W, _ = ctx.kkt_diag_get()
    r1, r2 = kkt_residual_scaled(A, W, st["a"], st["b"], xg, yg)
    ctx.set_pointer_mode(True)
    a, b = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"])
    x, y = ctx.vector(n + m), ctx.vector(m)
    for _ in range(2):
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
    ctx.synchronize()
    K = 10
    each = []                        # (auxiliary configs: every solve timed on its own and the MEDIAN reported -- a 2 ms solve of 280
    for _ in range(K):               # launches is at the mercy of one host hiccup; the headline keeps the contract's K-step total)
        t0 = time.perf_counter()
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
        ctx.synchronize()
        each.append(time.perf_counter() - t0)
    dt = float(np.median(each))
    dt_mean = float(np.mean(each))
    rhs_d, lhs_d = ctx.vector(m, np.random.default_rng(0).standard_normal(m)), ctx.vector(m)
    ctx.time_normal_apply(rhs_d, lhs_d, 5)
    apply_ms = ctx.time_normal_apply(rhs_d, lhs_d, 50) / 50
    nbytes = ctx.normal_apply_bytes
    layouts = ctx.spmv_layout()[0]
    ctx.set_pointer_mode(False)
    res = {"workload": label, "solves_per_sec": 1.0 / dt, "ms_per_solve": dt * 1e3, "ms_per_solve_mean_of_%d" % K: dt_mean * 1e3,
           "cr_iterations": it, "errflag": errflag, "factorize_ms": t_fact * 1e3, "first_factorize_ms": t_fact_first * 1e3, "num_dense_cols": ctx.num_dense_cols,
           "roofline": {"bound": "hbm", "us_per_apply": apply_ms * 1e3, "algorithmic_bytes": nbytes,
                        "achieved": nbytes / (apply_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": nbytes / (apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "layouts": list(layouts)},
           "parity": {"pcr_trajectory": traj, "kkt_residual_scaled_over_tol": r1 / tol, "primal_residual": r2}}
    ctx.close()
    if not args.no_cpu_baseline:
        base, (xc, yc, itc) = reference_diag_solve(A, st, tol, args.maxiter, reps_cpu)
        res["cpu_baseline"] = base
        res["gpu_over_cpu"] = res["solves_per_sec"] / base["value"]
        res["parity"]["iter_gpu_cpu"] = [int(itg), int(itc)]
    return res


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from ipx_amd import kkt, synth
    from ipx_amd.partition import row_slab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # rehearsal on a one-GPU box (IPXK_REHEARSAL=1): all ranks share GPU 0, the library's collectives run over
    # its direct exchange (hipIpc between the processes) and torch.distributed over gloo (RCCL refuses
    # duplicate devices)
    rehearsal = os.environ.get("IPXK_REHEARSAL") == "1"
    if rehearsal:
        os.environ["IPXK_COMM"] = "direct"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    tdev = "cpu" if rehearsal else "cuda"

    m, n = args.rows, args.cols
    A = synth.synthetic_lp(m, n, 8, 12345)
    st = synth.synthetic_ipm_state(m, n, args.spread, 12345)
    tol = 0.3 * np.sqrt(st["mu"])                 # kkt_tol*sqrt(mu), reference src/ipm.cc:572

    # ---- this rank's slab of rows (the whole matrix when N = 1) ----
    slab = row_slab(A, st, rank, world)
    t_create = time.perf_counter()
    ctx = kkt.KktContext(slab.A, device=local_rank)
    t_create = time.perf_counter() - t_create
    if world > 1:
        ids = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], rank, world)
    elif os.environ.get("IPXK_FORCE_COMM"):
        # rehearsal of the collective code path on one GPU (1-rank RCCL communicator)
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    err = ctx.kkt_diag_factorize(slab.xl, slab.xu, slab.zl, slab.zu, st["mu"])
    assert err == 0
    ctx.set_pointer_mode(True)
    mg = slab.A.nrow
    a = ctx.vector(n + mg, slab.a)
    b = ctx.vector(mg, slab.b)
    x = ctx.vector(n + mg)
    y = ctx.vector(mg)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
    sync()
    t0 = time.perf_counter()
    cr_time = 0.0
    for _ in range(args.steps):
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
        cr_time += tm.cr
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # parity inputs of THIS system (later measurements re-factorize the context with other states)
    parity_inputs = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ctx.set_pointer_mode(False)
        parity_inputs = (ctx.kkt_diag_get()[0], x.download(), y.download(),
                         pcr_trajectory_check(ctx, A, st["a"], st["b"], tol))
        ctx.set_pointer_mode(True)

    # ---- roofline of the dominant kernel pair, measured live with HIP events on the ctx stream ----
    rng = np.random.default_rng(0)
    rhs_d = ctx.vector(mg, rng.standard_normal(mg))
    lhs_d = ctx.vector(mg)
    ctx.time_normal_apply(rhs_d, lhs_d, 5)
    reps = 50
    apply_ms = ctx.time_normal_apply(rhs_d, lhs_d, reps) / reps
    nnz = A.nnz
    bytes_apply = 2 * nnz * 12 + (n + m + 2) * 4 + 8 * (3 * n + 4 * m)   # whole system (all ranks)
    achieved = bytes_apply / (apply_ms * 1e-3) / 1e9
    peak = HBM_PEAK_GBS * world
    # L2<->fabric bytes per apply from the PMC passes of the same command (profiles/, separate
    # rocprofv3 --pmc runs; bench.py cannot collect counters on itself)
    layouts, tuned_us = ctx.spmv_layout()
    kernel_names = {"phased": "spmv_phased_kernel", "sliced": "spmv_sliced_tile_kernel+spmv_sliced_combine_kernel",
                    "fused": "spmv_sliced_tile_kernel<fused>", "sorted": "spmv_sorted_tile_kernel+spmv_sliced_combine_kernel",
                    "sortedfused": "spmv_sorted_fused_kernel", "acc": "spmv_acc_tile_kernel+spmv_sliced_combine_kernel",
                    "plain": "spmv_rowgroup_kernel", "accfused": "spmv_acc_fused_kernel"}
    traffic, traffic_note = pmc_traffic("pmc_traffic.json", "traffic_bytes_per_apply",
                                        "C3 m=%d n=%d nnz=%d" % (m, n, nnz) if world == 1 else None, layouts)

    out = {
        "metric": "kkt_solves_per_sec",
        "value": args.steps / dt,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%ssynthetic LP m=%d n=%d 8 nnz/col (nnz=%d), KKTSolverDiag::Solve = diag-precond CR "
                               "to tol=0.3*sqrt(mu), scaling spread s=%g; rows of AI partitioned over %d GPU(s)"
                               % ("C3 " if (m, n) == (1000000, 2000000) else "", m, n, nnz, args.spread, world),
                   "cr_iterations_per_solve": it, "errflag": errflag,
                   "cr_iterations_per_sec": it * args.steps / dt,
                   "cr_loop_ms_per_solve": cr_time / args.steps * 1e3},
        "roofline": {"bound": "hbm",
                     "kernel": "NormalMatrix apply = pass 1 t=W.*(A'y) [%s] + pass 2 lhs=A t [%s]"
                               % (kernel_names[layouts[0]], kernel_names[layouts[1]]),
                     "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_note": traffic_note, "us_per_apply": apply_ms * 1e3, "algorithmic_bytes": bytes_apply,
                     "layouts": list(layouts), "layout_tuning_us": tuned_us,
                     "note_50_percent": "the north star's >= 0.50 is NOT reached on this matrix: its row indices are uniformly random, every pass "
                                        "gathers 8-byte words from 8 M distinct 128-byte lines, and the chip serves such gathers from L2 / fabric at "
                                        "the rate this fraction reflects (profiles/r04_ldsacc_microbench.txt, r03_gather_sorted_microbench.txt); no "
                                        "renumbering helps (an expander: ipxk_create finds half of the rows within 8 levels of any row).  With locality "
                                        "the same kernels reach it: banded_matrix_probe, and shuffled_banded_probe where the locality is first "
                                        "recovered by the renumbering of layout_device.hip"},
    }
    try:
        ri = ctx.reorder_info()
        out["roofline"]["renumbering"] = {"active": bool(ri["active"]), "levels": int(ri["levels"]), "ms_in_create": ri["ms"]}
    except Exception:            # noqa: BLE001
        pass

    if (world > 1 or os.environ.get("IPXK_FORCE_COMM")) and not args.no_column_partition:
        # every rank takes part; reported next to the north-star row partition, never as `value`
        out["config"]["column_partition"] = bench_column_partition(kkt, dist, torch, A, st, tol, args, rank, world,
                                                                   local_rank, tdev)
    if world > 1 and not args.no_direct_exchange and (os.environ.get("IPXK_COMM") != "direct" or
                                                       os.environ.get("IPXK_BENCH_DIRECT_AGAIN")):
        # the same row-partitioned solve over the library's direct exchange (hipIpc-mapped peer buffers) next to
        # the RCCL measurement above; never `value`.  Any failure is recorded instead of a number.
        out["config"]["direct_exchange"] = bench_direct_exchange(kkt, dist, torch, A, st, tol, args, rank, world,
                                                                 local_rank, it, tdev)
    # what the transport itself reports about the communicator (ncclCommCount / ncclCommUserRank for RCCL): lets a
    # SCALE record be checked against the number of ranks that really exchanged data
    try:
        tname, tn, tr = ctx.comm_info()
        out["config"]["communicator"] = {"transport": tname, "nranks_reported_by_transport": tn, "rank": tr, "world_size": world}
    except Exception as exc:            # noqa: BLE001
        out["config"]["communicator"] = {"error": str(exc)}
    if rehearsal:
        out["config"]["transport"] = "REHEARSAL: all ranks on one GPU, direct exchange between the rank processes"
    elif world > 1:
        out["config"]["transport"] = "direct exchange over hipIpc-mapped peer buffers" if os.environ.get("IPXK_COMM") == "direct" else "RCCL"
    if rank == 0 and world == 1:
        # what a plain streaming kernel reaches on this box (report next to the 8 TB/s spec peak)
        triad = measure_triad(torch)
        out["roofline"]["measured_triad_GBps"] = triad
        out["roofline"]["frac_of_measured_triad"] = achieved / triad
    if rank == 0 and world == 1 and not args.no_banded:
        out["roofline"]["banded_matrix_probe"] = bench_banded(kkt, synth, m, n)
        out["roofline"]["shuffled_banded_probe"] = bench_banded_shuffled(kkt, synth, m, n)
    if rank == 0 and world == 1 and not args.no_basis:
        out["config"]["basis_path"] = bench_basis(kkt, synth, m, n, args)
    if rank == 0 and world == 1 and not args.no_lu:
        out["config"]["lu_path"] = bench_lu(kkt, synth, m, n, args)
    if rank == 0 and world == 1 and not args.no_maxvolume:
        out["config"]["maxvolume_path"] = bench_maxvolume(kkt, synth, m, n, args)
    if rank == 0 and world == 1 and not args.no_newton:
        out["config"]["newton_step"] = bench_newton(kkt, synth, ctx, m, n, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_dropin and (m, n) == (1000000, 2000000):
        out["config"]["dropin_lp_solver"] = bench_dropin_lp_solver(m, n)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ctx.set_pointer_mode(False)
        Wd, xg, yg, traj = parity_inputs
        base, (xc, yc, itc) = cpu_baseline(A, st, tol, args.maxiter)
        out["cpu_baseline"] = base
        out["config"]["gpu_over_cpu"] = out["value"] / base["value"]
        r1, r2 = kkt_residual_scaled(A, Wd, st["a"], st["b"], xg, yg)
        out["config"]["parity_vs_cpu"] = {"iter_gpu": it, "iter_cpu": itc,
                                          "pcr_trajectory": traj,
                                          "kkt_residual_scaled_over_tol": r1 / tol, "primal_residual": r2,
                                          "note": "gates of SURVEY 8d: iteration counts, first residual norms of the CR loop, "
                                                  "recomputed scaled KKT residual (src/kkt_solver.h:21-27); the final y of an "
                                                  "80-iteration run is not gated (1 ulp in b moves it by 1e-6)"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        out["config"]["other_configs"] = {
            "C2": bench_diag_config(kkt, synth, "BASELINE config 2: m=50k n=100k 8 nnz/col, diag-precond CR", 50000, 100000, 0, args, 10),
            "C5": bench_diag_config(kkt, synth, "BASELINE config 5: m=200k n=400k + 32 dense columns (DiagonalPrecond with "
                                                "Sherman-Morrison-Woodbury, src/diagonal_precond.cc:48-101)", 200000, 400000, 32, args, 3)}
    if rank == 0 and world == 1:
        # ipxk_create of THIS run's context (the first in the process: includes the HIP runtime's start-up) and a second
        # one on the warm runtime -- what a solver object pays when no context of its Model exists yet
        t0 = time.perf_counter()
        c2 = kkt.KktContext(A, device=local_rank)
        t_second = time.perf_counter() - t0
        _, ms = c2.layout_info(0)
        c2.close()
        out["config"]["model_upload"] = {"ipxk_create_first_in_process_s": t_create, "ipxk_create_warm_s": t_second,
                                         "inside_the_library_ms": {"upload_narrow_transpose": ms[0], "layouts_of_A_transposed": ms[1],
                                                                   "layouts_of_A": ms[2], "dense_columns_and_rest": ms[3]},
                                         "note": "model uploaded as it is; validation, Transpose and the sliced / accumulated layouts by radix "
                                                 "sorts on the device (layout_device.hip); 3.2 s of host loops per solver object in round 3"}
    if rank == 0:
        # short copies of the numbers a reader looks for first, at the END of the line (the driver keeps the tail)
        cfg = out["config"]
        summ = {"headline_solves_per_s": out["value"], "apply_us": out["roofline"]["us_per_apply"], "apply_frac_of_hbm_peak": out["roofline"]["frac"],
                "apply_traffic_over_algorithmic": (out["roofline"]["traffic"] / bytes_apply) if out["roofline"].get("traffic") else None,
                "layouts": list(layouts)}
        bp = cfg.get("basis_path") or {}
        if "roofline" in bp:
            summ["basis_us_per_cr_iteration"] = bp.get("us_per_cr_iteration")
            summ["basis_frac_of_hbm_peak"] = bp["roofline"].get("frac")
            tr = bp["roofline"].get("traffic")
            summ["basis_traffic_over_algorithmic"] = (tr / bp["roofline"]["algorithmic_bytes"]) if tr and bp["roofline"].get("algorithmic_bytes") else None
            summ["basis_solves_per_s"] = bp.get("solves_per_sec")
        dl = cfg.get("dropin_lp_solver") or {}
        for k in ("time_kkt_solve_reference_s", "time_kkt_solve_hip_s", "time_ipm1_reference_s", "time_ipm1_hip_s", "ipxk_create_calls_over_solver_objects"):
            if k in dl:
                summ["dropin_" + k] = dl[k]
        if "model_upload" in cfg:
            summ["ipxk_create_warm_ms"] = cfg["model_upload"]["ipxk_create_warm_s"] * 1e3
        oc = cfg.get("other_configs") or {}
        for k in ("C2", "C5"):
            if k in oc and isinstance(oc[k], dict):
                summ[k + "_ms_per_solve"] = oc[k].get("ms_per_solve")
        if "roofline" in out and "banded_matrix_probe" in out["roofline"]:
            summ["banded_probe_frac"] = out["roofline"]["banded_matrix_probe"].get("frac")
        if "roofline" in out and "shuffled_banded_probe" in out["roofline"]:
            summ["shuffled_banded_probe_frac"] = out["roofline"]["shuffled_banded_probe"].get("frac_cr_iteration_apply")
            summ["shuffled_banded_probe_frac_as_given"] = out["roofline"]["shuffled_banded_probe"].get("frac_as_given")
        if out.get("cpu_baseline"):
            summ["gpu_over_cpu"] = cfg.get("gpu_over_cpu")
        ib = cfg.get("lu_path", {}).get("ipm_basis_16000", {})
        if "default_policy" in ib:
            summ["lu_ipm_basis_16000_ms"] = ib["default_policy"]["factorize_ms"]
            summ["lu_ipm_basis_16000_fill"] = ib["default_policy"]["fill_factor"]
            summ["lu_ipm_basis_16000_round4_policy_ms"] = ib.get("round4_policy", {}).get("factorize_ms")
            summ["lu_ipm_basis_16000_round4_policy_fill"] = ib.get("round4_policy", {}).get("fill_factor")
        out["summary"] = summ
    if rank == 0:
        out["multi_gpu_note"] = ("N > 1 is launched by the driver only; no 8-GPU curve exists in this repo until a "
                                 "SCALE_rNN.json is recorded" if world == 1 else "rows of AI partitioned over the ranks")
        if "summary" in out:
            out["summary"] = out.pop("summary")          # last key of the line
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def bench_direct_exchange(kkt, dist, torch, A, st, tol, args, rank, world, local_rank, it_rccl, tdev="cuda"):
    """Row partition again, collectives through IPXK_COMM=direct.  Every rank runs the same sequence; an error on
    any rank (set-up, a bounded wait that expired) is agreed on through torch.distributed and reported."""
    from ipx_amd.partition import row_slab
    res, ctx, failed = {}, None, 0
    try:
        slab = row_slab(A, st, rank, world)
        ctx = kkt.KktContext(slab.A, device=local_rank)
        before = os.environ.get("IPXK_COMM")
        os.environ["IPXK_COMM"] = "direct"
        try:
            ids = [ctx.comm_unique_id() if rank == 0 else None]
        finally:
            if before is None:
                os.environ.pop("IPXK_COMM", None)
            else:
                os.environ["IPXK_COMM"] = before
    except Exception as exc:                                 # noqa: BLE001 - recorded, not raised
        res["error"] = "set-up: %s" % exc
        ids, failed = [None], 1
    flag = torch.tensor([failed], dtype=torch.int32, device=tdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        return res or {"error": "set-up failed on another rank"}
    dist.broadcast_object_list(ids, src=0)
    try:
        ctx.comm_init(ids[0], rank, world)
        assert ctx.kkt_diag_factorize(slab.xl, slab.xu, slab.zl, slab.zu, st["mu"]) == 0
        ctx.set_pointer_mode(True)
        mg = slab.A.nrow
        n = A.ncol
        a, b = ctx.vector(n + mg, slab.a), ctx.vector(mg, slab.b)
        x, y = ctx.vector(n + mg), ctx.vector(mg)
        for _ in range(args.warmup):
            it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
        ctx.synchronize(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
        ctx.synchronize(); torch.cuda.synchronize(); dist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        rhs_d, lhs_d = ctx.vector(mg, np.random.default_rng(0).standard_normal(mg)), ctx.vector(mg)
        ctx.time_normal_apply(rhs_d, lhs_d, 5)
        apply_ms = ctx.time_normal_apply(rhs_d, lhs_d, 50) / 50
        res = {"solves_per_sec": args.steps / dt, "ms_per_solve": dt / args.steps * 1e3, "cr_iterations": it,
               "errflag": errflag, "us_per_apply": apply_ms * 1e3, "cr_iterations_rccl_run": it_rccl,
               "exchange": "reduce-scatter + all-gather kernels over hipIpc-mapped peer buffers (DESIGN section 7)"}
    except Exception as exc:                                 # noqa: BLE001
        res = {"error": str(exc)}
    finally:
        try:
            if ctx is not None:
                ctx.close()
        except Exception:                                    # noqa: BLE001
            pass
    return res


def bench_column_partition(kkt, dist, torch, A, st, tol, args, rank, world, local_rank, tdev="cuda"):
    """The SAME solve with the structural COLUMNS of A partitioned over the ranks (alternative):
    every m-vector and every CR scalar is replicated, the one exchange per NormalMatrix apply is an
    all-reduce of m doubles (half the bytes of the row partition's n doubles, and no scalar exchange)."""
    from ipx_amd.partition import col_slab
    slab = col_slab(A, st, rank, world)
    ctx = kkt.KktContext(slab.A, device=local_rank)
    if world > 1:
        ids = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], rank, world, columns=True)
    else:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1, columns=True)
    assert ctx.kkt_diag_factorize(slab.xl, slab.xu, slab.zl, slab.zu, st["mu"]) == 0
    ctx.set_pointer_mode(True)
    m, ng = A.nrow, slab.A.ncol
    a, b = ctx.vector(ng + m, slab.a), ctx.vector(m, slab.b)
    x, y = ctx.vector(ng + m), ctx.vector(m)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it, errflag, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, args.maxiter)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    rhs_d, lhs_d = ctx.vector(m, np.random.default_rng(0).standard_normal(m)), ctx.vector(m)
    ctx.time_normal_apply(rhs_d, lhs_d, 5)
    apply_ms = ctx.time_normal_apply(rhs_d, lhs_d, 50) / 50
    ctx.close()
    return {"solves_per_sec": args.steps / dt, "ms_per_solve": dt / args.steps * 1e3, "cr_iterations": it,
            "errflag": errflag, "us_per_apply": apply_ms * 1e3,
            "exchange": "one all-reduce of m = %d doubles per NormalMatrix apply, no scalar exchange" % m}


def bench_newton(kkt, synth, ctx, m, n, args):
    """IPM::SolveNewtonSystem (src/ipm.cc:532-645) on the device around the same KKTSolverDiag: right-hand side
    from residuals, solve, recovery of the six step components; all 16 vectors resident."""
    st = synth.synthetic_newton_state(m, n, 12345)
    ctx.set_pointer_mode(False)
    assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
    ctx.set_pointer_mode(True)
    tol = 0.3 * np.sqrt(st["mu"])
    N = n + m
    keys = ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu")
    dev_in = [ctx.vector(m if k == "rb" else N, st[k]) for k in keys]
    state = ctx.state_vector(st["state"])
    dev_out = [ctx.vector(m if k == 3 else N) for k in range(6)]
    it, err, tm = ctx.newton_solve_resident(False, dev_in, state, tol, args.maxiter, dev_out)
    K = 5
    t0 = time.perf_counter()
    for _ in range(K):
        it, err, tm = ctx.newton_solve_resident(False, dev_in, state, tol, args.maxiter, dev_out)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / K
    res = {"steps_per_sec": 1.0 / dt, "ms_per_step": dt * 1e3, "cr_iterations": it, "errflag": err,
           "ms_outside_cr_loop": dt * 1e3 - tm.cr * 1e3}
    # one whole IPM iteration on the device: KKTSolverDiag::Factorize from the resident iterate, then
    # IPM::Predictor + AddCorrector + StepSizes + MakeStep (src/ipm.cc:340-530)
    P = synth.synthetic_iterate(m, n, 12345, frac_special=0.0)
    ctx.set_pointer_mode(False)
    ctx.iterate_set(P["it"], P["state"])
    ctx.set_pointer_mode(True)
    mb = ctx.vector(m, P["rhs"])
    mc = ctx.vector(N, np.concatenate([P["obj"], np.zeros(m)]))
    mlb, mub = ctx.vector(N, P["lbs"]), ctx.vector(N, P["ubs"])
    t0 = time.perf_counter()
    assert ctx.iterate_factorize_diag() == 0
    info = ctx.ipm_step_resident(False, mb, mc, mlb, mub, 0.3, args.maxiter)
    ctx.synchronize()
    res["ipm_iteration"] = {"ms": (time.perf_counter() - t0) * 1e3,
                            "kkt_iterations": [info["kktiter_predictor"], info["kktiter_corrector"]],
                            "errflag": info["errflag"], "step_primal": info["step_primal"],
                            "step_dual": info["step_dual"],
                            "note": "Factorize + predictor + corrector + step sizes + update, nothing crosses PCIe"}
    return res


def measure_triad(torch, nbytes=1 << 29):
    """a = b + 1.5*c over three 512 MB fp64 vectors (plumbing kernel of torch): bytes moved / time"""
    n = nbytes // 8
    bt = torch.ones(n, dtype=torch.float64, device="cuda")
    ct = torch.ones(n, dtype=torch.float64, device="cuda")
    at = torch.empty_like(bt)
    for _ in range(3):
        torch.add(bt, ct, alpha=1.5, out=at)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        torch.add(bt, ct, alpha=1.5, out=at)
    e1.record()
    torch.cuda.synchronize()
    return 3.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def bench_banded(kkt, synth, m, n):
    """Same kernel, same size, but gathers with locality (rows of a column within a 4096-row band):
    shows what the SpMV reaches when the x gathers hit in cache."""
    A = synth.banded_lp(m, n, 8, 4096, 12345)
    ctx = kkt.KktContext(A)
    rng = np.random.default_rng(0)
    ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
    rhs, lhs = ctx.vector(m, rng.standard_normal(m)), ctx.vector(m)
    ctx.time_normal_apply(rhs, lhs, 5)
    ms = ctx.time_normal_apply(rhs, lhs, 50) / 50
    nbytes = ctx.normal_apply_bytes
    layouts = ctx.spmv_layout()[0]
    ctx.close()
    return {"us_per_apply": ms * 1e3, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
            "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "matrix": "8 rows per column within a 4096-row band",
            "layouts": list(layouts)}


def bench_banded_shuffled(kkt, synth, m, n):
    """The banded matrix of bench_banded with its rows and columns in random order -- structure a modelling tool has hidden.  ipxk_create
    looks for a renumbering (breadth-first levels, layout_device.hip), keeps a renumbered copy if its two products are faster, and the CR
    loop of the KKT solve runs on it.  Reported: the two products on the model as given (NormalMatrix::Apply through the ABI), on the
    renumbered copy (the library's own timing at creation), and what a CR iteration of the KKT solve costs with and without the copy."""
    import os, time
    A, _, _ = synth.shuffled(synth.banded_lp(m, n, 8, 4096, 12345), 7)
    st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
    tol = 0.3 * np.sqrt(st["mu"])
    out = {"matrix": "8 rows per column within a 4096-row band, rows and columns randomly permuted"}
    res = {}
    for mode in ("0", None):
        if mode is None: os.environ.pop("IPXK_REORDER", None)
        else: os.environ["IPXK_REORDER"] = mode
        t0 = time.time()
        ctx = kkt.KktContext(A)
        create_s = time.time() - t0
        info = ctx.reorder_info()
        assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
        ctx.set_pointer_mode(True)
        a, b, x, y = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"]), ctx.vector(n + m), ctx.vector(m)
        it, err, _ = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500)
        ctx.synchronize()
        t0 = time.time()
        reps = 5
        for _ in range(reps): it, err, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500)
        ctx.synchronize()
        ms_solve = (time.time() - t0) / reps * 1e3
        nbytes = ctx.normal_apply_bytes
        res[mode] = dict(create_s=create_s, info=info, iters=it, errflag=err, ms_per_solve=ms_solve, us_per_cr_iteration=ms_solve * 1e3 / max(it + 1, 1), nbytes=nbytes)
        ctx.close()
    os.environ.pop("IPXK_REORDER", None)
    i1 = res[None]["info"]
    nb = res[None]["nbytes"]
    out.update(renumbering_active=bool(i1["active"]), levels=int(i1["levels"]), reorder_ms=i1["ms"],
               us_two_products_as_given=i1["us_original"], us_two_products_renumbered=i1["us_reordered"],
               frac_as_given=(nb / (i1["us_original"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if i1["us_original"] else None,
               frac_cr_iteration_apply=(nb / (i1["us_reordered"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if i1["us_reordered"] else None,
               kkt_solve_as_given={k: res["0"][k] for k in ("iters", "errflag", "ms_per_solve", "us_per_cr_iteration")},
               kkt_solve_renumbered={k: res[None][k] for k in ("iters", "errflag", "ms_per_solve", "us_per_cr_iteration")},
               create_s_as_given=res["0"]["create_s"], create_s_with_renumbering=res[None]["create_s"],
               note="frac_* = algorithmic bytes of one NormalMatrix::Apply over the time of its two products (epilogue without weights), timed by the library at creation on both copies")
    return out


# Sources whose kernels move the bytes the PMC summaries under profiles/ report.  A summary names the git blob hash
# of each of these files as it was when the counters were collected; bench.py cannot collect counters on itself
# (rocprofv3 --pmc runs are separate passes), so it reports a committed figure only while those files are unchanged.
PMC_SOURCES = ("ipx_amd/csrc/spmv_kernels.hpp", "ipx_amd/csrc/spmv.hip", "ipx_amd/csrc/trisolve.hip",
               "ipx_amd/csrc/trisolve.hpp", "ipx_amd/csrc/cr.hip", "ipx_amd/csrc/internal.hpp")


def blob_hash(path):
    """git's blob hash of a working-tree file (sha1 of 'blob <size>\\0' + content), without needing git"""
    import hashlib
    try:
        data = open(os.path.join(ROOT, path), "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    return {p: blob_hash(p) for p in PMC_SOURCES if blob_hash(p) is not None}


def pmc_traffic(fname, key, workload, layouts):
    """(bytes, note): the committed PMC figure if it was collected for this workload, these layouts and THESE kernel
    sources (blob hashes stored next to the figure); otherwise (None, reason)"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", fname)))
    except (OSError, ValueError):
        return None, "no PMC summary profiles/%s" % fname
    if workload is None or pm.get("workload") != workload:
        return None, "PMC summary is for another workload (%s)" % pm.get("workload")
    if list(pm.get("layouts", [])) != list(layouts):
        return None, "PMC summary is for layouts %s, this run uses %s" % (pm.get("layouts"), list(layouts))
    have = pm.get("source_hashes")
    if not have:
        return None, "PMC summary carries no source hashes: cannot tell whether it matches these kernels"
    now = source_hashes()
    changed = sorted(p for p in have if now.get(p) != have[p])
    if changed:
        return None, "kernel sources changed since the PMC run (%s): re-run scripts/prof_r05.sh" % ", ".join(changed)
    return pm[key], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %s; sources unchanged since (%d files hashed)" % (
        pm.get("source", "profiles/"), len(have))


def bench_basis(kkt, synth, m, n, args):
    """BASELINE config 3: KKTSolverBasis::_Solve (src/kkt_solver_basis.cc:75-194) = plain CR on the basis-split
    operator C = I + inv(B) N N' inv(B') (src/splitted_normal_matrix.cc:90-117) on the planted-LU basis (synthetic
    factors).  Roofline per CR iteration (= one operator application + the CR vector work):
    algorithmic bytes of SURVEY 8d, 2 (nnz L + nnz U) 12 + 2 nnz(N) 12 + vector terms, against the HIP-event time
    of the CR loop.  CPU baseline: the reference's own ConjugateResiduals over its own BackwardSolve /
    AddNormalProduct / ForwardSolve composed into the operator (oracle/ref_driver.cc; the reference's Prepare and
    _Solve themselves need BASICLU and cannot run here), one core, a bounded number of iterations."""
    A0 = synth.synthetic_lp(m, n, 8, 12345)
    B = synth.planted_lu_basis(A0, offdiag=3, seed=12345)
    st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
    colscale = synth.synthetic_basis_state(B["status"], 1.0, 12345)
    ctx = kkt.KktContext(B["A"])
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)   # first call allocates
    t0 = time.perf_counter()
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.split_rescale(B["status"], colscale)
    resc = time.perf_counter() - t0
    ctx.set_pointer_mode(True)
    a, b = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"])
    x, y = ctx.vector(n + m), ctx.vector(m)
    tol = 0.3 * np.sqrt(st["mu"])
    it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, args.maxiter)
    t0 = time.perf_counter()
    K = 5
    cr = 0.0
    for _ in range(K):
        it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, args.maxiter)
        cr += tm.cr
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / K
    us_iter = cr / K / max(it, 1) * 1e6
    ctx.set_profiling(True)          # HIP events around the three parts of every operator application
    itp, errp, tmp = ctx.kkt_basis_solve_resident(a, b, x, y, tol, args.maxiter)
    ctx.set_profiling(False)
    napply = itp + 1
    lv = ctx.split_levels()
    layouts = ctx.spmv_layout()[0]
    nnzL, nnzU = B["L"].nnz, B["U"].nnz - m            # off-diagonal entries of U
    nb = B["status"] == -1
    AI = B["A"].with_identity()
    nnzN = int(np.diff(AI.p)[nb].sum())
    # four sweeps: every factor entry (4 B index + 8 B value) once per sweep, per unknown and sweep the
    # right-hand side, the result and the diagonal (24 B); N N': two passes over N's entries, the two
    # permuted m-vectors, the n-vector of column products written and read, weights; CR: 9 m-vector passes
    bytes_iter = 2 * (nnzL + nnzU) * 12 + 4 * m * 24 + 2 * nnzN * 12 + 8 * (4 * m + 3 * (n + m)) + 9 * 8 * m
    streamed = 2 * (nnzL + nnzU) * 12 + 4 * m * 32 + 2 * B["A"].nnz * 12
    achieved = bytes_iter / (us_iter * 1e-6) / 1e9
    btraffic, btraffic_note = pmc_traffic("pmc_traffic_basis.json", "traffic_bytes_per_iteration",
                                          "C3 basis path, planted factors: one operator application + CR vector kernels", layouts)
    res = {"solves_per_sec": 1.0 / dt, "ms_per_solve": dt * 1e3, "cr_iterations": it, "errflag": err,
           "levels_Ut_Lt_L_U": lv, "prepare_s": prep, "rescale_s": resc, "us_per_cr_iteration": us_iter,
           "cr_iterations_per_sec": 1e6 / us_iter,
           "us_per_apply_parts": {"backward_pair_Ut_Lt": tmp.solve_Bt / napply * 1e6, "N_Nt": tmp.op / napply * 1e6,
                                  "forward_pair_L_U": tmp.solve_B / napply * 1e6},
           "roofline": {"bound": "hbm", "kernel": "one CR iteration on the split operator = sweep_run_kernel (U', L', L, U) + N N' [%s, %s] + CR vector kernels"
                                                  % tuple(layouts),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes": bytes_iter, "streamed_bytes_estimate": streamed, "traffic": btraffic, "traffic_note": btraffic_note,
                        "dependency_levels": int(sum(lv)),
                        "us_per_level_of_the_sweeps": (tmp.solve_Bt + tmp.solve_B) / napply * 1e6 / max(sum(lv), 1),
                        "note": "the sweeps are bound by the dependency chain (one store-to-load hand-off per level), "
                                "not by bytes: the fraction of the HBM roof is reported next to microseconds per level"},
           "note": "planted (synthetic) LU factors, ~3 off-diagonals per column"}
    ctx.set_pointer_mode(False)
    if not args.no_cpu_baseline:
        res.update(basis_cpu_baseline(ctx, B, AI, colscale, m, n, us_iter))
    ctx.close()
    return res


def bench_lu(kkt, synth, m, n, args, bump=1000):
    """Basis::Factorize + GetLuFactors + SplittedNormalMatrix::Prepare (src/basis.cc:116-166,
    src/splitted_normal_matrix.cc:18-66) on the device for a nearly triangular basis of the C3 model size (planted:
    ~91 % column singletons, ~9 % row singletons, a bump of `bump` rows): ipxk_lu_factorize_basis takes B from the
    resident matrix, ipxk_split_prepare_lu builds the operator from the resident factors.  CPU baseline: the
    repo's restatement of the same method on one core ("port": the reference's LU kernel is BASICLU, absent)."""
    P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 12345)
    G = P["G"]
    ctx = kkt.KktContext(P["A"])
    ctx.lu_factorize_basis(P["basis"], 0.1, download=False)       # first call uploads the plain CSC copy of A
    ctx.split_prepare_lu(P["status"], colscale)
    K = 3
    tf = tp = 0.0
    for _ in range(K):
        t0 = time.perf_counter()
        F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
        t1 = time.perf_counter()
        ctx.split_prepare_lu(P["status"], colscale)
        tf += t1 - t0
        tp += time.perf_counter() - t1
    rhs = np.random.default_rng(1).standard_normal(m)
    x = ctx.solve_dense(rhs, "n")
    import scipy.sparse as sp
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
    resid = float(np.abs(B @ x - rhs).max() / (1.0 + np.abs(x).max()))
    nb = int(G["Bp"][-1])
    res = {"workload": "nearly triangular basis of the %d x %d model (nnz(B) %d): planted column / row singletons and a dense-ish bump of %d rows"
                       % (m, n, nb, bump),
           "factorize_ms": tf / K * 1e3, "prepare_from_resident_factors_ms": tp / K * 1e3,
           "phases_ms": {"singleton_rounds": F["seconds_singletons"] * 1e3, "dense_bump": F["seconds_bump"] * 1e3,
                         "assembly": F["seconds_assemble"] * 1e3},
           "col_singletons": F["col_singletons"], "row_singletons": F["row_singletons"], "bump": F["bump"], "rounds": F["rounds"],
           "nnz_L": F["lnz"], "nnz_U": F["unz"], "fill_factor": (F["lnz"] + F["unz"]) / nb, "levels_Ut_Lt_L_U": ctx.split_levels(),
           "parity": {"solve_dense_residual": resid}}
    # a bump beyond the dense limit (a timing of the torn path at >= 1M rows): the same basis after 200 exchanges at
    # random positions; tearing (the default) and the sparse elimination rounds (IPXK_LU_SPARSE=1), second call each
    try:
        rng = np.random.default_rng(12350)
        basis2 = P["basis"].copy()
        basis2[rng.choice(m, 200, replace=False)] = rng.choice(np.nonzero(P["status"][:n] == -1)[0], 200, replace=False)
        hard = {"workload": "the same basis after 200 exchanges at random positions (singular: the dependent columns are replaced by unit columns)"}
        old_env = os.environ.get("IPXK_LU_SPARSE")
        try:
          for mode, label in (("0", "tearing"), ("1", "elimination_rounds"), (None, "default_policy")):
              if mode is None: os.environ.pop("IPXK_LU_SPARSE", None)
              else: os.environ["IPXK_LU_SPARSE"] = mode
              for _ in range(2):
                  t0 = time.perf_counter()
                  Fh = ctx.lu_factorize_basis(basis2, 0.1, download=False)
                  dt = time.perf_counter() - t0
              hard[label] = {"factorize_ms": dt * 1e3, "fill_factor": (Fh["lnz"] + Fh["unz"]) / nb, "spikes": Fh["spikes"], "sparse_pivots": Fh["sparse_pivots"],
                             "sparse_rounds": Fh["sparse_rounds"], "dense_block": Fh["bump"], "rounds": Fh["rounds"], "dependent": Fh["num_dependent"]}
        finally:
          if old_env is None:
              os.environ.pop("IPXK_LU_SPARSE", None)
          else:
              os.environ["IPXK_LU_SPARSE"] = old_env
        res["beyond_the_dense_limit"] = hard
    except Exception as e:            # noqa: BLE001 -- an auxiliary measurement must not take the bench line down
        res["beyond_the_dense_limit"] = {"error": str(e)[:200]}
    # a basis of the IPM on a random 16000 x 40000 LP (tests/golden/ipm_basis_16000.npz): the policy of round 5 (elimination rounds, then a
    # dense rest on the matrix cores with cooperative panels) against round 4's (tearing / the bump dense as it stands)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "ipm_basis_16000.npz"))
        dim, Bp, Bi, Bx = int(g["dim"]), g["Bp"].astype(np.int64), g["Bi"].astype(np.int64), g["Bx"]
        ipm = {"workload": "a basis of the reference's IPM on general_lp(16000, 40000, 31): nnz(B) %d; the sequential minimum-Markowitz elimination "
                           "of its pattern ends with 22.19 M entries (profiles/r05_lu_fill_study.txt)" % len(Bi)}
        old_env = os.environ.get("IPXK_LU_SPARSE")
        try:
            for mode, label in ((None, "default_policy"), ("t", "round4_policy")):
                if mode is None: os.environ.pop("IPXK_LU_SPARSE", None)
                else: os.environ["IPXK_LU_SPARSE"] = mode
                for _ in range(2):
                    t0 = time.perf_counter()
                    Fi = ctx.lu_factorize(dim, Bp[:-1], Bp[1:], Bi, Bx, 0.1, download=False)
                    dt = time.perf_counter() - t0
                ipm[label] = {"factorize_ms": dt * 1e3, "nnz_L_plus_U": Fi["lnz"] + Fi["unz"], "fill_factor": (Fi["lnz"] + Fi["unz"]) / len(Bi),
                              "dense_block": Fi["bump"], "sparse_pivots": Fi["sparse_pivots"], "sparse_rounds": Fi["sparse_rounds"], "spikes": Fi["spikes"],
                              "dense_block_ms": Fi["seconds_bump"] * 1e3}
        finally:
            if old_env is None: os.environ.pop("IPXK_LU_SPARSE", None)
            else: os.environ["IPXK_LU_SPARSE"] = old_env
        res["ipm_basis_16000"] = ipm
    except Exception as e:            # noqa: BLE001
        res["ipm_basis_16000"] = {"error": str(e)[:200]}
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        Fd = ctx.lu_factorize_basis(P["basis"], 0.1, download=True)
        t0 = time.perf_counter()
        Fo = pyoracle.Oracle().lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        tc = time.perf_counter() - t0
        same = all(np.array_equal(Fd[k], Fo[k]) for k in ("rowperm", "colperm", "dependent")) and \
            all(np.array_equal(getattr(Fd[f], a), getattr(Fo[f], a)) for f in ("L", "U") for a in ("p", "i", "x"))
        res["parity"]["factors_equal_cpu_restatement_bitwise"] = bool(same)
        res["cpu_baseline"] = {"value": 1.0 / tc, "unit": "factorizations/s", "cores": 1, "kind": "port",
                               "sample": "1 x the same factorization by the repo's CPU restatement of the method (%.2f s); the reference's "
                                         "kernel for it, BASICLU, is not in the image" % tc}
        res["gpu_over_cpu"] = tc / (tf / K)
    ctx.close()
    return res


def bench_maxvolume(kkt, synth, m, n, args, entering=300):
    """Maxvolume::RunHeuristic (src/maxvolume.cc:108-337) + the refactorization and Prepare that
    follow it in KKTSolverBasis::_Factorize (src/kkt_solver_basis.cc:46-61), on the device, from the slack basis
    of the C3 model with `entering` structural variables carrying large scaling factors (the first Maxvolume call
    of a solve).  CPU baseline: the repo's restatement on one core ("port": Maxvolume needs ipx::Basis, hence
    BASICLU, and cannot run in the reference here; its dense tableau rows cost O(nnz(A)) per step where the
    reference would take its hypersparse branch)."""
    A = synth.synthetic_lp(m, n, 8, 12345)
    basis, status, colscale = synth.slack_basis_crash_state(m, n, entering, 1.0, 12345)
    ctx = kkt.KktContext(A)
    ctx.lu_factorize_basis(basis, 0.1, download=False)
    ctx.split_prepare_lu(status, colscale)
    t0 = time.perf_counter()
    r = ctx.maxvolume(status, colscale)
    dt = time.perf_counter() - t0
    steps = r["updates"] + r["skipped"]
    res = {"workload": "slack basis of the %d x %d model, %d structural variables with large scaling factors; volume_tol 2, "
                       "maxskip_updates 10, rows_per_slice 10000 (the reference's defaults)" % (m, n, entering),
           "seconds": dt, "updates": r["updates"], "skipped": r["skipped"], "slices": r["slices"], "refused": r["refused"],
           "refactorizations": r["factorizations"], "volinc": r["volinc"], "ms_per_step": dt * 1e3 / max(steps, 1),
           "note": "a step = one candidate column: FindLargest, tableau column, ScaleFtran, and for an exchange the tableau "
                   "row, the eta and the weight update; the time includes the refactorizations (LU + Prepare on the device)"}
    ctx.close()
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        t0 = time.perf_counter()
        B = pyoracle.Oracle().basis(pyoracle.Csc(m, n, A.p, A.i, A.x), basis, status)
        w = B.maxvolume(colscale)
        tc = time.perf_counter() - t0
        res["parity"] = {"same_exchanges_in_the_same_order": bool(np.array_equal(w["exchanges"], r["exchanges"])),
                         "same_final_basis": bool(np.array_equal(B.get()[0], r["basis"])),
                         "volinc_rel_diff": abs(w["volinc"] - r["volinc"]) / max(abs(w["volinc"]), 1e-300)}
        res["cpu_baseline"] = {"value": 1.0 / tc, "unit": "runs/s", "cores": 1, "kind": "port",
                               "sample": "1 x the same Maxvolume run by the repo's CPU restatement (%.1f s)" % tc}
        res["gpu_over_cpu"] = tc / dt
        # the same kind of run against the reference's OWN ipx::Maxvolume on its ipx::Basis (tests/dropin/maxvol_main.cc; the
        # program generates its model itself, same shape; the reference's Basis takes its LU from the device there)
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "test_maxvol_dropin")
        if os.path.exists(exe):
            import re, subprocess
            try:
                pr = subprocess.run([exe, str(m), str(n), str(entering), "12345"], capture_output=True, text=True, timeout=600)
                pat = r"errflag (\d+) updates (\d+) skipped (\d+) slices (\d+) volinc ([-+.\de]+) time ([.\d]+) s"
                mr = re.search("reference: +" + pat, pr.stdout)
                md = re.search("device: +" + pat, pr.stdout)
                if mr and md:
                    res["against_the_reference_itself"] = {
                        "workload": "tests/dropin/maxvol_main.cc %d %d %d 12345: ipx::Maxvolume::RunHeuristic on the reference's ipx::Basis "
                                    "(Forrest-Tomlin updates) next to ipxk_maxvolume, same slack basis / scaling factors / parameters" % (m, n, entering),
                        "reference_seconds_one_core": float(mr.group(6)), "device_seconds": float(md.group(6)),
                        "updates_reference_device": [int(mr.group(2)), int(md.group(2))],
                        "skipped_reference_device": [int(mr.group(3)), int(md.group(3))],
                        "volinc_reference_device": [float(mr.group(5)), float(md.group(5))],
                        "identical_final_basis_and_counts": "IDENTICAL decisions" in pr.stdout,
                        "device_over_reference": float(mr.group(6)) / max(float(md.group(6)), 1e-9)}
            except Exception as exc:
                sys.stderr.write("test_maxvol_dropin did not run (%s)\n" % exc)
    return res


def bench_dropin_lp_solver(m, n):
    """CPU-baseline leg: the reference's own LpSolver (oracle/_ref/test_lp_ref, test_lp_hip: tests/dropin/lp_main.cc) on a 1M x 2M
    synthetic LP, three iterations of the initial IPM (stop_at_switch = 1), once with the reference's KKTSolverDiag on one host core
    and once with KKTSolverDiagHip (host pointers at the boundary, PCIe included): Info::time_kkt_solve of both runs."""
    import re, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    if not os.path.exists(os.path.join(here, "oracle", "_ref", "test_lp_hip")):
        return {"skipped": "oracle/_ref/test_lp_hip not built"}
    try:
        pr = subprocess.run([sys.executable, os.path.join(here, "scripts", "gpu_lp_dropin_c3.py"), str(m), str(n), "3"], capture_output=True,
                            text=True, timeout=900)
        mt = re.search(r"KKT solve time \(Info::time_kkt_solve\): reference ([.\d]+) s, Hip ([.\d]+) s -> ([.\d]+) x;  initial IPM \(time_ipm1\): "
                       r"([.\d]+) / ([.\d]+) s -> [.\d]+ x; kktiter1 (\d+) / (\d+)", pr.stdout)
        if not mt:
            return {"error": (pr.stdout + pr.stderr)[-400:]}
        mc = re.search(r"device models: (-?\d+) ipxk_create for (-?\d+) solver objects", pr.stdout)
        return {"workload": "ipx::LpSolver::Solve on a %d x %d synthetic LP, 3 iterations of the initial IPM (stop_at_switch = 1), through "
                            "ipx::KKTSolverDiag (1 host core) and through ipx::KKTSolverDiagHip (host pointers at the boundary)" % (m, n),
                "time_kkt_solve_reference_s": float(mt.group(1)), "time_kkt_solve_hip_s": float(mt.group(2)),
                "hip_over_reference": float(mt.group(3)), "time_ipm1_reference_s": float(mt.group(4)), "time_ipm1_hip_s": float(mt.group(5)),
                "kktiter1_reference_hip": [int(mt.group(6)), int(mt.group(7))],
                "ipxk_create_calls_over_solver_objects": [int(mc.group(1)), int(mc.group(2))] if mc else None,
                "note": "the solver objects of LpSolver::Solve (src/lp_solver.cc:375,386,457) share ONE device model (HipModel registry); "
                        "ipxk_create builds the layouts on the device (config.model_upload)"}
    except Exception as exc:            # noqa: BLE001
        return {"error": str(exc)}


def basis_cpu_baseline(ctx, B, AI, colscale, m, n, us_iter_gpu, iters=8):
    """`iters` iterations of the reference's plain CR on the operator composed from the reference's own kernels,
    same factors; and the HIP operator / CR against it (apply 1e-12, iterate after `iters` iterations)."""
    from oracle import pyoracle as po
    cs = lambda M: po.Csc(M.nrow, M.ncol, M.p, M.i, M.x)
    orc = po.Oracle()
    t0 = time.perf_counter()
    S = orc.split_prepare(cs(AI), n, cs(B["L"]), cs(B["U"]), B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    pre = S.get()                                     # N (permuted, scaled), scaled U as the reference's Prepare builds them
    t_prep = time.perf_counter() - t0
    rhs = np.random.default_rng(3).standard_normal(m)
    rhs[pre["free_positions"]] = 0.0
    out = {}
    kind = "port"
    if po.ref_available():
        try:
            ref = po.Ref()
            op = ref.split(cs(B["L"]), po.Csc(m, m, B["U"].p, B["U"].i, pre["Ux"]), pre["N"], pre["free_positions"])
            apply_cpu = op.apply
            solve_cpu = lambda: op.cr_solve(rhs, 1e-300, iters)[:3]
            kind = "reference"
        except Exception as exc:
            sys.stderr.write("reference kernels unavailable (%s); using the port\n" % exc)
    if kind == "port":
        apply_cpu = S.apply
        solve_cpu = lambda: orc.cr_solve(S.apply, rhs, 1e-300, None, iters)[:3]
    l_cpu, d_cpu = apply_cpu(rhs)
    l_gpu, d_gpu = ctx.split_apply(rhs)
    t0 = time.perf_counter()
    y_cpu, it_cpu, e_cpu = solve_cpu()
    dt = time.perf_counter() - t0
    y_gpu, it_gpu, e_gpu, _, _ = ctx.cr_solve(rhs, 1e-300, None, iters)
    out["cpu_baseline"] = dict(value=it_cpu / dt, unit="CR iterations/s", cores=1, kind=kind,
                               sample="%d iterations of ConjugateResiduals::Solve (src/conjugate_residuals.cc:14-88) over "
                                      "BackwardSolve / AddNormalProduct / ForwardSolve of the %s composed into the split operator "
                                      "on the same factors (%.2f s per iteration; building the operator on the host took %.1f s)"
                                      % (it_cpu, "reference" if kind == "reference" else "restatement", dt / max(it_cpu, 1), t_prep))
    out["gpu_over_cpu"] = (1e6 / us_iter_gpu) / (it_cpu / dt)
    out["parity_vs_cpu"] = {"apply_relerr": float(np.abs(l_gpu - l_cpu).max() / np.abs(l_cpu).max()),
                            "apply_dot_relerr": float(abs(d_gpu - d_cpu) / abs(d_cpu)),
                            "cr_iterate_relerr_after_%d_iterations" % iters: float(np.abs(y_gpu - y_cpu).max() / np.abs(y_cpu).max()),
                            "iter_errflag_gpu": [int(it_gpu), int(e_gpu)], "iter_errflag_cpu": [int(it_cpu), int(e_cpu)]}
    return out


if __name__ == "__main__":
    main()
