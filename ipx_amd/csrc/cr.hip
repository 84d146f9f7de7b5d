// Conjugate Residuals on the device             reference src/conjugate_residuals.cc
//   preconditioned CR (:90-213) for the diag path, plain CR (:14-88) for the basis path.
//
// The whole loop runs on the GPU without a host round trip per iteration:
//  * vectors and the loop scalars (cdot, alpha, beta, iteration count, errflag) live in HBM;
//  * every dot product / norm is produced as per-workgroup partials by the kernel that
//    touches the data anyway and is finished redundantly (bitwise identically) by every
//    workgroup of the consuming kernel -- no atomics, no extra "finalize" launches;
//  * the termination test, the error checks (:139-171) and the every-5th-iteration
//    refresh with its monotonicity check (:186-207) are evaluated by the control kernel,
//    which turns all later kernels into no-ops once `done` is set;
//  * the host enqueues cycles of 5 iterations, stays at most kWindow cycles ahead of the
//    GPU and learns about termination from a flag in mapped host memory.
// Steady state per PCR iteration: 4 launches (control+update, SpMV pass 1, SpMV pass 2,
// direction) touching 12 m-vectors instead of the ~22 of the unfused formulation.
#include <cmath>

#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

constexpr int kWindow = 2;  // cycles (of 5 iterations) the host may run ahead

enum CrMode { kModePcrDiag = 0, kModePcrSmw = 1, kModePlain = 2 };

struct CrVecs {
    int m;
    double* lhs;
    double* residual;
    double* sresidual;   // PCR only
    double* step;
    double* Cstep;
    double* Cres;        // C*sresidual (PCR) or C*residual (plain)
    double* pCstep;      // P*Cstep, SMW mode only
    const double* resscale;
    const double* diag;
};

static int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// out = rhs - sub (sub == nullptr: out = rhs); partial max |resscale .* out|
__global__ __launch_bounds__(kBlock) void residual_init_kernel(int m, const double* __restrict__ rhs,
                                                               const double* __restrict__ sub,
                                                               const double* __restrict__ resscale,
                                                               double* __restrict__ out,
                                                               double* partial) {
    __shared__ double red[kBlock / 64 + 1];
    double mx = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const double r = sub ? rhs[i] - sub[i] : rhs[i];
        out[i] = r;
        mx = MaxOp::apply(mx, fabs(resscale ? resscale[i] * r : r));
    }
    mx = block_reduce<MaxOp>(mx, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

__global__ __launch_bounds__(kBlock) void cr_init_state_kernel(CrState* st, double tol, long long maxiter,
                                                               long long hist_cap, PartRef rsdot, int mode) {
    __shared__ double red[kBlock / 64 + 1];
    const double r = rsdot.p ? reduce_partials<SumOp>(rsdot, red) : 0.0;
    if (threadIdx.x == 0) {
        st->tol = tol;
        st->maxiter = maxiter;
        st->k_started = 0;
        st->k_finished = 0;
        st->cdot[0] = st->cdot[1] = 0.0;
        st->rps[0] = r;
        st->rps[1] = 0.0;
        st->resnorm = 0.0;
        st->iter = 0;
        st->errflag = 0;
        st->done = 0;
        st->hist_cap = hist_cap;
        st->diag_rps_old = st->diag_rps_new = 0.0;
        st->mode = mode;
    }
}

// Loop head + solution update of iteration k = st->k_finished.
template <int MODE>
__global__ __launch_bounds__(kBlock) void cr_control_update_kernel(
    CrState* st, CrVecs v, PartRef res0, PartRef res1, PartRef pdot_ref, PartRef rsdot_ref,
    double* res_next0, double* res_next1, double* hist) {
    if (st->done) return;
    __shared__ double red[kBlock / 64 + 1];
    const long long k = st->k_finished;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    int errflag = -1;  // -1: continue
    double resnorm = st->resnorm;

    if (MODE != kModePlain && k > 0 && k % 5 == 0) {
        // :195-206 monotonicity of residual'*P*residual over the last 5 iterations
        const double rsdot = reduce_partials<SumOp>(rsdot_ref, red);
        const long long c5 = k / 5;
        if (rsdot >= st->rps[(c5 - 1) & 1]) {
            errflag = 204;                                     // IPX_ERROR_cr_no_progress
            if (writer) { st->diag_rps_old = st->rps[(c5 - 1) & 1]; st->diag_rps_new = rsdot; }
        } else if (writer) st->rps[c5 & 1] = rsdot;
    }
    double alpha = 0.0;
    if (errflag < 0) {
        resnorm = reduce_partials<MaxOp>((k & 1) ? res1 : res0, red);
        if (writer && k < st->hist_cap) hist[k] = resnorm;
        const double cdot = st->cdot[k & 1];
        if (resnorm <= st->tol) errflag = 0;
        else if (k == st->maxiter) errflag = 201;             // IPX_ERROR_cr_iter_limit
        else if (cdot <= 0.0) errflag = 202;                  // IPX_ERROR_cr_matrix_not_posdef
        else {
            const double pdot = reduce_partials<SumOp>(pdot_ref, red);
            if (MODE != kModePlain && pdot <= 0.0) errflag = 203;  // IPX_ERROR_cr_precond_not_posdef
            else {
                alpha = cdot / pdot;
                if (!isfinite(alpha)) errflag = 205;          // IPX_ERROR_cr_inf_or_nan
            }
        }
    }
    if (errflag >= 0) {
        if (writer) {
            st->errflag = errflag;
            st->iter = k;
            st->resnorm = resnorm;
            st->done = 1;
        }
        return;
    }

    // :173-175 / :72-73
    double mx = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < v.m; i += gridDim.x * kBlock) {
        const double q = v.Cstep[i];
        v.lhs[i] += alpha * v.step[i];
        const double r = v.residual[i] - alpha * q;
        v.residual[i] = r;
        if (MODE == kModePcrDiag) v.sresidual[i] -= alpha * (q / v.diag[i]);
        if (MODE == kModePcrSmw) v.sresidual[i] -= alpha * v.pCstep[i];
        mx = MaxOp::apply(mx, fabs(v.resscale ? v.resscale[i] * r : r));
    }
    mx = block_reduce<MaxOp>(mx, red);
    if (threadIdx.x == 0) ((k & 1) ? res_next0 : res_next1)[blockIdx.x] = mx;
    if (writer) {
        st->k_started = k + 1;
        st->resnorm = resnorm;
    }
}

// :180-184 / :78-81  new search direction; INIT: step = (s)residual, Cstep = C*(s)residual
template <int MODE, bool INIT>
__global__ __launch_bounds__(kBlock) void cr_direction_kernel(CrState* st, CrVecs v, PartRef cdot_ref,
                                                              double* pdot_partial) {
    if (st->done) return;
    __shared__ double red[kBlock / 64 + 1];
    const long long k = INIT ? -1 : st->k_started - 1;
    const double cdotnew = reduce_partials<SumOp>(cdot_ref, red);
    const double beta = INIT ? 0.0 : cdotnew / st->cdot[k & 1];
    const double* src = MODE == kModePlain ? v.residual : v.sresidual;
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < v.m; i += gridDim.x * kBlock) {
        double p, q;
        if (INIT) {
            p = src[i];
            q = v.Cres[i];
        } else {
            p = src[i] + beta * v.step[i];
            q = v.Cres[i] + beta * v.Cstep[i];
        }
        v.step[i] = p;
        v.Cstep[i] = q;
        if (MODE == kModePcrDiag) acc += (q / v.diag[i]) * q;   // pdot of diagonal_precond.cc:152-154
        if (MODE == kModePlain) acc += q * q;                   // Dot(Cstep,Cstep), :66
    }
    if (MODE != kModePcrSmw) {
        acc = block_reduce<SumOp>(acc, red);
        if (threadIdx.x == 0) pdot_partial[blockIdx.x] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->cdot[(k + 1) & 1] = cdotnew;
        st->k_finished = k + 1;
    }
}

__global__ __launch_bounds__(kBlock) void finalize_scalar_kernel(PartRef ref, int op, double* out) {
    __shared__ double red[kBlock / 64 + 1];
    const double v = op == 1 ? reduce_partials<MaxOp>(ref, red)
                   : op == 2 ? reduce_partials<MinOp>(ref, red) : reduce_partials<SumOp>(ref, red);
    if (threadIdx.x == 0) *out = v;
}

// `done` at the end of a cycle, for the host (mapped pinned memory).
__global__ void snapshot_done_kernel(const CrState* st, int* host_slot) {
    __hip_atomic_store(host_slot, st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Partitioned solves: the ranks must leave the loop in the same cycle whatever happens (every cycle
// enqueues collectives), so what the host reads is the MAXIMUM over the ranks of (done, interrupt flag):
// this rank's pair goes into `flags`, an all-reduce follows, the second kernel hands the result to the host.
__global__ void cycle_flags_kernel(const CrState* st, double interrupt_flag, double* flags) {
    flags[0] = (double)st->done;
    flags[1] = interrupt_flag;
}
__global__ void snapshot_flags_kernel(const double* flags, int* host_done, int* host_interrupt) {
    __hip_atomic_store(host_interrupt, (int)flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_done, (int)flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
constexpr int kDoneRing = kWindow + 2;

static void ensure_comm_buffers(Context* c) {
    if (c->comm_scalars.size() < 64) c->comm_scalars.resize(64);
    if (c->comm_send.size() < (size_t)kNumPartialSlots) {
        c->comm_send.resize(kNumPartialSlots);
        IPXK_HIP(hipMemsetAsync(c->comm_send.get(), 0, sizeof(double) * kNumPartialSlots, c->stream));
    }
    const size_t need = (size_t)c->nranks * kNumPartialSlots;
    if (c->comm_gather.size() < need) c->comm_gather.resize(need);
}

// Scalars of the CR loop across ranks.  Single rank: a consumer kernel reduces the producer's
// per-workgroup partials itself.  Partitioned: one finalize launch turns this rank's partials into
// scalars, one all-gather of the (tiny) scalar table makes every rank's values visible, and the
// consumer reduces over ranks in rank order -- every rank obtains bitwise identical scalars and
// therefore takes identical decisions.
struct Pub {
    Context* c;
    bool multi;
    // row partition only: with partitioned columns every rank computes the same scalars itself
    explicit Pub(Context* ctx) : c(ctx), multi(comm_rows(ctx)) { if (multi) ensure_comm_buffers(ctx); }
    PartRef ref(int slot, int nparts) const {
        if (!multi) return PartRef{c->part(slot), nparts, 1};
        return PartRef{c->comm_gather.get() + slot, c->nranks, kNumPartialSlots};
    }
    // op: 0 sum, 1 max, 2 min
    void publish(int slot0, int n0, int op0, int slot1 = -1, int n1 = 0, int op1 = 0) const {
        if (!multi) return;
        hipLaunchKernelGGL(finalize_scalar_kernel, dim3(1), dim3(kBlock), 0, c->stream,
                           PartRef{c->part(slot0), n0, 1}, op0, c->comm_send.get() + slot0);
        if (slot1 >= 0)
            hipLaunchKernelGGL(finalize_scalar_kernel, dim3(1), dim3(kBlock), 0, c->stream,
                               PartRef{c->part(slot1), n1, 1}, op1, c->comm_send.get() + slot1);
        comm_allgather(c, c->comm_send.get(), c->comm_gather.get(), kNumPartialSlots);
    }
};

PartRef publish_scalar(Context* c, int slot, int count, int op) {
    const Pub pub(c);
    pub.publish(slot, count, op);
    return pub.ref(slot, count);
}

// finalize this rank's partials of `slot` and reduce the scalar over the ranks (op: 0 sum, 1 max, 2 min)
PartRef allreduce_scalar(Context* c, int slot, int count, int op) {
    ensure_comm_buffers(c);
    double* out = c->comm_scalars.get() + 62;
    hipLaunchKernelGGL(finalize_scalar_kernel, dim3(1), dim3(kBlock), 0, c->stream,
                       PartRef{c->part(slot), count, 1}, op, out);
    if (op == 0) comm_allreduce_sum(c, out, 1);
    else if (op == 1) comm_allreduce_max(c, out, 1);
    else comm_allreduce_min(c, out, 1);
    return PartRef{out, 1, 1};
}

double reduce_partials_host(Context* c, int slot, int count, bool is_max) {
    ensure_comm_buffers(c);
    double* out = c->comm_scalars.get() + 63;
    hipLaunchKernelGGL(finalize_scalar_kernel, dim3(1), dim3(kBlock), 0, c->stream,
                       PartRef{c->part(slot), count, 1}, is_max ? 1 : 0, out);
    if (comm_rows(c)) {   // with partitioned columns the vectors, hence the scalar, are replicated
        if (is_max) comm_allreduce_max(c, out, 1);
        else comm_allreduce_sum(c, out, 1);
    }
    double v = 0.0;
    IPXK_HIP(hipMemcpyAsync(&v, out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    IPXK_HIP(hipStreamSynchronize(c->stream));
    return v;
}

static void ensure_workspaces(Context* c) {
    const size_t m = (size_t)(c->m > 0 ? c->m : 1);
    if (c->v_residual.size() != m) {
        c->v_residual.resize(m); c->v_sresidual.resize(m); c->v_step.resize(m);
        c->v_Cstep.resize(m); c->v_Cres.resize(m); c->v_pCstep.resize(m);
    }
    if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
    if (c->state.size() == 0) c->state.resize(1);
    if (!c->h_state) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_state), sizeof(CrState)));
    if (!c->ev_a) { IPXK_HIP(hipEventCreate(&c->ev_a)); IPXK_HIP(hipEventCreate(&c->ev_b)); }
    if (!c->h_cycle_done) {
        IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_cycle_done), 64 * sizeof(int), hipHostMallocMapped));
        IPXK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_cycle_done), c->h_cycle_done, 0));
    }
    while ((int)c->ev_window.size() < kWindow + 1) {
        hipEvent_t e;
        IPXK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_window.push_back(e);
    }
}

// Operator / preconditioner hooks of one solve.
struct CrOps {
    int mode;
    // lhs = C rhs, dot partials into part(kPartCdot); returns their count
    int (*applyC)(Context*, const double* rhs, double* lhs, const int* done);
};

static int applyC_normal(Context* c, const double* rhs, double* lhs, const int* done) {
    int np = 0;
    time_mark(c, kTimeOp, true);
    normal_apply_dev(c, c->W, rhs, lhs, &np, done);
    time_mark(c, kTimeOp, false);
    return np;
}
static int applyC_split(Context* c, const double* rhs, double* lhs, const int* done) {
    return split_apply_dev(c, rhs, lhs, done);
}

template <int MODE>
static CrResult run_cr(Context* c, const CrOps& ops, const double* rhs, double tol,
                       const double* resscale, ipxint maxiter, double* lhs, bool lhs_is_zero,
                       ipxk_interrupt_fn interrupt, void* user, double* hist_host, ipxint hist_cap,
                       ipxk_times* times) {
    ensure_workspaces(c);
    time_start(c, times);
    hipStream_t s = c->stream;
    const int m = (int)c->m;
    if (maxiter < 0) maxiter = (comm_active(c) ? c->m_global : c->m) + 100;   // :114-115
    if (hist_cap < 0) hist_cap = 0;
    if (!hist_host) hist_cap = 0;
    if (c->hist.size() < (size_t)hist_cap + 1) c->hist.resize((size_t)hist_cap + 1);
    CrState* st = c->state.get();
    int* done = &st->done;
    const int g = vec_grid(m);
    const Pub pub(c);
    for (int i = 0; i < 2 * kDoneRing; i++) c->h_cycle_done[i] = 0;   // previous solve is synchronized
    const bool multi = comm_active(c);
    if (multi) ensure_comm_buffers(c);

    CrVecs v;
    v.m = m; v.lhs = lhs; v.residual = c->v_residual.get(); v.sresidual = c->v_sresidual.get();
    v.step = c->v_step.get(); v.Cstep = c->v_Cstep.get(); v.Cres = c->v_Cres.get();
    v.pCstep = c->v_pCstep.get(); v.resscale = resscale; v.diag = c->reord.in_use ? c->reord.diagonal.get() : c->diagonal.get();

    IPXK_HIP(hipEventRecord(c->ev_a, s));

    // ---- initialisation, :117-126 / :33-41 ----
    if (lhs_is_zero) {
        hipLaunchKernelGGL(residual_init_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs,
                           (const double*)nullptr, resscale, v.residual, c->part(kPartRes0));
    } else {
        ops.applyC(c, lhs, v.Cres, nullptr);
        hipLaunchKernelGGL(residual_init_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs,
                           (const double*)v.Cres, resscale, v.residual, c->part(kPartRes0));
    }
    int nrsdot = 0;
    if (MODE != kModePlain) nrsdot = diag_apply_dev(c, v.residual, v.sresidual, kPartRsdot, nullptr);
    if (MODE != kModePlain) pub.publish(kPartRes0, g, 1, kPartRsdot, nrsdot, 0);
    else pub.publish(kPartRes0, g, 1);
    const PartRef res0 = pub.ref(kPartRes0, g), res1 = pub.ref(kPartRes1, g);
    const PartRef rsdot = MODE != kModePlain ? pub.ref(kPartRsdot, nrsdot) : PartRef{nullptr, 0, 1};
    hipLaunchKernelGGL(cr_init_state_kernel, dim3(1), dim3(kBlock), 0, s, st, tol, (long long)maxiter,
                       (long long)hist_cap, rsdot, (int)MODE);
    const double* csrc = MODE == kModePlain ? v.residual : v.sresidual;
    int ncdot = ops.applyC(c, csrc, v.Cres, nullptr);
    pub.publish(kPartCdot, ncdot, 0);
    hipLaunchKernelGGL((cr_direction_kernel<MODE, true>), dim3(g), dim3(kBlock), 0, s, st, v,
                       pub.ref(kPartCdot, ncdot), c->part(kPartPdot));
    int npdot = g;
    if (MODE == kModePcrSmw) npdot = diag_apply_dev(c, v.Cstep, v.pCstep, kPartPdot, nullptr);
    pub.publish(kPartPdot, npdot, 0);
    const PartRef pdot_ref = pub.ref(kPartPdot, npdot);

    // ---- main loop: cycles of 5 iterations ----
    // Control::InterruptCheck (:209 / :84) is polled once per cycle of 5 iterations, not before the first
    // cycle (a system that needs no iteration returns its result whatever the callback says).  Single rank:
    // a nonzero flag ends the loop at once.  Partitioned: the flag travels with the cycle's `done` snapshot
    // through one all-reduce (max), so every rank leaves in the same cycle, kWindow cycles later.
    ipxint interrupt_flag = 0, my_interrupt = 0;
    bool agreed_done = false;
    for (long long cycle = 0;; cycle++) {
        const long long k0 = cycle * 5;
        if (k0 > maxiter) break;
        if (cycle >= kWindow) {
            const long long cw = cycle - kWindow;
            IPXK_HIP(hipEventSynchronize(c->ev_window[cw % (kWindow + 1)]));
            if (*(volatile int*)(c->h_cycle_done + cw % kDoneRing)) { agreed_done = true; break; }
            if (multi) {
                const int agreed = *(volatile int*)(c->h_cycle_done + kDoneRing + cw % kDoneRing);
                if (agreed != 0) { interrupt_flag = agreed; break; }
            }
        }
        if (interrupt && cycle > 0) {
            my_interrupt = interrupt(user);
            if (!multi && my_interrupt != 0) { interrupt_flag = my_interrupt; break; }
        }
        for (long long k = k0; k < k0 + 5 && k <= maxiter; k++) {
            hipLaunchKernelGGL((cr_control_update_kernel<MODE>), dim3(g), dim3(kBlock), 0, s, st, v,
                               res0, res1, pdot_ref, rsdot, c->part(kPartRes0), c->part(kPartRes1),
                               c->hist.get());
            ncdot = ops.applyC(c, csrc, v.Cres, done);
            // the control kernel of iteration k wrote the residual-norm partials of parity (k+1)&1
            pub.publish(kPartCdot, ncdot, 0, ((k + 1) & 1) ? kPartRes1 : kPartRes0, g, 1);
            hipLaunchKernelGGL((cr_direction_kernel<MODE, false>), dim3(g), dim3(kBlock), 0, s, st, v,
                               pub.ref(kPartCdot, ncdot), c->part(kPartPdot));
            if (MODE == kModePcrSmw) diag_apply_dev(c, v.Cstep, v.pCstep, kPartPdot, done);
            if (MODE != kModePlain && (k + 1) % 5 == 0) {
                diag_apply_dev(c, v.residual, v.sresidual, kPartRsdot, done);   // :187-194
                pub.publish(kPartPdot, npdot, 0, kPartRsdot, nrsdot, 0);
            } else {
                pub.publish(kPartPdot, npdot, 0);
            }
        }
        if (multi) {
            double* flags = c->comm_scalars.get() + 56;
            hipLaunchKernelGGL(cycle_flags_kernel, dim3(1), dim3(1), 0, s, st, (double)my_interrupt, flags);
            comm_allreduce_max(c, flags, 2);
            hipLaunchKernelGGL(snapshot_flags_kernel, dim3(1), dim3(1), 0, s, flags, c->d_cycle_done + cycle % kDoneRing,
                               c->d_cycle_done + kDoneRing + cycle % kDoneRing);
        } else {
            hipLaunchKernelGGL(snapshot_done_kernel, dim3(1), dim3(1), 0, s, st, c->d_cycle_done + cycle % kDoneRing);
        }
        IPXK_HIP(hipEventRecord(c->ev_window[cycle % (kWindow + 1)], s));
    }
    IPXK_HIP(hipEventRecord(c->ev_b, s));
    IPXK_HIP(hipMemcpyAsync(c->h_state, st, sizeof(CrState), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
    comm_check(c);

    // the ranks of a partitioned solve take identical decisions (identical scalars); if one of them saw the
    // others finish without finishing itself, the replicas have diverged
    if (multi && agreed_done && !c->h_state->done)
        throw Error(IPXK_E_HIP, "ranks of a partitioned solve diverged (another rank finished the CR loop, this one did not)");
    CrResult res;
    if (c->h_state->done) {
        res.iter = c->h_state->iter;
        res.errflag = c->h_state->errflag;
    } else {
        // interrupted: the iterate of the last finished iteration is in lhs
        res.iter = c->h_state->k_finished;
        res.errflag = interrupt_flag;
    }
    if (hist_cap > 0) {
        long long nh = c->h_state->done && c->h_state->errflag != 204 ? res.iter + 1 : res.iter;
        if (!c->h_state->done) nh = c->h_state->k_started;
        if (nh > hist_cap) nh = hist_cap;
        for (long long i = nh; i < hist_cap; i++) hist_host[i] = std::nan("");
        if (nh > 0) {
            c->hist.download(hist_host, (size_t)nh, s);
            IPXK_HIP(hipStreamSynchronize(s));
        }
    }
    if (times) {
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
        times->cr = ms * 1e-3;
        time_collect(c, times);
    }
    return res;
}

// infinity norm of a device vector (partials + host-side finish; diagnostics only)
__global__ __launch_bounds__(kBlock) void infnorm_partial_kernel(int m, const double* __restrict__ x, double* partial) {
    __shared__ double red[kBlock / 64 + 1];
    double mx = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) mx = MaxOp::apply(mx, fabs(x[i]));
    mx = block_reduce<MaxOp>(mx, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

void cr_diagnostics_dev(Context* c, ipxk_cr_diag* out) {
    *out = ipxk_cr_diag{};
    IPXK_REQUIRE(c->h_state != nullptr && c->state.size() > 0, "no CR run on this context yet");
    const CrState& st = *c->h_state;            // copied back at the end of the run
    out->errflag = st.done ? st.errflag : 0;
    out->iter = st.done ? st.iter : st.k_finished;
    out->maxiter = st.maxiter;
    out->resnorm = st.resnorm;
    out->tol = st.tol;
    out->cdot = st.cdot[out->iter & 1];
    out->rps_old = st.diag_rps_old;
    out->rps_new = st.diag_rps_new;
    if (st.done && st.errflag == 202) {         // the vectors of the run are still in the workspaces
        const int m = (int)c->m, g = vec_grid(m);
        auto norm = [&](const double* v) {
            hipLaunchKernelGGL(infnorm_partial_kernel, dim3(g), dim3(kBlock), 0, c->stream, m, v, c->part(kPartScratch));
            return reduce_partials_host(c, kPartScratch, g, true);
        };
        out->infnorm_residual = norm(c->v_residual.get());
        if (st.mode != kModePlain) out->infnorm_sresidual = norm(c->v_sresidual.get());
    }
}

CrResult pcr_solve_dev(Context* c, const double* rhs, double tol, const double* resscale,
                       ipxint maxiter, double* lhs, bool lhs_is_zero, ipxk_interrupt_fn interrupt,
                       void* user, double* hist_host, ipxint hist_cap, ipxk_times* times) {
    IPXK_REQUIRE(c->normal_prepared, "NormalMatrix not prepared");
    IPXK_REQUIRE(c->diag_factorized, "DiagonalPrecond not factorized");
    CrOps ops{c->kdense > 0 ? kModePcrSmw : kModePcrDiag, applyC_normal};
    if (c->kdense > 0)
        return run_cr<kModePcrSmw>(c, ops, rhs, tol, resscale, maxiter, lhs, lhs_is_zero, interrupt,
                                   user, hist_host, hist_cap, times);
    return run_cr<kModePcrDiag>(c, ops, rhs, tol, resscale, maxiter, lhs, lhs_is_zero, interrupt, user,
                                hist_host, hist_cap, times);
}

CrResult cr_solve_dev(Context* c, const double* rhs, double tol, const double* resscale,
                      ipxint maxiter, double* lhs, bool lhs_is_zero, ipxk_interrupt_fn interrupt,
                      void* user, double* hist_host, ipxint hist_cap, ipxk_times* times) {
    IPXK_REQUIRE(c->split != nullptr, "SplittedNormalMatrix not prepared");
    CrOps ops{kModePlain, applyC_split};
    return run_cr<kModePlain>(c, ops, rhs, tol, resscale, maxiter, lhs, lhs_is_zero, interrupt, user,
                              hist_host, hist_cap, times);
}

}  // namespace ipxk
