// Maxvolume on the device: Maxvolume::RunHeuristic with its Driver, ScaleFtran and
// FindLargest (reference src/maxvolume.cc:108-153, 179-337) and the part of ipx::Basis they drive -- SolveDense,
// SolveForUpdate, TableauRow, ExchangeIfStable (src/basis.cc:162-330) -- on the basis whose LU factors are
// resident (ipxk_lu_factorize_basis + ipxk_split_prepare_lu).
//
// What an exchange step costs on the CPU is a handful of sparse solves; here every piece is a data-parallel pass
// over vectors that stay in HBM, and the host only reads a block of scalars twice per step to take the
// reference's decisions:
//   * FindLargest: arg max |colweights| over the n+m columns (fixed-tree reduction, first index on ties);
//   * tableau column (FTRAN): the entering column scattered into an m-vector, the two forward sweeps on the
//     unscaled factors (trisolve.hip), then the update etas;
//   * ScaleFtran + the recomputed column weight: one fused reduction over the m positions;
//   * tableau row: e_p through the etas (transposed, last first), the two backward sweeps, then one gather
//     product A' btran masked to the NONBASIC columns (the SpMV of the KKT path; slack columns elementwise);
//   * the update of colweights / colscale / invscale_basic: one elementwise pass.
// The factorization is NOT updated in place: the factors of the last refactorized basis B0 stay fixed (so do the
// level schedules of the sweeps) and every exchange appends a product-form eta, B = B0 E_1 ... E_k with
// E_t = I + (eta_t - e_p) e_p', eta_t the tableau column of the entering variable.  The etas are applied by ONE
// workgroup in a single launch (they are sequential, and short for LP bases); after max_etas exchanges, or when an
// exchange fails the stability test (pivot from the row against pivot from the column, relative 1e-8 -- the role
// of kFtDiagErrorTol in the reference's Forrest-Tomlin update, src/ipx_internal.h:37), the basis is refactorized
// on the device (lu.hip) and the operator rebuilt (Basis::ExchangeIfStable, src/basis.cc:299-306, 318-319).
// Any exact update represents the same matrix, so the decisions are the reference's up to rounding; the CPU
// restatement the tests compare with keeps the same etas.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <exception>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "context.hpp"
#include "spmv_kernels.hpp"
#include "trisolve.hpp"

namespace ipxk {

namespace {

constexpr int kRedGrid = 512;            // workgroups of the two-stage reductions
constexpr int kEtaThreads = 1024;
constexpr double kPivotZeroTol = 1e-7;   // src/maxvolume.h:34

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }
#define IPXK_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// the scalars of one step, written by the device, read by the host
struct Scalars {
    int jn;                  // FindLargest
    double weight;
    int pmax, jb;            // ScaleFtran
    double vmax, weight_recomp, colscale_jn, invscale_pmax, pivot_col;
    int used_pmax, eta_nnz;
    double pivot_row;        // row[jn]
    int eta_total;           // entries of all etas after the last exchange
};

struct Part { double v; int i; double s; int c; };     // per-workgroup partial of the reductions

// ---- FindLargest (src/maxvolume.cc:179-200): first index of the largest |w| ----------------------------------
__global__ __launch_bounds__(kBlock) void mv_argmax_kernel(int64_t N, const double* __restrict__ w, Part* part) {
    __shared__ double sv[kBlock / 64];
    __shared__ int si[kBlock / 64];
    double best = 0.0;
    int bi = INT_MAX;
    IPXK_GRID_STRIDE(j, N) {
        const double a = fabs(w[j]);
        if (a > best || (a == best && a > 0.0 && (int)j < bi)) { best = a; bi = (int)j; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; k++)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        part[blockIdx.x].v = best;
        part[blockIdx.x].i = bi;
    }
}
// one workgroup of kRedGrid threads: thread k holds partial k, the tree keeps (largest value, smallest index)
__device__ __forceinline__ void final_argmax(double& best, int& bi, double* sv, int* si) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 1; k < kRedGrid / 64; k++)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
}
__global__ __launch_bounds__(kRedGrid) void mv_argmax_final_kernel(int nparts, const Part* part, const double* __restrict__ w, Scalars* S) {
    __shared__ double sv[kRedGrid / 64];
    __shared__ int si[kRedGrid / 64];
    double best = (int)threadIdx.x < nparts ? part[threadIdx.x].v : 0.0;
    int bi = (int)threadIdx.x < nparts ? part[threadIdx.x].i : INT_MAX;
    final_argmax(best, bi, sv, si);
    if (threadIdx.x == 0) {
        S->jn = bi == INT_MAX ? 0 : bi;             // all weights zero: index 0, weight 0 (the loop ends)
        S->weight = w[S->jn];
    }
}

// ---- tableau column -----------------------------------------------------------------------------------------
__global__ void mv_scatter_column_kernel(int n, const Scalars* S, const int* __restrict__ Ap, const int* __restrict__ Ai,
                                         const double* __restrict__ Ax, double* __restrict__ rhs) {
    const int j = S->jn;
    if (j >= n) { if (blockIdx.x == 0 && threadIdx.x == 0) rhs[j - n] = 1.0; return; }
    for (int q = Ap[j] + blockIdx.x * blockDim.x + threadIdx.x; q < Ap[j + 1]; q += gridDim.x * blockDim.x) rhs[Ai[q]] = Ax[q];
}
// the etas, B^{-1} direction, oldest first: v_p <- v_p / piv; v_i <- v_i - eta_i v_p
__global__ __launch_bounds__(kEtaThreads) void mv_eta_ftran_kernel(int K, const int* __restrict__ ptr, const int* __restrict__ pos,
                                                                   const double* __restrict__ piv, const int* __restrict__ idx,
                                                                   const double* __restrict__ val, double* v) {
    __shared__ double s_vp;
    for (int t = 0; t < K; t++) {
        if (threadIdx.x == 0) { s_vp = v[pos[t]] / piv[t]; v[pos[t]] = s_vp; }
        __syncthreads();
        const double vp = s_vp;
        for (int e = ptr[t] + threadIdx.x; e < ptr[t + 1]; e += kEtaThreads) v[idx[e]] -= val[e] * vp;
        __syncthreads();
    }
}
// transposed direction, newest first: v_p <- (v_p - sum_i eta_i v_i) / piv
__global__ __launch_bounds__(kEtaThreads) void mv_eta_btran_kernel(int K, const int* __restrict__ ptr, const int* __restrict__ pos,
                                                                   const double* __restrict__ piv, const int* __restrict__ idx,
                                                                   const double* __restrict__ val, double* v) {
    __shared__ double red[kEtaThreads / 64];
    for (int t = K - 1; t >= 0; t--) {
        double sum = 0.0;
        for (int e = ptr[t] + threadIdx.x; e < ptr[t + 1]; e += kEtaThreads) sum += val[e] * v[idx[e]];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int k = 0; k < kEtaThreads / 64; k++) tot += red[k];
            v[pos[t]] = (v[pos[t]] - tot) / piv[t];
        }
        __syncthreads();
    }
}
// ---- the etas as DENSE vectors (round 5) -------------------------------------------------------------------------------------
// The two kernels above walk the etas one after the other -- 6.5 us per eta of 8000 entries, 0.65 ms per application with 100 of
// them, three applications per exchange: a third of Maxvolume's kernel time on the IPM's bases of a 24 000-row LP, whose tableau
// columns fill a third of the vector.  Stored as rows of a dense K x m matrix E (row s = eta s, 0 at its own pivot position), the
// same product form splits into a K x K triangular system for the multipliers and ONE pass over E:
//   forward:  alpha_t piv_t = base_t - sum_{prev(t) < s < t} E[s][pos_t] alpha_s,  base_t = alpha_prev(t) if position pos_t was
//             replaced before (prev(t) = the last such exchange), else v[pos_t];  then
//             v[i] = (alpha_last(i) or v[i]) - sum_{s > last(i)} E[s][i] alpha_s          (last(i): the last exchange at position i)
//   backward: d_t = E[t] . v;  w_t piv_t = cur_t - d_t - sum_{s > t, prev(s) <= t} E[t][pos_s] (w_s - v[pos_s]),  cur_t = w_next(t)
//             if the position is replaced again later, else v[pos_t];  then v[pos_t] = w_t for the first exchange of each position.
// The sums of the forward direction run in the order of the sequential kernel (s ascending, every product rounded before it is
// subtracted, zeros skipped): the same result bit for bit.  The triangular systems are solved by one workgroup, a barrier per
// eta (K <= 1024); T[t][s] = E[s][pos_t] (s < t) is kept both ways round so that either direction reads it contiguously.
constexpr int kEtaDenseMax = 1024;
__global__ void mv_eta_dense_append_kernel(int m, int K, int cap, const Scalars* S, const double* __restrict__ lhs, double* __restrict__ E,
                                           int* pos, double* piv, int* prev, int* next, int* last, double* __restrict__ T, double* __restrict__ Tt) {
    const int pmax = S->pmax;
    IPXK_GRID_STRIDE(p, m) E[(size_t)K * m + p] = (int)p == pmax ? 0.0 : lhs[p];
    IPXK_GRID_STRIDE(t, K) {                         // the older etas at the new pivot position
        const double e = E[(size_t)t * m + pmax];
        T[(size_t)K * cap + t] = e;
        Tt[(size_t)t * cap + K] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pos[K] = pmax;
        piv[K] = lhs[pmax];
        const int pr = last[pmax];
        prev[K] = pr;
        next[K] = -1;
        if (pr >= 0) next[pr] = K;
        last[pmax] = K;
    }
}
// (Blocked since the end of round 5: the 64 etas of a block are solved by ONE wavefront -- the multiplier of a step goes to the later lanes
// by a lane read, no barrier -- and the threads of the later blocks then subtract the block's 64 products in the same order from LDS: one
// workgroup barrier per 64 etas instead of two per eta.  Every r_t still receives its products in the order of the etas: the same bits.)
__global__ __launch_bounds__(kEtaDenseMax) void mv_eta_dense_ftran_solve_kernel(int K, int cap, const double* __restrict__ v, const int* __restrict__ pos,
                                                                               const double* __restrict__ piv, const int* __restrict__ prev,
                                                                               const double* __restrict__ Tt, double* __restrict__ alpha) {
    __shared__ double s_a[2][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pr = t < K ? prev[t] : -1;
    double r = (t < K && pr < 0) ? v[pos[t]] : 0.0;
    const double pv = t < K ? piv[t] : 1.0;
    for (int b0 = 0, blk = 0; b0 < K; b0 += 64, blk++) {
        const int b1 = min(b0 + 64, K);
        double* sa = s_a[blk & 1];
        // (the entries of T a thread needs for 16 steps are fetched together, ahead of the steps: a dependent load per step was most of a step)
        constexpr int CH = 16;
        if (wave == blk) {
            for (int s0 = b0; s0 < b1; s0 += CH) {
                double e[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) e[q] = (s0 + q < b1 && t > s0 + q && t < K) ? Tt[(size_t)(s0 + q) * cap + t] : 0.0;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int s = s0 + q;
                    if (s < b1) {                                       // wave-uniform
                        const double mine = r / pv;                     // (only lane s - b0's value is used)
                        const int hi = __builtin_amdgcn_readlane(__double2hiint(mine), s - b0), lo = __builtin_amdgcn_readlane(__double2loint(mine), s - b0);
                        const double a = __hiloint2double(hi, lo);
                        if (t == s) { alpha[s] = a; sa[s - b0] = a; }
                        if (t > s && t < K) {
                            if (s == pr) r = a;
                            else if (s > pr && e[q] != 0.0) r -= e[q] * a;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (wave > blk && t < K) {
            for (int s0 = b0; s0 < b1; s0 += CH) {
                double e[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) e[q] = s0 + q < b1 ? Tt[(size_t)(s0 + q) * cap + t] : 0.0;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int s = s0 + q;
                    if (s < b1) {
                        const double a = sa[s - b0];
                        if (s == pr) r = a;
                        else if (s > pr && e[q] != 0.0) r -= e[q] * a;
                    }
                }
            }
        }
        // (the other buffer is written next, after everybody has passed this block's barrier: nobody still reads it)
    }
}
__global__ __launch_bounds__(kBlock) void mv_eta_dense_ftran_apply_kernel(int m, int K, const double* __restrict__ E, const double* __restrict__ alpha,
                                                                         const int* __restrict__ last, double* __restrict__ v) {
    __shared__ double sa[kEtaDenseMax];
    for (int t = threadIdx.x; t < K; t += kBlock) sa[t] = alpha[t];
    __syncthreads();
    IPXK_GRID_STRIDE(i, m) {
        const int l = last[i];
        double x = l >= 0 ? sa[l] : v[i];
        // eight entries of the column in flight at a time; the products are still subtracted one after the other in the order of the etas
        // (a thread walked its column one dependent load at a time before: 82 us per application with 400 etas of 24 000 entries)
        int s = l + 1;
        for (; s + 8 <= K; s += 8) {
            double e[8];
#pragma unroll
            for (int q = 0; q < 8; q++) e[q] = E[(size_t)(s + q) * m + i];
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (e[q] != 0.0) x -= e[q] * sa[s + q];
        }
        for (; s < K; s++) {
            const double e = E[(size_t)s * m + i];
            if (e != 0.0) x -= e * sa[s];
        }
        v[i] = x;
    }
}
// d_t = E[t] . v (one workgroup per eta, fixed tree)
__global__ __launch_bounds__(kBlock) void mv_eta_dense_dots_kernel(int m, const double* __restrict__ E, const double* __restrict__ v, double* __restrict__ d) {
    __shared__ double red[kBlock / 64];
    const double* e = E + (size_t)blockIdx.x * m;
    double sum = 0.0;
    for (int i = threadIdx.x; i < m; i += kBlock) sum += e[i] * v[i];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) sum += __shfl_xor(sum, k, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < kBlock / 64; k++) tot += red[k];
        d[blockIdx.x] = tot;
    }
}
// (blocked like the forward solve: the 64 etas of a block by one wavefront, from the last eta down; the earlier threads then add the block's
// products in the same descending order)
__global__ __launch_bounds__(kEtaDenseMax) void mv_eta_dense_btran_solve_kernel(int K, int cap, double* v, const int* __restrict__ pos,
                                                                               const double* __restrict__ piv, const int* __restrict__ prev,
                                                                               const int* __restrict__ next, const double* __restrict__ T,
                                                                               const double* __restrict__ d) {
    __shared__ double s_w[kEtaDenseMax];
    __shared__ double s_diff[2][64];
    __shared__ int s_prev[2][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double cv = t < K ? v[pos[t]] : 0.0;          // the vector as it came in, at this eta's position
    const double dt = t < K ? d[t] : 0.0, pv = t < K ? piv[t] : 1.0;
    const int nx = t < K ? next[t] : -1, myprev = t < K ? prev[t] : -1;
    double acc = 0.0;
    __syncthreads();
    const int nblk = (K + 63) / 64;
    for (int blk = nblk - 1, it = 0; blk >= 0; blk--, it++) {
        const int b0 = blk * 64, b1 = min(b0 + 64, K);
        double* sd = s_diff[it & 1];
        int* sp = s_prev[it & 1];
        constexpr int CH = 16;
        if (wave == blk) {
            for (int s1 = b1 - 1; s1 >= b0; s1 -= CH) {
                double e[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) e[q] = (s1 - q >= b0 && t < s1 - q) ? T[(size_t)(s1 - q) * cap + t] : 0.0;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int s = s1 - q;
                    if (s >= b0) {                                      // wave-uniform
                        double diff_mine = 0.0;
                        if (t == s) {
                            // (s_w[nx], nx > s: written by this wavefront in an earlier step of this loop, or by a later block before its barrier)
                            const double cur = nx >= 0 ? s_w[nx] : cv;
                            const double w = (cur - dt - acc) / pv;
                            s_w[s] = w;
                            diff_mine = w - cv;
                            sd[s - b0] = diff_mine;
                            sp[s - b0] = myprev;
                        }
                        const int hi = __builtin_amdgcn_readlane(__double2hiint(diff_mine), s - b0), lo = __builtin_amdgcn_readlane(__double2loint(diff_mine), s - b0);
                        const double diff = __hiloint2double(hi, lo);
                        const int prs = __builtin_amdgcn_readlane(myprev, s - b0);
                        if (t < s && t >= b0 && e[q] != 0.0 && prs <= t) acc += e[q] * diff;
                    }
                }
            }
        }
        __syncthreads();
        if (wave < blk) {
            for (int s1 = b1 - 1; s1 >= b0; s1 -= CH) {
                double e[CH];
#pragma unroll
                for (int q = 0; q < CH; q++) e[q] = s1 - q >= b0 ? T[(size_t)(s1 - q) * cap + t] : 0.0;
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int s = s1 - q;
                    // (prev[s] <= t: position pos_s is not replaced again between t and s, so eta t meets the value w_s there)
                    if (s >= b0 && e[q] != 0.0 && sp[s - b0] <= t) acc += e[q] * sd[s - b0];
                }
            }
        }
    }
    __syncthreads();
    if (t < K && myprev < 0) v[pos[t]] = s_w[t];
}

// ---- the two triangular systems as MATRICES (for the eta file that stays behind the factors after Maxvolume, see maxvol_apply_etas) ----
// The multipliers are linear in what the solve kernels read: forward  alpha = F vp  (vp = the vector at the positions of the etas that
// are the FIRST at their position, Kd of them), backward  w = G [d; vp].  One workgroup per unit input runs the solve kernel's own
// recurrence (same chains of repeated positions), all unit inputs in parallel; an application inside the CR loop of the KKT solve is
// then two small matrix-vector products instead of K dependent steps with two barriers each (0.55 us per eta: 0.25 ms at K = 450).
// Other rounding than the sequential form (sums in another order), which Maxvolume itself keeps for its decisions.
__global__ __launch_bounds__(kEtaDenseMax) void mv_eta_forward_matrix_kernel(int K, int Kd, int cap, const int* __restrict__ first, const double* __restrict__ piv,
                                                                            const int* __restrict__ prev, const double* __restrict__ Tt,
                                                                            double* __restrict__ F) {
    __shared__ double s_alpha;
    const int t = threadIdx.x, j = blockIdx.x;
    const int pr = t < K ? prev[t] : -1;
    double r = (t < K && t == first[j]) ? 1.0 : 0.0;                // unit input: 1 at the position whose first eta is first[j]
    const double pv = t < K ? piv[t] : 1.0;
    for (int s = 0; s < K; s++) {
        if (t == s) { const double a = r / pv; s_alpha = a; F[(size_t)s * Kd + j] = a; }
        __syncthreads();
        const double a = s_alpha;
        if (t > s && t < K) {
            if (s == pr) r = a;
            else if (s > pr) r -= Tt[(size_t)s * cap + t] * a;
        }
        __syncthreads();
    }
}
// unit input u < K: d = e_u, vp = 0;  u >= K: d = 0, vp = e_(u-K)
__global__ __launch_bounds__(kEtaDenseMax) void mv_eta_backward_matrix_kernel(int K, int Kd, int cap, const int* __restrict__ jof, const double* __restrict__ piv,
                                                                             const int* __restrict__ prev, const int* __restrict__ next,
                                                                             const double* __restrict__ T, double* __restrict__ G) {
    __shared__ double s_diff;
    __shared__ double s_w[kEtaDenseMax];
    const int t = threadIdx.x, u = blockIdx.x, W = K + Kd;
    const double cv = (t < K && u >= K && jof[t] == u - K) ? 1.0 : 0.0;
    const double dt = (t < K && u == t) ? 1.0 : 0.0, pv = t < K ? piv[t] : 1.0;
    const int nx = t < K ? next[t] : -1;
    double acc = 0.0;
    __syncthreads();
    for (int s = K - 1; s >= 0; s--) {
        if (t == s) {
            const double cur = nx >= 0 ? s_w[nx] : cv;
            const double w = (cur - dt - acc) / pv;
            s_w[s] = w;
            s_diff = w - cv;
            G[(size_t)s * W + u] = w;
        }
        __syncthreads();
        if (t < s && prev[s] <= t) acc += T[(size_t)s * cap + t] * s_diff;
        __syncthreads();
    }
}
// alpha[t] = F[t] . vp,  vp[j] = v[pos[first[j]]]   (one wavefront per row)
__global__ __launch_bounds__(kBlock) void mv_eta_forward_gemv_kernel(int K, int Kd, const double* __restrict__ F, const int* __restrict__ first,
                                                                    const int* __restrict__ pos, const double* __restrict__ v, double* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    for (int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < K; t += gridDim.x * (kBlock / 64)) {
        double sum = 0.0;
        for (int j = lane; j < Kd; j += 64) sum += F[(size_t)t * Kd + j] * v[pos[first[j]]];
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) sum += __shfl_xor(sum, k, 64);
        if (lane == 0) alpha[t] = sum;
    }
}
// w[t] = G[t] . [d; vp]
__global__ __launch_bounds__(kBlock) void mv_eta_backward_gemv_kernel(int K, int Kd, const double* __restrict__ G, const int* __restrict__ first,
                                                                     const int* __restrict__ pos, const double* __restrict__ d, const double* __restrict__ v,
                                                                     double* __restrict__ w) {
    const int lane = threadIdx.x & 63, W = K + Kd;
    for (int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < K; t += gridDim.x * (kBlock / 64)) {
        double sum = 0.0;
        for (int u = lane; u < W; u += 64) sum += G[(size_t)t * W + u] * (u < K ? d[u] : v[pos[first[u - K]]]);
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) sum += __shfl_xor(sum, k, 64);
        if (lane == 0) w[t] = sum;
    }
}
// v[pos[first[j]]] = w[first[j]]
__global__ void mv_eta_backward_scatter_kernel(int Kd, const int* __restrict__ first, const int* __restrict__ pos, const double* __restrict__ w,
                                               double* __restrict__ v) {
    IPXK_GRID_STRIDE(j, Kd) v[pos[first[j]]] = w[first[j]];
}

// ---- ScaleFtran (src/maxvolume.cc:322-337) + the recomputed weight (:269-275) + # nonzeros of the column ----
__global__ __launch_bounds__(kBlock) void mv_scale_ftran_kernel(int m, const Scalars* S, const double* __restrict__ lhs,
                                                                const double* __restrict__ colscale, const double* __restrict__ invscale,
                                                                const int* __restrict__ slice_of, int slice, Part* part) {
    __shared__ double sv[kBlock / 64], ss[kBlock / 64];
    __shared__ int si[kBlock / 64], sc[kBlock / 64];
    const double dj = colscale[S->jn];
    double best = 0.0, sum = 0.0;
    int bi = INT_MAX, cnt = 0;
    IPXK_GRID_STRIDE(p, m) {
        const double pivot = lhs[p];
        const double scaled = pivot * dj * invscale[p];
        const double v = fabs(scaled);
        if (fabs(pivot) > kPivotZeroTol && (v > best || (v == best && v > 0.0 && (int)p < bi))) { best = v; bi = (int)p; }
        if (slice_of[p] == slice) sum += scaled;
        cnt += pivot != 0.0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        sum += __shfl_xor(sum, d, 64);
        cnt += __shfl_xor(cnt, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sv[w] = best; si[w] = bi; ss[w] = sum; sc[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; k++) {
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
            sum += ss[k];
            cnt += sc[k];
        }
        part[blockIdx.x] = Part{best, bi, sum, cnt};
    }
}
__global__ __launch_bounds__(kRedGrid) void mv_scale_ftran_final_kernel(int nparts, const Part* part, const double* __restrict__ lhs,
                                            const double* __restrict__ colscale, const double* __restrict__ invscale,
                                            const int* __restrict__ slice_of, int slice, const ipxint* __restrict__ basis, Scalars* S) {
    __shared__ double sv[kRedGrid / 64], ss[kRedGrid / 64];
    __shared__ int si[kRedGrid / 64], sc[kRedGrid / 64];
    const bool have = (int)threadIdx.x < nparts;
    double best = have ? part[threadIdx.x].v : 0.0, sum = have ? part[threadIdx.x].s : 0.0;
    int bi = have ? part[threadIdx.x].i : INT_MAX, cnt = have ? part[threadIdx.x].c : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum += __shfl_xor(sum, d, 64); cnt += __shfl_xor(cnt, d, 64); }
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = sum; sc[threadIdx.x >> 6] = cnt; }
    final_argmax(best, bi, sv, si);                  // (its barrier also publishes ss / sc)
    if (threadIdx.x != 0) return;
    for (int k = 1; k < kRedGrid / 64; k++) { sum += ss[k]; cnt += sc[k]; }
    const int pmax = bi == INT_MAX ? 0 : bi;        // no entry qualified: position 0 (:325, :255-256)
    const double dj = colscale[S->jn];
    S->pmax = pmax;
    S->jb = (int)basis[pmax];
    S->vmax = fabs(lhs[pmax] * dj * invscale[pmax]);
    S->weight_recomp = sum;
    S->colscale_jn = dj;
    S->invscale_pmax = invscale[pmax];
    S->pivot_col = lhs[pmax];
    S->used_pmax = slice_of[pmax] == slice ? 1 : 0;
    S->eta_nnz = cnt;
}
// skipped column (:259-266)
__global__ void mv_skip_kernel(const Scalars* S, double* colweights, double* colscale) {
    colweights[S->jn] = 0.0;
    colscale[S->jn] = 0.0;
}
// ---- tableau row ----------------------------------------------------------------------------------------------
__global__ void mv_unit_kernel(int m, const Scalars* S, double* v) {
    IPXK_GRID_STRIDE(p, m) v[p] = (int)p == S->pmax ? 1.0 : 0.0;
}
__global__ void mv_row_slack_kernel(int m, int n, const double* __restrict__ btran, const double* __restrict__ mask,
                                    double* __restrict__ row) {
    IPXK_GRID_STRIDE(i, m) row[n + i] = mask[n + i] != 0.0 ? btran[i] : 0.0;
}
__global__ void mv_read_pivot_kernel(const double* __restrict__ row, Scalars* S) { S->pivot_row = row[S->jn]; }
// ---- exchange ---------------------------------------------------------------------------------------------------
__global__ void mv_eta_flag_kernel(int m, const Scalars* S, const double* __restrict__ lhs, int* __restrict__ flag) {
    IPXK_GRID_STRIDE(p, m) flag[p] = ((int)p != S->pmax && lhs[p] != 0.0) ? 1 : 0;
}
__global__ void mv_eta_store_kernel(int m, int K, const Scalars* S, const double* __restrict__ lhs, const int* __restrict__ flag,
                                    const int* __restrict__ rank, int* ptr, int* pos, double* piv, int* idx, double* val, Scalars* Sout) {
    const int base = ptr[K];
    IPXK_GRID_STRIDE(p, m) {
        if (flag[p]) { idx[base + rank[p]] = (int)p; val[base + rank[p]] = lhs[p]; }
        if (p == m - 1) {
            ptr[K + 1] = base + rank[p] + flag[p];
            pos[K] = S->pmax;
            piv[K] = lhs[S->pmax];
            Sout->eta_total = base + rank[p] + flag[p];
        }
    }
}
// colweights update (:307-314); colscale / invscale_basic / basis / the NONBASIC mask by the kernel that follows
__global__ void mv_weights_kernel(int64_t N, const Scalars* S, double alpha, const double* __restrict__ row,
                                  const double* __restrict__ colscale, double* __restrict__ colweights) {
    const int jn = S->jn, jb = S->jb;
    const double wjb = (double)S->used_pmax + alpha / S->invscale_pmax;
    IPXK_GRID_STRIDE(j, N) {
        if ((int)j == jb) colweights[j] = wjb;
        else if ((int)j == jn) colweights[j] = 0.0;
        else colweights[j] += alpha * row[j] * colscale[j];
    }
}
__global__ void mv_exchange_kernel(const Scalars* S, ipxint* basis, int* map2basis, double* colscale, double* invscale, double* mask) {
    const int jn = S->jn, jb = S->jb, p = S->pmax;
    colscale[jb] = 1.0 / S->invscale_pmax;          // :301-303
    invscale[p] = 1.0 / S->colscale_jn;
    colscale[jn] = 0.0;
    basis[p] = jn;                                  // Basis::ExchangeIfStable :308-313
    map2basis[jn] = p;
    map2basis[jb] = -1;
    mask[jn] = 0.0;
    mask[jb] = 1.0;
}
// ---- Maxvolume::RunSequential (src/maxvolume.cc:14-106) ---------------------------------------------------------
// the candidate chosen by the host's pass order
__global__ void mvs_set_candidate_kernel(int j, double dj, Scalars* S) { S->jn = j; S->colscale_jn = dj; S->weight = dj; }
// search_pivot (:61-70): first position of the largest v = |x_p| * invscale_basic[p] * d_j; # nonzeros and sum of
// squares of the scaled column (tblnnz, frobnorm_squared)
__global__ __launch_bounds__(kBlock) void mvs_search_pivot_kernel(int m, const Scalars* S, const double* __restrict__ lhs,
                                                                  const double* __restrict__ invscale, Part* part) {
    __shared__ double sv[kBlock / 64], ss[kBlock / 64];
    __shared__ int si[kBlock / 64], sc[kBlock / 64];
    const double dj = S->colscale_jn;
    double best = 0.0, sum = 0.0;
    int bi = INT_MAX, cnt = 0;
    IPXK_GRID_STRIDE(p, m) {
        const double v = fabs(lhs[p]) * invscale[p] * dj;
        if (v > best || (v == best && v > 0.0 && (int)p < bi)) { best = v; bi = (int)p; }
        sum += v * v;
        cnt += v != 0.0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        sum += __shfl_xor(sum, d, 64);
        cnt += __shfl_xor(cnt, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sv[w] = best; si[w] = bi; ss[w] = sum; sc[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; k++) {
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
            sum += ss[k];
            cnt += sc[k];
        }
        part[blockIdx.x] = Part{best, bi, sum, cnt};
    }
}
__global__ __launch_bounds__(kRedGrid) void mvs_search_pivot_final_kernel(int nparts, const Part* part, const double* __restrict__ lhs,
                                                                          const double* __restrict__ invscale, const ipxint* __restrict__ basis,
                                                                          Scalars* S) {
    __shared__ double sv[kRedGrid / 64], ss[kRedGrid / 64];
    __shared__ int si[kRedGrid / 64], sc[kRedGrid / 64];
    const bool have = (int)threadIdx.x < nparts;
    double best = have ? part[threadIdx.x].v : 0.0, sum = have ? part[threadIdx.x].s : 0.0;
    int bi = have ? part[threadIdx.x].i : INT_MAX, cnt = have ? part[threadIdx.x].c : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum += __shfl_xor(sum, d, 64); cnt += __shfl_xor(cnt, d, 64); }
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = sum; sc[threadIdx.x >> 6] = cnt; }
    final_argmax(best, bi, sv, si);
    if (threadIdx.x != 0) return;
    for (int k = 1; k < kRedGrid / 64; k++) { sum += ss[k]; cnt += sc[k]; }
    const int pmax = bi == INT_MAX ? -1 : bi;
    S->pmax = pmax;
    S->vmax = best;
    S->weight_recomp = sum;                            // sum of squares of the scaled column
    S->eta_nnz = cnt;                                  // # nonzeros of the column (its eta has one fewer)
    S->jb = pmax >= 0 ? (int)basis[pmax] : -1;
    S->pivot_col = pmax >= 0 ? lhs[pmax] : 0.0;
    S->invscale_pmax = pmax >= 0 ? invscale[pmax] : 0.0;
    S->used_pmax = 0;
}
// the pivot from the row: btran' a_jn (one workgroup)
__global__ __launch_bounds__(kBlock) void mvs_pivot_row_kernel(int n, const int* __restrict__ Ap, const int* __restrict__ Ai,
                                                               const double* __restrict__ Ax, const double* __restrict__ btran, Scalars* S) {
    __shared__ double red[kBlock / 64];
    const int j = S->jn;
    double sum = 0.0;
    if (j >= n) { if (threadIdx.x == 0) S->pivot_row = btran[j - n]; return; }
    // (sequential order of the column's entries for few entries; a fixed tree over the threads otherwise)
    for (int q = Ap[j] + threadIdx.x; q < Ap[j + 1]; q += kBlock) sum += Ax[q] * btran[Ai[q]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < kBlock / 64; k++) t += red[k]; S->pivot_row = t; }
}
__global__ void mvs_exchange_kernel(const Scalars* S, ipxint* basis, int* map2basis, double* invscale) {
    const int jn = S->jn, jb = S->jb, p = S->pmax;
    invscale[p] = 1.0 / S->colscale_jn;                 // :88
    basis[p] = jn;                                      // Basis::ExchangeIfStable :308-313
    map2basis[jn] = p;
    map2basis[jb] = -1;
}

// ---- set-up -------------------------------------------------------------------------------------------------------
__global__ void mv_init_columns_kernel(int64_t N, const ipxint* __restrict__ status, const double* __restrict__ colscale_in,
                                       double* __restrict__ colscale, double* __restrict__ mask, int* __restrict__ map2basis) {
    IPXK_GRID_STRIDE(j, N) {
        const ipxint st = status[j];
        colscale[j] = st == IPXK_NONBASIC ? colscale_in[j] : 0.0;        // :130-133
        mask[j] = st == IPXK_NONBASIC ? 1.0 : 0.0;                       // TableauRow with ignore_fixed
        map2basis[j] = st == IPXK_NONBASIC_FIXED ? -2 : -1;              // basic ones by mv_init_basis_kernel
    }
}
__global__ void mv_init_basis_kernel(int m, const ipxint* __restrict__ basis, const ipxint* __restrict__ status,
                                     const double* __restrict__ colscale_in, double* __restrict__ invscale, int* __restrict__ map2basis) {
    IPXK_GRID_STRIDE(p, m) {
        const ipxint j = basis[p];
        invscale[p] = status[j] == IPXK_BASIC ? 1.0 / colscale_in[j] : 0.0;   // :120-126 (BASIC_FREE: 0, never leaves)
        map2basis[j] = status[j] == IPXK_BASIC_FREE ? (int)p + m : (int)p;
    }
}
__global__ void mv_slice_work_kernel(int m, const double* __restrict__ invscale, const int* __restrict__ slice_of, int slice,
                                     double* __restrict__ work) {
    IPXK_GRID_STRIDE(p, m) work[p] = slice_of[p] == slice ? invscale[p] : 0.0;     // :221-223
}
__global__ void mv_weights_slack_kernel(int m, int n, const double* __restrict__ work, const double* __restrict__ colscale,
                                        double* __restrict__ colweights) {
    IPXK_GRID_STRIDE(i, m) colweights[n + i] = colscale[n + i] != 0.0 ? work[i] * colscale[n + i] : 0.0;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct MaxvolState {
    DevBuf<double> colscale, invscale, colweights, row, mask, rhs, lhs, unit, btran, work;
    DevBuf<int> map2basis, slice_of, flag, rank, eta_ptr, eta_pos, eta_idx;
    DevBuf<double> eta_piv, eta_val;
    // the etas as dense vectors (mv_eta_dense_*): E [cap][m], T / Tt [cap][cap], multipliers, dots, links between the exchanges of a position
    DevBuf<double> etaE, etaT, etaTt, eta_alpha, eta_d;
    DevBuf<int> eta_prev, eta_next, eta_last;
    DevBuf<double> etaF, etaG, eta_w;      // the triangular systems of a kept eta file as matrices (mv_eta_*_matrix_kernel)
    DevBuf<int> eta_first, eta_jof;
    DevBuf<ipxint> basis, status;
    DevBuf<Part> part;
    DevBuf<Scalars> scalars;
    DevBuf<unsigned char> tmp;
    Scalars* h = nullptr;      // pinned
    // the eta file kept BEHIND the resident factors between two calls (Context::etas_live): what EtaFile needs to go on, and the basis
    // the factors + etas represent (by basis position; the device copy is `basis`)
    struct Saved {
        bool live = false, dense = false, have_history = false;
        int K = 0, cap = 0, m = 0;
        int64_t sparse_used = 0, seg_nnz = 0;
        double overhead_s = 0.0, refactor_s = 0.0;
        long lu_generation = -1;       // of the factors the etas stand behind (a later factorization in the context: no resuming)
        int Kd = 0;                    // > 0: etaF / etaG hold the matrices of the two triangular systems (Kd etas are the first at their position)
    } saved;
    std::vector<ipxint> basis_h;
    ~MaxvolState() { if (h) (void)hipHostFree(h); }
};
void destroy_maxvol(MaxvolState* M) { delete M; }

// The etas of the exchanges since the last refactorization (both Maxvolume variants): dense rows (mv_eta_dense_*) where the
// K x m matrix fits and pays (EtaFile::reset: long vectors with short etas -- the slack bases of a 1M-row model -- keep the lists), else the
// compact lists walked one after the other.  When to refactorize: after max_etas exchanges (the
// reference's update limit, src/maxvolume.cc:318-319) -- or, with max_etas < 0 (what KKTSolverBasisHip passes), when the time
// the etas have cost since the last refactorization reaches the time a refactorization costs, both taken from a MODEL so that a
// run does not depend on the clock: a refactorization 25 ms + 3.5e-13 s x (rows of the dense block)^3 (LU + the block's inverse:
// 0.16 s at 7350 rows, 1.2 s at 15 000), an application of K etas K x (0.6 us + 8 m bytes at 2 TB/s), three applications per
// exchange; at least 100, at most 1024 etas.  (Measured on the 24 000 x 60 000 LP: 40 refactorizations of 0.15 s inside
// Maxvolume with the fixed limit of 100.)
struct EtaFile {
    Context* c;
    MaxvolState& M;
    int m;
    hipStream_t s;
    bool dense = false, dense_possible = false, adaptive = false;
    int cap = 100;                 // most etas the buffers hold
    int limit = 100;               // fixed mode: refactorize after so many
    int64_t sparse_cap = 0, sparse_used = 0;
    int K = 0;
    double overhead_s = 0.0, refactor_s = 0.0;
    int64_t seg_nnz = 0;           // entries of the etas of the current segment (decides the next segment's form)
    bool have_history = false;

    // cost of one eta in one application (seconds): the list kernels spend two workgroup barriers and a dependent load per eta plus
    // its entries through one workgroup; the dense form a barrier of the triangular solve plus the eta's row of E in the one pass
    double cost_list(double nnz) const { return 2.5e-6 + 0.5e-9 * nnz; }
    double cost_dense() const { return 0.6e-6 + 8.0 * (double)m / 2e12; }

    EtaFile(Context* ctx, MaxvolState& state, int rows, ipxint max_etas_in, bool resume = false) : c(ctx), M(state), m(rows), s(ctx->stream) {
        adaptive = max_etas_in < 0;
        limit = (int)std::max<ipxint>(1, max_etas_in > 0 ? max_etas_in : 100);
        static const bool dense_off = getenv("IPXK_MAXVOL_DENSE_ETAS") && getenv("IPXK_MAXVOL_DENSE_ETAS")[0] == '0';
        const int64_t fit = (int64_t(1) << 28) / std::max(m, 1);                   // 2 GiB of etas
        cap = adaptive ? (int)std::min<int64_t>(kEtaDenseMax, std::max<int64_t>(limit, fit)) : limit;
        dense_possible = !dense_off && cap <= kEtaDenseMax && (int64_t)cap <= std::max<int64_t>(fit, 1);
        if (!dense_possible) { cap = limit; adaptive = false; }
        sparse_cap = std::max<int64_t>(4 * (int64_t)m, int64_t(1) << 20);
        M.eta_pos.ensure((size_t)cap); M.eta_piv.ensure((size_t)cap);
        const MaxvolState::Saved& sv = M.saved;
        if (resume && sv.live && sv.cap == cap && sv.m == m && (sv.dense ? dense_possible : true)) {
            // the etas of the previous call are still behind the factors: go on where it stopped
            dense = sv.dense; have_history = sv.have_history; K = sv.K; sparse_used = sv.sparse_used; seg_nnz = sv.seg_nnz;
            overhead_s = sv.overhead_s; refactor_s = sv.refactor_s;
            resumed = true;
        } else {
            reset(0);
        }
    }
    bool resumed = false;
    void save() {
        MaxvolState::Saved& sv = M.saved;
        sv.live = true; sv.dense = dense; sv.have_history = have_history; sv.K = K; sv.cap = cap; sv.m = m; sv.sparse_used = sparse_used;
        sv.seg_nnz = seg_nnz; sv.overhead_s = overhead_s; sv.refactor_s = refactor_s;
        sv.lu_generation = lu_generation(c);
        sv.Kd = 0;
        static const bool matrices_off = getenv("IPXK_MAXVOL_ETA_MATRICES") && getenv("IPXK_MAXVOL_ETA_MATRICES")[0] == '0';
        if (dense && K > 0 && !matrices_off) {
            // the two triangular systems as matrices, for the applications inside the KKT solve (mv_eta_*_matrix_kernel)
            std::vector<int> prev_h((size_t)K), pos_h((size_t)K), first, jof((size_t)K, -1);
            M.eta_prev.download(prev_h.data(), (size_t)K, s);
            M.eta_pos.download(pos_h.data(), (size_t)K, s);
            IPXK_HIP(hipStreamSynchronize(s));
            for (int t = 0; t < K; t++) {
                if (prev_h[t] < 0) { jof[t] = (int)first.size(); first.push_back(t); }
                else jof[t] = jof[prev_h[t]];
            }
            const int Kd = (int)first.size();
            M.eta_first.upload(first, s); M.eta_jof.upload(jof, s);
            M.etaF.ensure((size_t)K * Kd); M.etaG.ensure((size_t)K * (K + Kd)); M.eta_w.ensure((size_t)K);
            hipLaunchKernelGGL(mv_eta_forward_matrix_kernel, dim3(Kd), dim3(std::min(kEtaDenseMax, (K + 63) / 64 * 64)), 0, s, K, Kd, cap, M.eta_first.get(), M.eta_piv.get(),
                               M.eta_prev.get(), M.etaTt.get(), M.etaF.get());
            hipLaunchKernelGGL(mv_eta_backward_matrix_kernel, dim3(K + Kd), dim3(std::min(kEtaDenseMax, (K + 63) / 64 * 64)), 0, s, K, Kd, cap, M.eta_jof.get(), M.eta_piv.get(),
                               M.eta_prev.get(), M.eta_next.get(), M.etaT.get(), M.etaG.get());
            IPXK_HIP(hipStreamSynchronize(s));               // (the host vectors go out of scope)
            IPXK_HIP(hipGetLastError());
            sv.Kd = Kd;
        }
    }
    // what the K etas cost in the solves of one KKT solve (~ 100 CR iterations, one application per direction and iteration, a few dense
    // solves around them) against what a refactorization costs: whether the etas stay behind the factors when Maxvolume is over
    bool worth_keeping() const {
        if (K == 0 || full()) return false;
        // (dense form: the triangular systems go through their matrices inside the KKT solve -- what is left per eta is its row of E)
        const double per_eta = dense ? 8.0 * (double)m / 2e12 + 0.05e-6 : cost_list(K > 0 ? (double)seg_nnz / K : 0.0);
        return 220.0 * ((double)K * per_eta + 40e-6) < refactor_s;          // (40 us: the seven extra launches of an application)
    }
    // after a (re)factorization whose dense block has `block_rows` rows: the next segment's etas as dense rows or as lists, whichever
    // the previous segment's etas would have cost less in (no segment yet: from m alone -- the lists only pay beyond ~ 475 000 rows)
    void reset(int block_rows) {
        if (dense_possible) {
            const double avg = have_history && K > 0 ? (double)seg_nnz / K : 0.0;
            dense = cost_dense() < cost_list(avg);
            static const bool force = getenv("IPXK_MAXVOL_DENSE_ETAS") && getenv("IPXK_MAXVOL_DENSE_ETAS")[0] == '1';
            if (force) dense = true;
        } else {
            dense = false;
        }
        if (K > 0) have_history = true;
        K = 0;
        sparse_used = 0;
        seg_nnz = 0;
        overhead_s = 0.0;
        refactor_s = 0.025 + 3.5e-13 * (double)block_rows * (double)block_rows * (double)block_rows;
        if (dense) {
            M.etaE.ensure((size_t)cap * m); M.etaT.ensure((size_t)cap * cap); M.etaTt.ensure((size_t)cap * cap);
            M.eta_alpha.ensure((size_t)cap); M.eta_d.ensure((size_t)cap);
            M.eta_prev.ensure((size_t)cap); M.eta_next.ensure((size_t)cap);
            M.eta_last.ensure((size_t)m);
            IPXK_HIP(hipMemsetAsync(M.eta_last.get(), 0xff, (size_t)m * sizeof(int), s));
        } else {
            M.flag.ensure((size_t)m); M.rank.ensure((size_t)m);
            M.eta_ptr.ensure((size_t)cap + 1);
            M.eta_idx.ensure((size_t)sparse_cap + m); M.eta_val.ensure((size_t)sparse_cap + m);
            IPXK_HIP(hipMemsetAsync(M.eta_ptr.get(), 0, sizeof(int), s));
        }
    }
    // the eta of the exchange described by *S (pmax) from the tableau column lhs; eta_nnz: its number of nonzeros
    void append(const Scalars* S, const double* lhs, int eta_nnz) {
        const int gm = grid_for(m);
        if (dense) {
            hipLaunchKernelGGL(mv_eta_dense_append_kernel, dim3(gm), dim3(kBlock), 0, s, m, K, cap, S, lhs, M.etaE.get(), M.eta_pos.get(), M.eta_piv.get(),
                               M.eta_prev.get(), M.eta_next.get(), M.eta_last.get(), M.etaT.get(), M.etaTt.get());
        } else {
            hipLaunchKernelGGL(mv_eta_flag_kernel, dim3(gm), dim3(kBlock), 0, s, m, S, lhs, M.flag.get());
            size_t bytes = 0;
            IPXK_HIP(rocprim::exclusive_scan(nullptr, bytes, M.flag.get(), M.rank.get(), 0, (size_t)m, rocprim::plus<int>(), s));
            if (M.tmp.size() < bytes) M.tmp.resize(bytes);
            IPXK_HIP(rocprim::exclusive_scan(M.tmp.get(), bytes, M.flag.get(), M.rank.get(), 0, (size_t)m, rocprim::plus<int>(), s));
            hipLaunchKernelGGL(mv_eta_store_kernel, dim3(gm), dim3(kBlock), 0, s, m, K, S, lhs, M.flag.get(), M.rank.get(), M.eta_ptr.get(), M.eta_pos.get(),
                               M.eta_piv.get(), M.eta_idx.get(), M.eta_val.get(), M.scalars.get());
            sparse_used += eta_nnz;
        }
        K++;
        seg_nnz += eta_nnz;
        overhead_s += 3.0 * (dense ? (double)K * cost_dense() : (double)K * 2.5e-6 + 0.5e-9 * (double)seg_nnz);
    }
    bool full() const {                                                             // NeedFreshFactorization (src/maxvolume.cc:318-319)
        if (K >= cap) return true;
        if (!dense && sparse_used + m > sparse_cap) return true;
        if (adaptive) return K >= 100 && overhead_s >= refactor_s;
        return K >= limit;
    }
    void apply(bool transposed, double* v) { apply_etas(M, m, K, cap, dense, transposed, v, s); }
    // Kd > 0: the triangular systems through their matrices (a kept file inside the KKT solve); 0: the sequential, bit-reproducible form
    static void apply_etas(MaxvolState& M, int m, int K, int cap, bool dense, bool transposed, double* v, hipStream_t s, int Kd = 0) {
        if (K == 0) return;
        if (dense && Kd > 0) {
            const int gk = (K + kBlock / 64 - 1) / (kBlock / 64);
            if (transposed) {
                hipLaunchKernelGGL(mv_eta_dense_dots_kernel, dim3(K), dim3(kBlock), 0, s, m, M.etaE.get(), v, M.eta_d.get());
                hipLaunchKernelGGL(mv_eta_backward_gemv_kernel, dim3(gk), dim3(kBlock), 0, s, K, Kd, M.etaG.get(), M.eta_first.get(), M.eta_pos.get(),
                                   M.eta_d.get(), v, M.eta_w.get());
                hipLaunchKernelGGL(mv_eta_backward_scatter_kernel, dim3(grid_for(Kd)), dim3(kBlock), 0, s, Kd, M.eta_first.get(), M.eta_pos.get(),
                                   M.eta_w.get(), v);
            } else {
                hipLaunchKernelGGL(mv_eta_forward_gemv_kernel, dim3(gk), dim3(kBlock), 0, s, K, Kd, M.etaF.get(), M.eta_first.get(), M.eta_pos.get(), v,
                                   M.eta_alpha.get());
                hipLaunchKernelGGL(mv_eta_dense_ftran_apply_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, K, M.etaE.get(), M.eta_alpha.get(),
                                   M.eta_last.get(), v);
            }
            return;
        }
        // (the one-workgroup solves: only as many wavefronts as there are etas -- their two barriers per eta cost by the wavefront)
        const int solve_threads = std::min(kEtaDenseMax, (K + 63) / 64 * 64);
        if (dense && transposed) {
            hipLaunchKernelGGL(mv_eta_dense_dots_kernel, dim3(K), dim3(kBlock), 0, s, m, M.etaE.get(), v, M.eta_d.get());
            hipLaunchKernelGGL(mv_eta_dense_btran_solve_kernel, dim3(1), dim3(solve_threads), 0, s, K, cap, v, M.eta_pos.get(), M.eta_piv.get(),
                               M.eta_prev.get(), M.eta_next.get(), M.etaT.get(), M.eta_d.get());
        } else if (dense) {
            hipLaunchKernelGGL(mv_eta_dense_ftran_solve_kernel, dim3(1), dim3(solve_threads), 0, s, K, cap, v, M.eta_pos.get(), M.eta_piv.get(),
                               M.eta_prev.get(), M.etaTt.get(), M.eta_alpha.get());
            hipLaunchKernelGGL(mv_eta_dense_ftran_apply_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, K, M.etaE.get(), M.eta_alpha.get(),
                               M.eta_last.get(), v);
        } else if (transposed) {
            hipLaunchKernelGGL(mv_eta_btran_kernel, dim3(1), dim3(kEtaThreads), 0, s, K, M.eta_ptr.get(), M.eta_pos.get(), M.eta_piv.get(), M.eta_idx.get(),
                               M.eta_val.get(), v);
        } else {
            hipLaunchKernelGGL(mv_eta_ftran_kernel, dim3(1), dim3(kEtaThreads), 0, s, K, M.eta_ptr.get(), M.eta_pos.get(), M.eta_piv.get(), M.eta_idx.get(),
                               M.eta_val.get(), v);
        }
    }
};

// ---- the etas behind the resident factors (Context::etas_live) -------------------------------------------------------------
// When Maxvolume is over and its last exchanges are few, the fresh factorization of the final basis that the reference asks for
// (src/kkt_solver_basis.cc:56-61, Basis::GetLuFactors) costs more than carrying the etas through the solves of the KKT solve that
// follows: B_new = B_old E_1 ... E_K, so  inverse(B_new) v = inverse(E_K) ... inverse(E_1) inverse(B_old) v  -- the resident factors and
// the eta file as they stand (the form Basis::SolveDense has after Forrest-Tomlin updates, src/forrest_tomlin.cc:67-78).  The operator
// of trisolve.hip and solve_dense_dev apply the etas through the two functions below; the next call of Maxvolume goes on with the same
// file.  A new operator (ipxk_split_prepare*) ends this state; a new factorization in the context (the LU kernel of the reference's
// Basis may share it) leaves operator and etas as they are -- they do not read the LU state -- but the next Maxvolume starts from fresh factors.
void maxvol_apply_etas(Context* c, bool transposed, double* v) {
    IPXK_REQUIRE(c->maxvol && c->maxvol->saved.live, "no eta file behind the factors");
    MaxvolState& M = *c->maxvol;
    EtaFile::apply_etas(M, M.saved.m, M.saved.K, M.saved.cap, M.saved.dense, transposed, v, c->stream, M.saved.Kd);
}
const ipxint* maxvol_current_basis(Context* c) {
    IPXK_REQUIRE(c->maxvol && c->maxvol->saved.live, "no eta file behind the factors");
    return c->maxvol->basis.get();
}
void maxvol_drop_etas(Context* c) {
    c->etas_live = false;
    if (c->maxvol) c->maxvol->saved.live = false;
}

void maxvolume_dev(Context* c, const ipxint* status_in, const double* colscale_in, const ipxk_maxvolume_params* prm_in,
                   ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info, ipxint* log, ipxint log_cap) {
    const ipxk_maxvolume_params defaults{2.0, 10, 10000, 100};       // include/ipx_parameters.h:69-71
    const ipxk_maxvolume_params* prm = prm_in ? prm_in : &defaults;
    LuView V;
    IPXK_REQUIRE(lu_view(c, &V) && V.from_basis && V.ndep == 0, "maxvolume needs the factorization of the current basis (ipxk_lu_factorize_basis)");
    IPXK_REQUIRE(c->split, "maxvolume needs the operator of the current basis (ipxk_split_prepare_lu)");
    const int m = (int)c->m, n = (int)c->n;
    const int64_t N = (int64_t)n + m;
    IPXK_REQUIRE(m > 0, "empty model");
    hipStream_t s = c->stream;
    if (!c->maxvol) c->maxvol = new MaxvolState;
    MaxvolState& M = *c->maxvol;
    const double t_start = now_s();
    const double volumetol = std::max(prm->volume_tol, 1.0);
    for (DevBuf<double>* b : {&M.colscale, &M.colweights, &M.row, &M.mask}) b->ensure((size_t)N);
    for (DevBuf<double>* b : {&M.invscale, &M.rhs, &M.lhs, &M.unit, &M.btran, &M.work}) b->ensure((size_t)m);
    M.map2basis.ensure((size_t)N); M.slice_of.ensure((size_t)m);
    M.part.ensure(kRedGrid); M.scalars.ensure(1);
    // the etas of the previous call may still stand behind the factors (see maxvol_apply_etas): this call goes on with them, and while it
    // runs it applies them itself -- the solves of trisolve.hip must not
    const bool resume = c->etas_live && M.saved.live && (int64_t)M.basis_h.size() == (int64_t)m && M.saved.lu_generation == lu_generation(c);
    EtaFile etas(c, M, m, prm->max_etas, resume);
    const bool resumed = etas.resumed;
    std::vector<ipxint> basis_h((size_t)m), status_h(status_in, status_in + N);
    if (resumed) basis_h = M.basis_h;
    if (c->etas_live && !resumed) {
        // an eta file this call cannot take over (other parameters): a fresh factorization of the basis it stands for first
        IPXK_REQUIRE((int64_t)M.basis_h.size() == (int64_t)m, "eta file without its basis");
        basis_h = M.basis_h;
        maxvol_drop_etas(c);
        ipxk_lu_info li{};
        lu_factorize_basis(c, basis_h.data(), c->maxvol_pivottol, false, &li);
        IPXK_REQUIRE(li.num_dependent == 0, "the basis behind the eta file is singular");
        split_prepare_lu(c, status_h.data(), colscale_in);
        IPXK_REQUIRE(lu_view(c, &V), "no factorization");
    }
    c->etas_live = false;
    M.saved.live = false;
    // (a call that went on with an eta file and ends by an exception leaves an operator without the etas it needs: it goes, so that what
    // follows fails loudly -- "SplittedNormalMatrix not prepared" -- instead of solving with the wrong basis)
    struct ResumeGuard {
        Context* c; bool armed;
        ~ResumeGuard() { if (armed && std::uncaught_exceptions() > 0 && c->split) { destroy_split(c->split); c->split = nullptr; } }
    } resume_guard{c, resumed};
    if (!resumed) etas.reset(V.bump_size);
    int& K = etas.K;
    if (!M.h) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&M.h), sizeof(Scalars)));

    // host mirrors of basis and status (refactorizations, results)
    if (!resumed) IPXK_HIP(hipMemcpyAsync(basis_h.data(), V.basis, (size_t)m * sizeof(ipxint), hipMemcpyDeviceToHost, s));
    DevBuf<double> colscale_dev;
    colscale_dev.upload(colscale_in, (size_t)N, s);
    M.status.upload(status_in, (size_t)N, s);
    M.basis.ensure((size_t)m);
    if (!resumed) IPXK_HIP(hipMemcpyAsync(M.basis.get(), V.basis, (size_t)m * sizeof(ipxint), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(mv_init_columns_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, N, M.status.get(), colscale_dev.get(),
                       M.colscale.get(), M.mask.get(), M.map2basis.get());
    hipLaunchKernelGGL(mv_init_basis_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, M.basis.get(), M.status.get(), colscale_dev.get(),
                       M.invscale.get(), M.map2basis.get());
    // slices: Sortperm of invscale_basic ascending (value, index), row perm[i] belongs to slice i % num_slices (:138-142)
    std::vector<double> inv_h((size_t)m);
    M.invscale.download(inv_h.data(), (size_t)m, s);
    IPXK_HIP(hipStreamSynchronize(s));
    for (int p = 0; p < m; p++) IPXK_REQUIRE(basis_h[p] >= 0 && basis_h[p] < N && status_h[basis_h[p]] >= 0, "status of a basic variable is not BASIC / BASIC_FREE");
    int num_slices = (int)std::min<int64_t>(m, 5 + std::max<int64_t>(m / std::max<ipxint>(prm->rows_per_slice, 1), 0));
    {
        std::vector<std::pair<double, int>> vi((size_t)m);
        for (int p = 0; p < m; p++) vi[p] = std::make_pair(inv_h[p], p);
        std::sort(vi.begin(), vi.end());
        std::vector<int> slice_of((size_t)m);
        for (int i = 0; i < m; i++) slice_of[vi[i].second] = i % num_slices;
        M.slice_of.upload(slice_of, s);
    }
    IPXK_HIP(hipStreamSynchronize(s));

    const int *Ap = nullptr, *Ai = nullptr;
    const double* Ax = nullptr;
    lu_plain_matrix(c, &Ap, &Ai, &Ax);

    ipxk_maxvolume_info I{};
    I.slices = num_slices;
    auto read_scalars = [&]() {
        IPXK_HIP(hipMemcpyAsync(M.h, M.scalars.get(), sizeof(Scalars), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
    };
    auto apply_etas = [&](bool transposed, double* v) { etas.apply(transposed, v); };
    // the reference's ladder for the LU pivot tolerance (Basis::TightenLuPivotTol, src/basis.cc:490-503); the value is the context's
    double& pivottol = c->maxvol_pivottol;
    auto tighten_pivottol = [&]() {
        if (pivottol <= 0.05) pivottol = 0.1;
        else if (pivottol <= 0.25) pivottol = 0.3;
        else if (pivottol <= 0.5) pivottol = 0.9;
        else return false;
        return true;
    };
    auto refactorize = [&]() {                   // Basis::Factorize (src/basis.cc:116-156) + the operator of the sweeps
        ipxk_lu_info li{};
        lu_factorize_basis(c, basis_h.data(), pivottol, false, &li);
        if (li.num_dependent > 0) {              // Basis::Factorize returns IPX_ERROR_basis_singular (:131-137)
            I.errflag = 301;
            return false;
        }
        split_prepare_lu(c, status_h.data(), colscale_in);
        lu_plain_matrix(c, &Ap, &Ai, &Ax);
        etas.reset((int)li.bump);
        I.factorizations++;
        return true;
    };
    const int gm = grid_for(m), gN = grid_for(N);
    const bool verbose = getenv("IPXK_VERBOSE") != nullptr && getenv("IPXK_VERBOSE")[0] == '2';

    for (int slice = 0; slice < num_slices; slice++) {
        // ---- Driver: column weights of the slice (:221-232)
        hipLaunchKernelGGL(mv_slice_work_kernel, dim3(gm), dim3(kBlock), 0, s, m, M.invscale.get(), M.slice_of.get(), slice, M.unit.get());
        apply_etas(true, M.unit.get());
        solve_dense_dev(c, M.unit.get(), M.work.get(), 'T');
        {
            EpiScale e{{}, M.colscale.get(), M.colweights.get()};
            launch_spmv(c->Acols, M.work.get(), e, nullptr, nullptr, s);
            hipLaunchKernelGGL(mv_weights_slack_kernel, dim3(gm), dim3(kBlock), 0, s, m, n, M.work.get(), M.colscale.get(), M.colweights.get());
        }
        int64_t skipped = 0;
        while (true) {
            // FindLargest, tableau column, ScaleFtran
            hipLaunchKernelGGL(mv_argmax_kernel, dim3(kRedGrid), dim3(kBlock), 0, s, N, M.colweights.get(), M.part.get());
            hipLaunchKernelGGL(mv_argmax_final_kernel, dim3(1), dim3(kRedGrid), 0, s, kRedGrid, M.part.get(), M.colweights.get(), M.scalars.get());
            IPXK_HIP(hipMemsetAsync(M.rhs.get(), 0, (size_t)m * sizeof(double), s));
            hipLaunchKernelGGL(mv_scatter_column_kernel, dim3(4), dim3(kBlock), 0, s, n, M.scalars.get(), Ap, Ai, Ax, M.rhs.get());
            solve_dense_dev(c, M.rhs.get(), M.lhs.get(), 'N');
            apply_etas(false, M.lhs.get());
            hipLaunchKernelGGL(mv_scale_ftran_kernel, dim3(kRedGrid), dim3(kBlock), 0, s, m, M.scalars.get(), M.lhs.get(), M.colscale.get(),
                               M.invscale.get(), M.slice_of.get(), slice, M.part.get());
            hipLaunchKernelGGL(mv_scale_ftran_final_kernel, dim3(1), dim3(kRedGrid), 0, s, kRedGrid, M.part.get(), M.lhs.get(), M.colscale.get(),
                               M.invscale.get(), M.slice_of.get(), slice, M.basis.get(), M.scalars.get());
            read_scalars();
            const Scalars a = *M.h;
            if (verbose)
                fprintf(stderr, "ipxk: maxvolume slice %d: jn %d weight %.3e pmax %d jb %d vmax %.3e (etas %d, updates %lld, skipped %lld)\n", slice,
                        a.jn, a.weight, a.pmax, a.jb, a.vmax, K, (long long)I.updates, (long long)skipped);
            if (a.weight == 0.0) break;                                             // :243-244
            if (c->interrupt && (I.errflag = c->interrupt(c->interrupt_user)) != 0) break;   // :250-251
            if (a.vmax <= volumetol) {                                              // :259-266
                hipLaunchKernelGGL(mv_skip_kernel, dim3(1), dim3(1), 0, s, M.scalars.get(), M.colweights.get(), M.colscale.get());
                if (++skipped > prm->maxskip_updates && prm->maxskip_updates >= 0) break;
                continue;
            }
            // tableau row of the leaving variable (:278-280)
            hipLaunchKernelGGL(mv_unit_kernel, dim3(gm), dim3(kBlock), 0, s, m, M.scalars.get(), M.unit.get());
            apply_etas(true, M.unit.get());
            solve_dense_dev(c, M.unit.get(), M.btran.get(), 'T');
            {
                EpiScale e{{}, M.mask.get(), M.row.get()};
                launch_spmv(c->Acols, M.btran.get(), e, nullptr, nullptr, s);
                hipLaunchKernelGGL(mv_row_slack_kernel, dim3(gm), dim3(kBlock), 0, s, m, n, M.btran.get(), M.mask.get(), M.row.get());
            }
            hipLaunchKernelGGL(mv_read_pivot_kernel, dim3(1), dim3(1), 0, s, M.row.get(), M.scalars.get());
            read_scalars();
            const double pivot = M.h->pivot_row;
            // Basis::ExchangeIfStable (:286-321): the pivot from the row against the pivot from the column
            const bool stable = a.pivot_col != 0.0 && std::abs(a.pivot_col - pivot) <= 1e-8 * std::abs(a.pivot_col);
            if (!stable) {
                // Basis::ExchangeIfStable (src/basis.cc:299-306): on fresh factors the pivot tolerance is tightened
                // first, and only when that is no longer possible the basis is declared too ill conditioned
                I.refused++;
                if (K == 0 && !tighten_pivottol()) { I.errflag = 306; break; }      // IPX_ERROR_basis_too_ill_conditioned
                if (!refactorize()) break;
                continue;                                                           // "try again" (:290-291)
            }
            // the eta of this exchange
            etas.append(M.scalars.get(), M.lhs.get(), a.eta_nnz);
            const double alpha = ((double)a.used_pmax - a.weight_recomp) / (a.colscale_jn * pivot);      // :307
            hipLaunchKernelGGL(mv_weights_kernel, dim3(gN), dim3(kBlock), 0, s, N, M.scalars.get(), alpha, M.row.get(), M.colscale.get(),
                               M.colweights.get());
            hipLaunchKernelGGL(mv_exchange_kernel, dim3(1), dim3(1), 0, s, M.scalars.get(), M.basis.get(), M.map2basis.get(), M.colscale.get(),
                               M.invscale.get(), M.mask.get());
            if (log && I.updates < log_cap) { log[2 * I.updates] = a.jb; log[2 * I.updates + 1] = a.jn; }
            I.updates++;
            I.volinc += std::log2(a.vmax);                                          // :294
            basis_h[(size_t)a.pmax] = a.jn;
            status_h[(size_t)a.jn] = IPXK_BASIC;
            status_h[(size_t)a.jb] = IPXK_NONBASIC;
            if (etas.full())                                                        // NeedFreshFactorization (:318-319)
                if (!refactorize()) break;
        }
        I.skipped += skipped;
        if (I.errflag) break;
    }
    IPXK_HIP(hipStreamSynchronize(s));
    check_sweep_abort(c);
    // the tail of KKTSolverBasis::_Factorize (src/kkt_solver_basis.cc:56-61): a fresh factorization of the final
    // basis and the operator built from it
    // (IPXK_MAXVOL_SKIP_FINAL=1, measurements only: leaves the context with the factors of the last refactorized basis)
    // ... unless carrying the etas through the solves that follow is cheaper (EtaFile::worth_keeping; IPXK_MAXVOL_KEEP_ETAS=0: never,
    // =1: whenever the file is not full): the operator then only learns the new basis and its scaling
    // (a run that ends with an error flag keeps them too: factors + etas stay a consistent operator of the basis reported)
    if (K > 0 && !getenv("IPXK_MAXVOL_SKIP_FINAL")) {
        const char* keep_env = getenv("IPXK_MAXVOL_KEEP_ETAS");
        const bool keep = I.errflag != 0 || (keep_env ? (keep_env[0] == '1' && !etas.full()) : etas.worth_keeping());
        if (keep) {
            etas.save();
            M.basis_h = basis_h;
            c->etas_live = true;
            split_follow_basis(c, M.basis.get(), status_h.data(), colscale_in);
            I.kept_etas = K;
        } else {
            (void)refactorize();
        }
    }
    IPXK_HIP(hipStreamSynchronize(s));
    I.seconds = now_s() - t_start;
    if (basis_out) std::copy(basis_h.begin(), basis_h.end(), basis_out);
    if (status_out) std::copy(status_h.begin(), status_h.end(), status_out);
    if (info) *info = I;
}

// Maxvolume::RunSequential on the device (update_heuristic == 0, src/kkt_solver_basis.cc:47-51): every NONBASIC column in
// decreasing order of its scaling factor gets its tableau column (the forward sweep pair + the etas), the host reads
// one block of scalars per candidate and takes the reference's decisions.  The reference does this with hypersparse
// solves; here a candidate costs two sweeps over all m unknowns, so the sequential variant is for moderate sizes --
// the heuristic (maxvolume_dev) is the one built for 1M rows.
void maxvolume_sequential_dev(Context* c, const ipxint* status_in, const double* colscale_in, double volume_tol, ipxint maxpasses,
                              ipxint max_etas_in, ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info, ipxint* log,
                              ipxint log_cap) {
    LuView V;
    if (c->etas_live && c->maxvol && (int64_t)c->maxvol->basis_h.size() == c->m) {
        // (the sequential variant starts from fresh factors: the basis behind an eta file of the other variant is factorized first)
        std::vector<ipxint> bh = c->maxvol->basis_h;
        maxvol_drop_etas(c);
        ipxk_lu_info li{};
        lu_factorize_basis(c, bh.data(), c->maxvol_pivottol, false, &li);
        IPXK_REQUIRE(li.num_dependent == 0, "the basis behind the eta file is singular");
        split_prepare_lu(c, status_in, colscale_in);
    }
    IPXK_REQUIRE(lu_view(c, &V) && V.from_basis && V.ndep == 0, "maxvolume needs the factorization of the current basis (ipxk_lu_factorize_basis)");
    IPXK_REQUIRE(c->split, "maxvolume needs the operator of the current basis (ipxk_split_prepare_lu)");
    const int m = (int)c->m, n = (int)c->n;
    const int64_t N = (int64_t)n + m;
    IPXK_REQUIRE(m > 0, "empty model");
    hipStream_t s = c->stream;
    if (!c->maxvol) c->maxvol = new MaxvolState;
    MaxvolState& M = *c->maxvol;
    const double t_start = now_s();
    const double volumetol = std::max(volume_tol, 1.0);
    for (DevBuf<double>* b : {&M.invscale, &M.rhs, &M.lhs, &M.unit, &M.btran}) b->ensure((size_t)m);
    M.map2basis.ensure((size_t)N);
    M.part.ensure(kRedGrid); M.scalars.ensure(1);
    EtaFile etas(c, M, m, max_etas_in);
    etas.reset(V.bump_size);
    int& K = etas.K;
    M.colscale.ensure((size_t)N); M.mask.ensure((size_t)N);
    if (!M.h) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&M.h), sizeof(Scalars)));
    std::vector<ipxint> basis_h((size_t)m), status_h(status_in, status_in + N);
    IPXK_HIP(hipMemcpyAsync(basis_h.data(), V.basis, (size_t)m * sizeof(ipxint), hipMemcpyDeviceToHost, s));
    DevBuf<double> colscale_dev;
    colscale_dev.upload(colscale_in, (size_t)N, s);
    M.status.upload(status_in, (size_t)N, s);
    M.basis.ensure((size_t)m);
    IPXK_HIP(hipMemcpyAsync(M.basis.get(), V.basis, (size_t)m * sizeof(ipxint), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(mv_init_columns_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, N, M.status.get(), colscale_dev.get(),
                       M.colscale.get(), M.mask.get(), M.map2basis.get());
    hipLaunchKernelGGL(mv_init_basis_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, M.basis.get(), M.status.get(), colscale_dev.get(),
                       M.invscale.get(), M.map2basis.get());
    IPXK_HIP(hipStreamSynchronize(s));
    for (int p = 0; p < m; p++) IPXK_REQUIRE(basis_h[p] >= 0 && basis_h[p] < N && status_h[basis_h[p]] >= 0, "status of a basic variable is not BASIC / BASIC_FREE");
    const int *Ap = nullptr, *Ai = nullptr;
    const double* Ax = nullptr;
    lu_plain_matrix(c, &Ap, &Ai, &Ax);
    ipxk_maxvolume_info I{};
    double& pivottol = c->maxvol_pivottol;
    auto tighten_pivottol = [&]() {                   // Basis::TightenLuPivotTol, src/basis.cc:490-503
        if (pivottol <= 0.05) pivottol = 0.1;
        else if (pivottol <= 0.25) pivottol = 0.3;
        else if (pivottol <= 0.5) pivottol = 0.9;
        else return false;
        return true;
    };
    auto apply_etas = [&](bool transposed, double* v) { etas.apply(transposed, v); };
    auto refactorize = [&]() {
        ipxk_lu_info li{};
        lu_factorize_basis(c, basis_h.data(), pivottol, false, &li);
        if (li.num_dependent > 0) { I.errflag = 301; return false; }
        split_prepare_lu(c, status_h.data(), colscale_in);
        lu_plain_matrix(c, &Ap, &Ai, &Ax);
        etas.reset((int)li.bump);
        I.factorizations++;
        return true;
    };
    const int gm = grid_for(m);
    ipxint passes = 0;
    // candidates of a pass: Sortperm(n+m, colscale, false) (src/utils.cc:87-104), taken from the back
    std::vector<std::pair<double, ipxint>> cand;
    while ((passes < maxpasses || maxpasses < 0) && !I.errflag) {
        ipxint updates_last = 0;
        cand.resize((size_t)N);
        for (int64_t j = 0; j < N; j++) cand[(size_t)j] = std::make_pair(colscale_in[j], (ipxint)j);
        std::sort(cand.begin(), cand.end());
        while (!cand.empty()) {
            const ipxint j = cand.back().second;
            const double dj = cand.back().first;
            if (dj == 0.0) break;
            if (status_h[(size_t)j] != IPXK_NONBASIC) { cand.pop_back(); continue; }
            if (c->interrupt && (I.errflag = c->interrupt(c->interrupt_user)) != 0) break;   // :52-53
            // tableau column and search_pivot
            hipLaunchKernelGGL(mvs_set_candidate_kernel, dim3(1), dim3(1), 0, s, (int)j, dj, M.scalars.get());
            IPXK_HIP(hipMemsetAsync(M.rhs.get(), 0, (size_t)m * sizeof(double), s));
            hipLaunchKernelGGL(mv_scatter_column_kernel, dim3(4), dim3(kBlock), 0, s, n, M.scalars.get(), Ap, Ai, Ax, M.rhs.get());
            solve_dense_dev(c, M.rhs.get(), M.lhs.get(), 'N');
            apply_etas(false, M.lhs.get());
            hipLaunchKernelGGL(mvs_search_pivot_kernel, dim3(kRedGrid), dim3(kBlock), 0, s, m, M.scalars.get(), M.lhs.get(), M.invscale.get(),
                               M.part.get());
            hipLaunchKernelGGL(mvs_search_pivot_final_kernel, dim3(1), dim3(kRedGrid), 0, s, kRedGrid, M.part.get(), M.lhs.get(),
                               M.invscale.get(), M.basis.get(), M.scalars.get());
            IPXK_HIP(hipMemcpyAsync(M.h, M.scalars.get(), sizeof(Scalars), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            const Scalars a = *M.h;
            if (a.vmax <= volumetol || a.pmax < 0) { I.skipped++; cand.pop_back(); continue; }      // :72-76
            // the BTRAN of the leaving variable (ExchangeIfStable with sys = -1, src/basis.cc:292-293): pivot from the row
            hipLaunchKernelGGL(mv_unit_kernel, dim3(gm), dim3(kBlock), 0, s, m, M.scalars.get(), M.unit.get());
            apply_etas(true, M.unit.get());
            solve_dense_dev(c, M.unit.get(), M.btran.get(), 'T');
            hipLaunchKernelGGL(mvs_pivot_row_kernel, dim3(1), dim3(kBlock), 0, s, n, Ap, Ai, Ax, M.btran.get(), M.scalars.get());
            IPXK_HIP(hipMemcpyAsync(M.h, M.scalars.get(), sizeof(Scalars), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            const double pivot = M.h->pivot_row;
            const bool stable = a.pivot_col != 0.0 && std::abs(a.pivot_col - pivot) <= 1e-8 * std::abs(a.pivot_col);
            if (!stable) {
                I.refused++;
                if (K == 0 && !tighten_pivottol()) { I.errflag = 306; break; }
                if (!refactorize()) break;
                continue;                                                           // "try again" (:86-87)
            }
            etas.append(M.scalars.get(), M.lhs.get(), a.eta_nnz);
            hipLaunchKernelGGL(mvs_exchange_kernel, dim3(1), dim3(1), 0, s, M.scalars.get(), M.basis.get(), M.map2basis.get(), M.invscale.get());
            if (log && I.updates + updates_last < log_cap) { log[2 * (I.updates + updates_last)] = a.jb; log[2 * (I.updates + updates_last) + 1] = j; }
            updates_last++;
            I.volinc += std::log2(a.vmax);                                          // :90
            basis_h[(size_t)a.pmax] = j;
            status_h[(size_t)j] = IPXK_BASIC;
            status_h[(size_t)a.jb] = IPXK_NONBASIC;
            cand.pop_back();
            if (etas.full())
                if (!refactorize()) break;
        }
        I.updates += updates_last;
        passes++;
        if (updates_last == 0) break;
    }
    IPXK_HIP(hipStreamSynchronize(s));
    check_sweep_abort(c);
    if (K > 0 && !I.errflag) (void)refactorize();      // the tail of KKTSolverBasis::_Factorize (src/kkt_solver_basis.cc:56-61)
    IPXK_HIP(hipStreamSynchronize(s));
    I.slices = passes;                                 // (the field reports the passes for this variant)
    I.seconds = now_s() - t_start;
    if (basis_out) std::copy(basis_h.begin(), basis_h.end(), basis_out);
    if (status_out) std::copy(status_h.begin(), status_h.end(), status_out);
    if (info) *info = I;
}

}  // namespace ipxk