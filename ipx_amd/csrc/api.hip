// extern "C" entry points of libipx_kkt_hip.so (see include/ipx_kkt_hip.h).
// Thin glue: argument checks, host<->device staging according to the pointer mode,
// exception -> return-code mapping.  No arithmetic lives here.

#include <algorithm>
#include <map>
#include <thread>

#include "context.hpp"

namespace ipxk {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

Context::~Context() {
    if (split) destroy_split(split);
    if (split_spare) destroy_split(split_spare);
    if (prepare_host) destroy_prepare_host(prepare_host);
    if (lu) destroy_lu(lu);
    if (maxvol) destroy_maxvol(maxvol);
    if (nmat) destroy_nmatrix(nmat);
    comm_destroy(this);
    if (h_state) (void)hipHostFree(h_state);
    if (h_cycle_done) (void)hipHostFree(h_cycle_done);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
    for (hipEvent_t e : ev_window) (void)hipEventDestroy(e);
    for (hipEvent_t e : time_events) (void)hipEventDestroy(e);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

namespace {
struct Staging {
    static constexpr size_t kChunk = size_t(8) << 20;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    void ensure() {
        if (pin[0]) return;
        for (int i = 0; i < 2; i++) {
            IPXK_HIP(hipHostMalloc(&pin[i], kChunk, hipHostMallocDefault));
            IPXK_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
};
// one pair of pinned chunks per (calling thread, device): events and pinned memory belong to the device
// that was current when they were created, and a host thread may drive contexts on several devices
// (independent replicas).  Every entry point binds its context's device before it copies anything.
thread_local std::map<int, Staging> g_staging_by_device;
Staging& staging_for_current_device() {
    int dev = 0;
    IPXK_HIP(hipGetDevice(&dev));
    Staging& st = g_staging_by_device[dev];
    st.ensure();
    return st;
}
}  // namespace

// memcpy between caller memory and a pinned chunk; large chunks are split over a few host threads
// (one core moves ~15 GB/s, a PCIe Gen5 x16 link ~50 GB/s: a single thread would be the bottleneck)
static void chunk_memcpy(void* dst, const void* src, size_t n) {
    constexpr size_t kPerThread = size_t(2) << 20;
    const int nt = (int)std::min<size_t>(4, n / kPerThread);
    if (nt <= 1) { memcpy(dst, src, n); return; }
    const size_t piece = (n / nt + 63) & ~size_t(63);
    std::thread th[3];
    for (int t = 1; t < nt; t++) {
        const size_t off = (size_t)t * piece, len = std::min(piece, n - off);
        th[t - 1] = std::thread([=] { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
    }
    memcpy(dst, src, std::min(piece, n));
    for (int t = 1; t < nt; t++) th[t - 1].join();
}

void staged_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    Staging& st = staging_for_current_device();
    const char* src = static_cast<const char*>(src_host);
    char* dst = static_cast<char*>(dst_dev);
    int i = 0;
    for (size_t off = 0; off < bytes; off += Staging::kChunk, i ^= 1) {
        const size_t n = std::min(Staging::kChunk, bytes - off);
        IPXK_HIP(hipEventSynchronize(st.ev[i]));          // buffer i free again (no-op the first time)
        chunk_memcpy(st.pin[i], src + off, n);
        IPXK_HIP(hipMemcpyAsync(dst + off, st.pin[i], n, hipMemcpyHostToDevice, s));
        IPXK_HIP(hipEventRecord(st.ev[i], s));
    }
}

void staged_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s) {
    Staging& st = staging_for_current_device();
    char* dst = static_cast<char*>(dst_host);
    const char* src = static_cast<const char*>(src_dev);
    // chunk c is copied out of pinned buffer c&1 while chunk c+1 is in flight into the other one
    size_t off_prev = 0, n_prev = 0;
    int i = 0;
    for (size_t off = 0; off < bytes; off += Staging::kChunk, i ^= 1) {
        const size_t n = std::min(Staging::kChunk, bytes - off);
        IPXK_HIP(hipEventSynchronize(st.ev[i]));
        IPXK_HIP(hipMemcpyAsync(st.pin[i], src + off, n, hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipEventRecord(st.ev[i], s));
        if (n_prev) {
            IPXK_HIP(hipEventSynchronize(st.ev[i ^ 1]));
            chunk_memcpy(dst + off_prev, st.pin[i ^ 1], n_prev);
        }
        off_prev = off; n_prev = n;
    }
    if (n_prev) {
        IPXK_HIP(hipEventSynchronize(st.ev[i ^ 1]));
        chunk_memcpy(dst + off_prev, st.pin[i ^ 1], n_prev);
    }
}

void time_start(Context* c, ipxk_times* times) {
    c->timing_active = times != nullptr && c->profile_ops;
    c->time_used = 0;
    c->time_kinds.clear();
}

void time_mark(Context* c, int kind, bool begin) {
    if (!c->timing_active) return;
    if (c->time_used == c->time_events.size()) {
        hipEvent_t e;
        IPXK_HIP(hipEventCreate(&e));
        c->time_events.push_back(e);
    }
    IPXK_HIP(hipEventRecord(c->time_events[c->time_used++], c->stream));
    c->time_kinds.push_back(begin ? kind : -1 - kind);
}

void time_collect(Context* c, ipxk_times* times) {
    if (!c->timing_active || !times) return;
    c->timing_active = false;
    double sum[kNumTimeKinds] = {0, 0, 0, 0};
    size_t open_at[kNumTimeKinds] = {0, 0, 0, 0};
    for (size_t i = 0; i < c->time_used; i++) {
        const int k = c->time_kinds[i];
        if (k >= 0) { open_at[k] = i; continue; }
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, c->time_events[open_at[-1 - k]], c->time_events[i]));
        sum[-1 - k] += ms * 1e-3;
    }
    times->op = sum[kTimeOp];
    times->precond = sum[kTimePrecond];
    times->solve_B = sum[kTimeB];
    times->solve_Bt = sum[kTimeBt];
}

const double* stage_in(Context* c, const double* p, size_t len, DevBuf<double>& buf) {
    if (!p) return nullptr;
    if (c->pointer_mode == IPXK_POINTER_DEVICE) return p;
    if (buf.size() < len) buf.resize(len > 0 ? len : 1);
    buf.upload(p, len, c->stream);
    return buf.get();
}

double* stage_out(Context* c, double* p, size_t len, DevBuf<double>& buf) {
    if (!p) return nullptr;
    if (c->pointer_mode == IPXK_POINTER_DEVICE) return p;
    if (buf.size() < len) buf.resize(len > 0 ? len : 1);
    return buf.get();
}

void finish_out(Context* c, double* user, const double* dev, size_t len) {
    if (!user || c->pointer_mode == IPXK_POINTER_DEVICE) return;
    staged_d2h(user, dev, len * sizeof(double), c->stream);
}

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return IPXK_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return IPXK_E_ALLOC;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return IPXK_E_HIP;
    }
}

static void bind_device(Context* c) { IPXK_HIP(hipSetDevice(c->device)); }

}  // namespace ipxk

using namespace ipxk;

extern "C" {

const char* ipxk_last_error(void) { return g_last_error.c_str(); }

int ipxk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ipxk_create(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai, const double* Ax, int device,
                ipxk_context** out) {
    if (out) *out = nullptr;
    return guarded([&] {
        IPXK_REQUIRE(out != nullptr, "out is NULL");
        IPXK_REQUIRE(m >= 0 && n >= 0, "negative dimension");
        IPXK_REQUIRE(Ap != nullptr && (Ap[n] == 0 || (Ai && Ax)), "matrix argument is NULL");
        int ndev = 0;
        IPXK_HIP(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw Error(IPXK_E_HIP, "no HIP device available (the GPU path has no CPU fallback)");
        IPXK_REQUIRE(device >= 0 && device < ndev, "device ordinal out of range");
        std::unique_ptr<ipxk_context> c(new ipxk_context);
        c->device = device;
        IPXK_HIP(hipSetDevice(device));
        IPXK_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        c->m = m;
        c->n = n;
        build_model(c.get(), Ap, Ai, Ax);
        *out = c.release();
    });
}

void ipxk_destroy(ipxk_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    delete ctx;
}

int ipxk_set_pointer_mode(ipxk_context* c, int mode) {
    return guarded([&] {
        IPXK_REQUIRE(c && (mode == IPXK_POINTER_HOST || mode == IPXK_POINTER_DEVICE), "bad argument");
        c->pointer_mode = mode;
    });
}

int ipxk_set_stream(ipxk_context* c, void* hip_stream) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        bind_device(c);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    });
}

int ipxk_set_profiling(ipxk_context* c, int on) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        c->profile_ops = on != 0;
    });
}

int ipxk_set_interrupt(ipxk_context* c, ipxint (*interrupt)(void*), void* interrupt_user) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        c->interrupt = interrupt;
        c->interrupt_user = interrupt_user;
    });
}

int ipxk_reset_solver_state(ipxk_context* c, double lu_pivottol) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        IPXK_REQUIRE(lu_pivottol <= 1.0, "lu_pivottol must lie in (0,1] (or <= 0 for the default)");
        bind_device(c);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        // what a solver object computed: W / the diagonal preconditioner / resscale, the iterate, the operator of a basis and its
        // factors, N, the tightened pivot tolerance.  The buffers stay (grow-only workspaces); nothing of their content is used again.
        c->W = nullptr;
        c->normal_prepared = c->diag_factorized = c->kkt_diag_factorized = c->it_set = false;
        c->kdense = 0;
        if (c->split) {
            if (c->split_spare) destroy_split(c->split_spare);
            c->split_spare = c->split;
            c->split = nullptr;
        }
        lu_invalidate(c);
        nmatrix_invalidate(c);
        c->maxvol_pivottol = lu_pivottol > 0.0 ? lu_pivottol : 0.1;
        maxvol_drop_etas(c);
        c->interrupt = nullptr;
        c->interrupt_user = nullptr;
        c->profile_ops = false;
        c->pointer_mode = IPXK_POINTER_HOST;
        c->stream = c->own_stream;
    });
}

int ipxk_get_reorder_info(const ipxk_context* c, ipxk_reorder_info* info) {
    return guarded([&] {
        IPXK_REQUIRE(c && info, "NULL argument");
        const Reordered& R = c->reord;
        info->active = R.active ? 1 : 0;
        info->levels = R.levels;
        info->components = R.components;
        info->ms = R.ms;
        info->us_original = R.us_original;
        info->us_reordered = R.us_reordered;
    });
}

int ipxk_get_reordering(ipxk_context* c, ipxint* rowperm, ipxint* colperm) {
    return guarded([&] {
        IPXK_REQUIRE(c, "NULL argument");
        IPXK_REQUIRE(c->reord.rowperm.size() >= (size_t)c->m && c->reord.colperm.size() >= (size_t)c->n, "no renumbering of this model was computed");
        bind_device(c);
        std::vector<int> r((size_t)c->m), q((size_t)c->n);
        c->reord.rowperm.download(r.data(), r.size(), c->stream);
        c->reord.colperm.download(q.data(), q.size(), c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        if (rowperm) for (size_t i = 0; i < r.size(); i++) rowperm[i] = r[i];
        if (colperm) for (size_t j = 0; j < q.size(); j++) colperm[j] = q[j];
    });
}

int ipxk_synchronize(ipxk_context* c) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        bind_device(c);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

ipxint ipxk_num_dense_cols(const ipxk_context* c) { return c ? c->num_dense : 0; }

int ipxk_get_rowwise(const ipxk_context* c, ipxint* ATp, ipxint* ATi, double* ATx) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        IPXK_HIP(hipSetDevice(c->device));
        // the row-wise copy is produced by the library's own Transpose arithmetic (spmv.hip)
        const size_t m = (size_t)c->m, nz = (size_t)c->nnz;
        Context* cc = const_cast<Context*>(static_cast<const Context*>(c));
        ensure_host_model(cc, true);      // a download of the device's row-wise copy (layout_device.hip)
        if (ATp) for (size_t r = 0; r <= m; r++) ATp[r] = c->h_ATp[r];
        if (ATi) for (size_t q = 0; q < nz; q++) ATi[q] = c->h_ATi[q];
        if (ATx) for (size_t q = 0; q < nz; q++) ATx[q] = c->h_ATx[q];
    });
}

// ---- NormalMatrix ------------------------------------------------------------
int ipxk_normal_prepare(ipxk_context* c, const double* W) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        bind_device(c);
        const size_t N = (size_t)(c->n + c->m);
        if (!W) {
            std::vector<double> w(N, 0.0);
            std::fill(w.begin(), w.begin() + c->n, 1.0);
            c->W_own.resize(N);
            c->W_own.upload(w, c->stream);
            IPXK_HIP(hipStreamSynchronize(c->stream));
            c->W = c->W_own.get();
        } else if (c->pointer_mode == IPXK_POINTER_DEVICE) {
            c->W = W;
        } else {
            c->W_own.resize(N);
            c->W_own.upload(W, N, c->stream);
            IPXK_HIP(hipStreamSynchronize(c->stream));
            c->W = c->W_own.get();
        }
        c->normal_prepared = true;
    });
}

int ipxk_normal_apply(ipxk_context* c, const double* rhs, double* lhs, double* rhs_dot_lhs) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs, "NULL argument");
        IPXK_REQUIRE(c->normal_prepared, "NormalMatrix not prepared");
        bind_device(c);
        if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
        const size_t m = (size_t)c->m;
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        int np = 0;
        normal_apply_dev(c, c->W, drhs, dlhs, rhs_dot_lhs ? &np : nullptr, nullptr);
        IPXK_HIP(hipGetLastError());
        if (rhs_dot_lhs) *rhs_dot_lhs = reduce_partials_host(c, kPartCdot, np, false);
        finish_out(c, lhs, dlhs, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

// ---- DiagonalPrecond -----------------------------------------------------------
int ipxk_diag_factorize(ipxk_context* c, const double* W, int precond_dense_cols, ipxint* errflag) {
    return guarded([&] {
        IPXK_REQUIRE(c && errflag, "NULL argument");
        bind_device(c);
        if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
        const size_t N = (size_t)(c->n + c->m);
        const double* dW;
        DevBuf<double> tmp;
        if (!W) {
            std::vector<double> w(N, 0.0);
            std::fill(w.begin(), w.begin() + c->n, 1.0);
            tmp.upload(w, c->stream);
            IPXK_HIP(hipStreamSynchronize(c->stream));
            dW = tmp.get();
        } else if (c->pointer_mode == IPXK_POINTER_DEVICE) {
            dW = W;
        } else {
            tmp.upload(W, N, c->stream);
            dW = tmp.get();
        }
        diag_factorize_dev(c, dW, precond_dense_cols != 0, errflag);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_diag_apply(ipxk_context* c, const double* rhs, double* lhs, double* rhs_dot_lhs) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs, "NULL argument");
        IPXK_REQUIRE(c->diag_factorized, "DiagonalPrecond not factorized");
        bind_device(c);
        const size_t m = (size_t)c->m;
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        const int np = diag_apply_dev(c, drhs, dlhs, kPartScratch, nullptr);
        IPXK_HIP(hipGetLastError());
        if (rhs_dot_lhs) *rhs_dot_lhs = reduce_partials_host(c, kPartScratch, np, false);
        finish_out(c, lhs, dlhs, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_diag_get(const ipxk_context* c, double* diagonal, double* chol) {
    return guarded([&] {
        IPXK_REQUIRE(c && c->diag_factorized, "DiagonalPrecond not factorized");
        IPXK_HIP(hipSetDevice(c->device));
        if (diagonal) c->diagonal.download(diagonal, (size_t)c->m, c->stream);
        if (chol && c->kdense > 0) c->chol.download(chol, (size_t)(c->kdense * c->kdense), c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

// ---- ConjugateResiduals -----------------------------------------------------------
static bool host_is_zero(const double* x, size_t len) {
    for (size_t i = 0; i < len; i++)
        if (x[i] != 0.0) return false;
    return true;
}

int ipxk_pcr_solve(ipxk_context* c, const double* rhs, double tol, const double* resscale,
                   ipxint maxiter, double* lhs, ipxint* iter, ipxint* errflag,
                   ipxk_interrupt_fn interrupt, void* interrupt_user, double* resnorm_hist,
                   ipxint hist_cap, ipxk_times* times) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs && iter && errflag, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m;
        const bool host = c->pointer_mode == IPXK_POINTER_HOST;
        // Infnorm(lhs) == 0 saves one operator application (conjugate_residuals.cc:118-123);
        // with device pointers the general branch is taken (same result: C*0 == 0).
        const bool zero = host && host_is_zero(lhs, m);
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        const double* dscale = stage_in(c, resscale, m, c->v_resscale_in);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        if (host) staged_h2d(dlhs, lhs, m * sizeof(double), c->stream);
        if (times) *times = ipxk_times{};
        CrResult r = pcr_solve_dev(c, drhs, tol, dscale, maxiter, dlhs, zero, interrupt, interrupt_user,
                                   resnorm_hist, hist_cap, times);
        *iter = r.iter;
        *errflag = r.errflag;
        finish_out(c, lhs, dlhs, m);
    });
}

int ipxk_cr_solve(ipxk_context* c, const double* rhs, double tol, const double* resscale,
                  ipxint maxiter, double* lhs, ipxint* iter, ipxint* errflag,
                  ipxk_interrupt_fn interrupt, void* interrupt_user, double* resnorm_hist,
                  ipxint hist_cap, ipxk_times* times) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs && iter && errflag, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m;
        const bool host = c->pointer_mode == IPXK_POINTER_HOST;
        const bool zero = host && host_is_zero(lhs, m);
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        const double* dscale = stage_in(c, resscale, m, c->v_resscale_in);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        if (host) staged_h2d(dlhs, lhs, m * sizeof(double), c->stream);
        if (times) *times = ipxk_times{};
        CrResult r = cr_solve_dev(c, drhs, tol, dscale, maxiter, dlhs, zero, interrupt, interrupt_user,
                                  resnorm_hist, hist_cap, times);
        *iter = r.iter;
        *errflag = r.errflag;
        finish_out(c, lhs, dlhs, m);
        check_sweep_abort(c);
    });
}

// ---- KKTSolverDiag ------------------------------------------------------------------
int ipxk_kkt_diag_factorize(ipxk_context* c, const double* xl, const double* xu, const double* zl,
                            const double* zu, double mu, int precond_dense_cols, ipxint* errflag) {
    return guarded([&] {
        IPXK_REQUIRE(c && errflag, "NULL argument");
        IPXK_REQUIRE(!xl || (xu && zl && zu), "iterate vectors must be all NULL or all given");
        bind_device(c);
        const size_t N = (size_t)(c->n + c->m);
        DevBuf<double> t0, t1, t2, t3;
        const double* dxl = stage_in(c, xl, N, t0);
        const double* dxu = stage_in(c, xu, N, t1);
        const double* dzl = stage_in(c, zl, N, t2);
        const double* dzu = stage_in(c, zu, N, t3);
        kkt_diag_factorize_dev(c, dxl, dxu, dzl, dzu, mu, precond_dense_cols != 0, errflag);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_kkt_diag_solve(ipxk_context* c, const double* a, const double* b, double tol, ipxint maxiter,
                        double* x, double* y, ipxint* iter, ipxint* errflag,
                        ipxk_interrupt_fn interrupt, void* interrupt_user, ipxk_times* times) {
    return guarded([&] {
        IPXK_REQUIRE(c && a && b && x && y && iter && errflag, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* da = stage_in(c, a, N, c->k_a);
        const double* db = stage_in(c, b, m, c->k_b);
        double* dx = stage_out(c, x, N, c->k_x);
        double* dy = stage_out(c, y, m, c->k_y);
        if (times) *times = ipxk_times{};
        CrResult r = kkt_diag_solve_dev(c, da, db, tol, maxiter, dx, dy, interrupt, interrupt_user, times);
        *iter = r.iter;
        *errflag = r.errflag;
        finish_out(c, x, dx, N);
        finish_out(c, y, dy, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_kkt_diag_get(const ipxk_context* c, double* W, double* resscale) {
    return guarded([&] {
        IPXK_REQUIRE(c && c->kkt_diag_factorized, "KKTSolverDiag not factorized");
        IPXK_HIP(hipSetDevice(c->device));
        if (W) c->W_own.download(W, (size_t)(c->n + c->m), c->stream);
        if (resscale) c->resscale.download(resscale, (size_t)c->m, c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

// ---- IPM::SolveNewtonSystem -----------------------------------------------------------------
int ipxk_newton_solve(ipxk_context* c, int use_basis, const double* rb, const double* rc, const double* rl,
                      const double* ru, const double* sl, const double* su, const double* xl, const double* xu,
                      const double* zl, const double* zu, const unsigned char* state, double tol, ipxint maxiter,
                      double* dx, double* dxl, double* dxu, double* dy, double* dzl, double* dzu, ipxint* iter,
                      ipxint* errflag, ipxk_interrupt_fn interrupt, void* interrupt_user, ipxk_times* times) {
    return guarded([&] {
        IPXK_REQUIRE(c && sl && su && xl && xu && zl && zu && state && dx && dxl && dxu && dy && dzl && dzu &&
                     iter && errflag, "NULL argument");
        IPXK_REQUIRE(use_basis ? c->split != nullptr : c->kkt_diag_factorized, "KKT solver not factorized");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* in[10] = {rb, rc, rl, ru, sl, su, xl, xu, zl, zu};
        const double* din[10];
        for (int k = 0; k < 10; k++) din[k] = in[k] ? stage_in(c, in[k], k == 0 ? m : N, c->nw_in[k]) : nullptr;
        const unsigned char* dstate = state;
        if (c->pointer_mode == IPXK_POINTER_HOST) {
            c->nw_state.upload(state, N, c->stream);
            dstate = c->nw_state.get();
        }
        double* out[6] = {dx, dxl, dxu, dy, dzl, dzu};
        double* dout[6];
        for (int k = 0; k < 6; k++) dout[k] = stage_out(c, out[k], k == 3 ? m : N, c->nw_out[k]);
        if (times) *times = ipxk_times{};
        CrResult r = newton_solve_dev(c, use_basis != 0, din[0], din[1], din[2], din[3], din[4], din[5], din[6],
                                      din[7], din[8], din[9], dstate, tol, maxiter, dout[0], dout[1], dout[2],
                                      dout[3], dout[4], dout[5], interrupt, interrupt_user, times);
        *iter = r.iter;
        *errflag = r.errflag;
        for (int k = 0; k < 6; k++) finish_out(c, out[k], dout[k], k == 3 ? m : N);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

// ---- the IPM iterate -------------------------------------------------------------------------
static void copy_in(Context* c, void* dst_dev, const void* src, size_t bytes) {
    if (c->pointer_mode == IPXK_POINTER_HOST) staged_h2d(dst_dev, src, bytes, c->stream);
    else IPXK_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyDeviceToDevice, c->stream));
}
static void copy_out(Context* c, void* dst, const void* src_dev, size_t bytes) {
    if (c->pointer_mode == IPXK_POINTER_HOST) staged_d2h(dst, src_dev, bytes, c->stream);
    else IPXK_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
}

int ipxk_iterate_set(ipxk_context* c, const double* x, const double* xl, const double* xu, const double* y,
                     const double* zl, const double* zu, const unsigned char* state) {
    return guarded([&] {
        IPXK_REQUIRE(c && x && xl && xu && y && zl && zu && state, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        DevBuf<double>* dst[6] = {&c->it_x, &c->it_xl, &c->it_xu, &c->it_y, &c->it_zl, &c->it_zu};
        const double* src[6] = {x, xl, xu, y, zl, zu};
        for (int k = 0; k < 6; k++) {
            const size_t len = k == 3 ? m : N;
            dst[k]->resize(std::max<size_t>(len, 1));
            copy_in(c, dst[k]->get(), src[k], len * sizeof(double));
        }
        c->it_state.resize(std::max<size_t>(N, 1));
        copy_in(c, c->it_state.get(), state, N);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        c->it_set = true;
    });
}

int ipxk_iterate_get(ipxk_context* c, double* x, double* xl, double* xu, double* y, double* zl, double* zu) {
    return guarded([&] {
        IPXK_REQUIRE(c && c->it_set, "no iterate on the device (ipxk_iterate_set)");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const DevBuf<double>* src[6] = {&c->it_x, &c->it_xl, &c->it_xu, &c->it_y, &c->it_zl, &c->it_zu};
        double* dst[6] = {x, xl, xu, y, zl, zu};
        for (int k = 0; k < 6; k++)
            if (dst[k]) copy_out(c, dst[k], src[k]->get(), (k == 3 ? m : N) * sizeof(double));
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_iterate_update(ipxk_context* c, double sp, const double* dx, const double* dxl, const double* dxu,
                        double sd, const double* dy, const double* dzl, const double* dzu) {
    return guarded([&] {
        IPXK_REQUIRE(c, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* in[6] = {dx, dxl, dxu, dy, dzl, dzu};
        const double* din[6];
        for (int k = 0; k < 6; k++) din[k] = in[k] ? stage_in(c, in[k], k == 3 ? m : N, c->nw_out[k]) : nullptr;
        iterate_update_dev(c, sp, din[0], din[1], din[2], sd, din[3], din[4], din[5]);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_iterate_residuals(ipxk_context* c, const double* b, const double* cc, const double* lb, const double* ub,
                           double* rb, double* rc, double* rl, double* ru, double* presidual,
                           double* dresidual) {
    return guarded([&] {
        IPXK_REQUIRE(c && b && cc && lb && ub && rb && rc && rl && ru, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* db = stage_in(c, b, m, c->nw_in[0]);
        const double* dc = stage_in(c, cc, N, c->nw_in[1]);
        const double* dlb = stage_in(c, lb, N, c->nw_in[2]);
        const double* dub = stage_in(c, ub, N, c->nw_in[3]);
        double* drb = stage_out(c, rb, m, c->nw_out[0]);
        double* drc = stage_out(c, rc, N, c->nw_out[1]);
        double* drl = stage_out(c, rl, N, c->nw_out[2]);
        double* dru = stage_out(c, ru, N, c->nw_out[4]);
        iterate_residuals_dev(c, db, dc, dlb, dub, drb, drc, drl, dru, presidual, dresidual);
        finish_out(c, rb, drb, m);
        finish_out(c, rc, drc, N);
        finish_out(c, rl, drl, N);
        finish_out(c, ru, dru, N);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_iterate_complementarity(ipxk_context* c, double out4[4]) {
    return guarded([&] {
        IPXK_REQUIRE(c && out4, "NULL argument");
        bind_device(c);
        iterate_complementarity_dev(c, out4);
    });
}

int ipxk_step_to_boundary(ipxk_context* c, const double* x, const double* dx, ipxint len, double alpha0,
                          double* alpha, ipxint* blocking_index) {
    return guarded([&] {
        IPXK_REQUIRE(c && x && dx && alpha && len >= 0, "bad argument");
        bind_device(c);
        const double* dxv = stage_in(c, x, (size_t)len, c->nw_in[0]);
        const double* ddx = stage_in(c, dx, (size_t)len, c->nw_in[1]);
        *alpha = step_to_boundary_dev(c, dxv, ddx, len, alpha0, blocking_index);
    });
}

int ipxk_ipm_step(ipxk_context* c, int use_basis, const double* b, const double* cc, const double* lb,
                  const double* ub, double kkt_tol, ipxint maxiter, ipxk_ipm_step_info* info,
                  ipxk_interrupt_fn interrupt, void* interrupt_user) {
    return guarded([&] {
        IPXK_REQUIRE(c && b && cc && lb && ub && info, "NULL argument");
        IPXK_REQUIRE(use_basis ? c->split != nullptr : c->kkt_diag_factorized, "KKT solver not factorized");
        IPXK_REQUIRE(!comm_active(c), "ipxk_ipm_step is not available on a partitioned system");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* db = stage_in(c, b, m, c->nw_in[0]);
        const double* dc = stage_in(c, cc, N, c->nw_in[1]);
        const double* dlb = stage_in(c, lb, N, c->nw_in[2]);
        const double* dub = stage_in(c, ub, N, c->nw_in[3]);
        ipm_step_dev(c, use_basis != 0, db, dc, dlb, dub, kkt_tol, maxiter, info, interrupt, interrupt_user);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_iterate_objectives(ipxk_context* c, const double* b, const double* cc, const double* lb, const double* ub,
                            double out3[3]) {
    return guarded([&] {
        IPXK_REQUIRE(c && b && cc && lb && ub && out3, "NULL argument");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* db = stage_in(c, b, m, c->nw_in[0]);
        const double* dc = stage_in(c, cc, N, c->nw_in[1]);
        const double* dlb = stage_in(c, lb, N, c->nw_in[2]);
        const double* dub = stage_in(c, ub, N, c->nw_in[3]);
        iterate_objectives_dev(c, db, dc, dlb, dub, out3);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_ipm_driver(ipxk_context* c, const double* b, const double* cc, const double* lb, const double* ub,
                    const ipxk_ipm_params* params, ipxk_ipm_info* info, ipxk_interrupt_fn interrupt, void* interrupt_user) {
    return guarded([&] {
        IPXK_REQUIRE(c && b && cc && lb && ub && params && info, "NULL argument");
        IPXK_REQUIRE(!comm_active(c), "ipxk_ipm_driver is not available on a partitioned system");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* db = stage_in(c, b, m, c->nw_in[0]);
        const double* dc = stage_in(c, cc, N, c->nw_in[1]);
        const double* dlb = stage_in(c, lb, N, c->nw_in[2]);
        const double* dub = stage_in(c, ub, N, c->nw_in[3]);
        ipm_driver_dev(c, db, dc, dlb, dub, params, info, interrupt, interrupt_user);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_ipm_driver_basis(ipxk_context* c, const double* b, const double* cc, const double* lb, const double* ub,
                          const ipxk_ipm_params* params, ipxk_ipm_info* info, ipxint* basis_out, ipxint* status_out,
                          ipxk_interrupt_fn interrupt, void* interrupt_user) {
    return guarded([&] {
        IPXK_REQUIRE(c && b && cc && lb && ub && params && info, "NULL argument");
        IPXK_REQUIRE(!comm_active(c), "ipxk_ipm_driver_basis is not available on a partitioned system");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* db = stage_in(c, b, m, c->nw_in[0]);
        const double* dc = stage_in(c, cc, N, c->nw_in[1]);
        const double* dlb = stage_in(c, lb, N, c->nw_in[2]);
        const double* dub = stage_in(c, ub, N, c->nw_in[3]);
        ipm_driver_dev(c, db, dc, dlb, dub, params, info, interrupt, interrupt_user, true, basis_out, status_out);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_iterate_factorize_diag(ipxk_context* c, int precond_dense_cols, ipxint* errflag) {
    return guarded([&] {
        IPXK_REQUIRE(c && errflag && c->it_set, "no iterate on the device (ipxk_iterate_set)");
        bind_device(c);
        double comp[4];
        iterate_complementarity_dev(c, comp);
        kkt_diag_factorize_dev(c, c->it_xl.get(), c->it_xu.get(), c->it_zl.get(), c->it_zu.get(), comp[1],
                               precond_dense_cols != 0, errflag);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

// ---- SplittedNormalMatrix / basis path -------------------------------------------------
int ipxk_split_prepare(ipxk_context* c, const ipxint* Lp, const ipxint* Li, const double* Lx,
                       const ipxint* Up, const ipxint* Ui, const double* Ux, const ipxint* rowperm,
                       const ipxint* colperm, const ipxint* basis, const ipxint* status,
                       const double* colscale) {
    return guarded([&] {
        IPXK_REQUIRE(c && Lp && Up && rowperm && colperm && basis && status && colscale, "NULL argument");
        bind_device(c);
        split_prepare_host(c, Lp, Li, Lx, Up, Ui, Ux, rowperm, colperm, basis, status, colscale);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_equilibrate(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai, double* Ax, double* colscale,
                     double* rowscale, ipxint* rounds, int device) {
    return guarded([&] {
        IPXK_REQUIRE(Ap && (Ap[n] == 0 || (Ai && Ax)) && colscale && rowscale && rounds, "NULL argument");
        equilibrate_device(device, m, n, Ap, Ai, Ax, colscale, rowscale, rounds);
    });
}

int ipxk_transpose(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai, const double* Ax, ipxint* ATp, ipxint* ATi,
                   double* ATx, int device) {
    return guarded([&] {
        IPXK_REQUIRE(Ap && ATp && (Ap[n] == 0 || (Ai && Ax && ATi && ATx)), "NULL argument");
        transpose_device(device, m, n, Ap, Ai, Ax, ATp, ATi, ATx);
    });
}

int ipxk_lu_factorize(ipxk_context* c, ipxint dim, const ipxint* Bbegin, const ipxint* Bend, const ipxint* Bi,
                      const double* Bx, double pivottol, int strict_abs_pivottol, ipxk_lu_info* info) {
    return guarded([&] {
        IPXK_REQUIRE(c && dim >= 0 && (dim == 0 || (Bbegin && Bend)), "NULL argument");
        bind_device(c);
        lu_factorize_host(c, dim, Bbegin, Bend, Bi, Bx, pivottol, strict_abs_pivottol != 0, info);
    });
}

int ipxk_lu_factorize_basis(ipxk_context* c, const ipxint* basis, double pivottol, int strict_abs_pivottol,
                            ipxk_lu_info* info) {
    return guarded([&] {
        IPXK_REQUIRE(c && (basis || c->m == 0), "NULL argument");
        bind_device(c);
        lu_factorize_basis(c, basis, pivottol, strict_abs_pivottol != 0, info);
    });
}

ipxint ipxk_lu_generation(const ipxk_context* c) { return c ? (ipxint)lu_generation(c) : 0; }

int ipxk_lu_get_factors(ipxk_context* c, ipxint* Lp, ipxint* Li, double* Lx, ipxint* Up, ipxint* Ui, double* Ux,
                        ipxint* rowperm, ipxint* colperm, ipxint* dependent_cols) {
    return guarded([&] {
        IPXK_REQUIRE(c, "NULL argument");
        bind_device(c);
        lu_get_factors(c, Lp, Li, Lx, Up, Ui, Ux, rowperm, colperm, dependent_cols);
    });
}

int ipxk_split_prepare_lu(ipxk_context* c, const ipxint* status, const double* colscale) {
    return guarded([&] {
        IPXK_REQUIRE(c && status && colscale, "NULL argument");
        bind_device(c);
        split_prepare_lu(c, status, colscale);
    });
}

int ipxk_maxvolume(ipxk_context* c, const ipxint* status, const double* colscale, const ipxk_maxvolume_params* params,
                   ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info, ipxint* exchange_log, ipxint log_cap) {
    return guarded([&] {
        IPXK_REQUIRE(c && status && colscale, "NULL argument");
        IPXK_REQUIRE(log_cap >= 0 && (exchange_log || log_cap == 0), "bad log arguments");
        bind_device(c);
        maxvolume_dev(c, status, colscale, params, basis_out, status_out, info, exchange_log, log_cap);
    });
}

int ipxk_maxvolume_sequential(ipxk_context* c, const ipxint* status, const double* colscale, double volume_tol, ipxint maxpasses,
                              ipxint max_etas, ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info,
                              ipxint* exchange_log, ipxint log_cap) {
    return guarded([&] {
        IPXK_REQUIRE(c && status && colscale, "NULL argument");
        IPXK_REQUIRE(log_cap >= 0 && (exchange_log || log_cap == 0), "bad log arguments");
        bind_device(c);
        maxvolume_sequential_dev(c, status, colscale, volume_tol, maxpasses, max_etas, basis_out, status_out, info, exchange_log, log_cap);
    });
}

int ipxk_cr_diagnostics(ipxk_context* c, ipxk_cr_diag* out) {
    return guarded([&] {
        IPXK_REQUIRE(c && out, "NULL argument");
        bind_device(c);
        cr_diagnostics_dev(c, out);
    });
}

int ipxk_split_rescale(ipxk_context* c, const ipxint* status, const double* colscale) {
    return guarded([&] {
        IPXK_REQUIRE(c && status && colscale, "NULL argument");
        IPXK_REQUIRE(c->split != nullptr, "SplittedNormalMatrix not prepared");
        bind_device(c);
        split_rescale_host(c, status, colscale);
    });
}

int ipxk_split_apply(ipxk_context* c, const double* rhs, double* lhs, double* rhs_dot_lhs) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs, "NULL argument");
        IPXK_REQUIRE(c->split != nullptr, "SplittedNormalMatrix not prepared");
        bind_device(c);
        if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
        const size_t m = (size_t)c->m;
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        const int np = split_apply_dev(c, drhs, dlhs, nullptr);
        IPXK_HIP(hipGetLastError());
        if (rhs_dot_lhs) *rhs_dot_lhs = reduce_partials_host(c, kPartCdot, np, false);
        finish_out(c, lhs, dlhs, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        check_sweep_abort(c);
    });
}

static int inplace_solve(ipxk_context* c, double* x, bool forward) {
    return guarded([&] {
        IPXK_REQUIRE(c && x, "NULL argument");
        IPXK_REQUIRE(c->split != nullptr, "SplittedNormalMatrix not prepared");
        bind_device(c);
        const size_t m = (size_t)c->m;
        double* dx = stage_out(c, x, m, c->v_lhs);
        if (c->pointer_mode == IPXK_POINTER_HOST) staged_h2d(dx, x, m * sizeof(double), c->stream);
        if (forward) forward_solve_dev(c, dx, dx, true, nullptr);
        else backward_solve_dev(c, dx, dx, true, nullptr);
        IPXK_HIP(hipGetLastError());
        finish_out(c, x, dx, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        check_sweep_abort(c);
    });
}

int ipxk_forward_solve(ipxk_context* c, double* x) { return inplace_solve(c, x, true); }
int ipxk_backward_solve(ipxk_context* c, double* x) { return inplace_solve(c, x, false); }

int ipxk_solve_dense(ipxk_context* c, const double* rhs, double* lhs, char trans) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs && lhs, "NULL argument");
        IPXK_REQUIRE(c->split != nullptr, "SplittedNormalMatrix not prepared");
        bind_device(c);
        const size_t m = (size_t)c->m;
        const double* drhs = stage_in(c, rhs, m, c->v_rhs);
        double* dlhs = stage_out(c, lhs, m, c->v_lhs);
        solve_dense_dev(c, drhs, dlhs, trans);
        IPXK_HIP(hipGetLastError());
        finish_out(c, lhs, dlhs, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        check_sweep_abort(c);
    });
}

int ipxk_split_levels(const ipxk_context* c, ipxint levels[4]) {
    return guarded([&] {
        IPXK_REQUIRE(c && c->split && levels, "SplittedNormalMatrix not prepared");
        split_levels(c, levels);
    });
}

int ipxk_kkt_basis_solve(ipxk_context* c, const double* a, const double* b, double tol, ipxint maxiter,
                         double* x, double* y, ipxint* iter, ipxint* errflag,
                         ipxk_interrupt_fn interrupt, void* interrupt_user, ipxk_times* times) {
    return guarded([&] {
        IPXK_REQUIRE(c && a && b && x && y && iter && errflag, "NULL argument");
        IPXK_REQUIRE(c->split != nullptr, "KKTSolverBasis not factorized (split operator missing)");
        bind_device(c);
        const size_t m = (size_t)c->m, N = (size_t)(c->n + c->m);
        const double* da = stage_in(c, a, N, c->k_a);
        const double* db = stage_in(c, b, m, c->k_b);
        double* dx = stage_out(c, x, N, c->k_x);
        double* dy = stage_out(c, y, m, c->k_y);
        if (times) *times = ipxk_times{};
        CrResult r = kkt_basis_solve_dev(c, da, db, tol, maxiter, dx, dy, interrupt, interrupt_user, times);
        *iter = r.iter;
        *errflag = r.errflag;
        finish_out(c, x, dx, N);
        finish_out(c, y, dy, m);
        IPXK_HIP(hipStreamSynchronize(c->stream));
        check_sweep_abort(c);
    });
}

// ---- measurement ---------------------------------------------------------------------
int ipxk_time_normal_apply(ipxk_context* c, const double* rhs_dev, double* lhs_dev, int reps,
                           double* ms_total) {
    return guarded([&] {
        IPXK_REQUIRE(c && rhs_dev && lhs_dev && ms_total && reps > 0, "bad argument");
        IPXK_REQUIRE(c->normal_prepared, "NormalMatrix not prepared");
        bind_device(c);
        if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
        hipEvent_t e0, e1;
        IPXK_HIP(hipEventCreate(&e0));
        IPXK_HIP(hipEventCreate(&e1));
        int np = 0;
        IPXK_HIP(hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps; r++) normal_apply_dev(c, c->W, rhs_dev, lhs_dev, &np, nullptr);
        IPXK_HIP(hipEventRecord(e1, c->stream));
        IPXK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_total = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        IPXK_HIP(hipGetLastError());
    });
}

// tuning aid (not part of the reference-facing surface): times one pass of the product
int ipxk_debug_time_pass(ipxk_context* c, int which, const double* x_dev, double* out_dev, int reps,
                         double* ms_total) {
    return guarded([&] {
        bind_device(c);
        hipEvent_t e0, e1;
        IPXK_HIP(hipEventCreate(&e0));
        IPXK_HIP(hipEventCreate(&e1));
        IPXK_HIP(hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps; r++) debug_single_pass(c, which, x_dev, out_dev);
        IPXK_HIP(hipEventRecord(e1, c->stream));
        IPXK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_total = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    });
}

int ipxk_debug_get_stamps(ipxk_context* c, int which, unsigned long long* out, ipxint cap, int* geom) {
    return guarded([&] {
        bind_device(c);
        const GatherMatrix& M = which == 1 ? c->Acols : c->Arows;
        geom[0] = M.P; geom[1] = M.G; geom[2] = M.RT; geom[3] = M.Q;
        const size_t n = std::min<size_t>((size_t)cap, M.stamps.size());
        M.stamps.download(out, n, c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}

int ipxk_split_inverse_stats(const ipxk_context* c, ipxint* probes, ipxint* rejected, double* worst_residual) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        if (probes) *probes = c->split_stats.inverse_probes;
        if (rejected) *rejected = c->split_stats.inverse_rejected;
        if (worst_residual) *worst_residual = c->split_stats.worst_probe;
    });
}

ipxint ipxk_split_inverse_refined(const ipxk_context* c) { return c ? (ipxint)c->split_stats.inverse_refined : 0; }

// Layout inspection (tests/test_gpu_layout.py: the device builders against the host builders, array by array).
int ipxk_layout_info(const ipxk_context* c, int which, ipxint info[40], double create_ms[4]) {
    return guarded([&] {
        IPXK_REQUIRE(c && info && (which == 0 || which == 1), "bad argument");
        const GatherMatrix& M = which == 0 ? c->Acols : c->Arows;
        for (int i = 0; i < 40; i++) info[i] = 0;
        info[0] = M.use_sliced; info[1] = M.use_sorted; info[2] = M.use_sorted_fused; info[3] = M.nlong;
        info[4] = M.sliced.built; info[5] = M.sliced.R; info[6] = M.sliced.nslices; info[7] = M.sliced.nrb;
        info[8] = M.sliced.nrows_pad; info[9] = M.sliced.max_tile;
        { double d = M.sliced.dominant_fraction; memcpy(&info[10], &d, sizeof d); }
        info[11] = M.sorted.built; info[12] = M.sorted.nslices; info[13] = M.sorted.nsub; info[14] = M.sorted.nrb;
        info[15] = M.sorted.RB; info[16] = M.sorted.nrows_pad; info[17] = M.sorted.max_sub; info[18] = M.sorted.slice_elems;
        info[19] = M.sorted.fused; info[20] = M.nnz;
        info[21] = M.P; info[22] = M.G; info[23] = (ipxint)M.RT * 1000000 + (ipxint)M.Q * 1000 + 0;
        info[24] = M.use_acc; info[25] = M.acc.built; info[26] = M.acc.nslices; info[27] = M.acc.nrb; info[28] = M.acc.RB;
        info[29] = M.acc.nrows_pad; info[30] = M.acc.slice_elems; info[31] = M.acc.nbatches; info[32] = M.acc.deferred;
        info[33] = M.use_acc_fused; info[34] = M.accf.built; info[35] = M.accf.nrb; info[36] = M.accf.RB; info[37] = M.accf.nbatches;
        info[38] = M.use_plain;
        if (create_ms) for (int i = 0; i < 4; i++) create_ms[i] = c->create_ms[i];
    });
}
// array: 0 sliced.tile_ptr (u32) 1 sliced.cnt (u8) 2 sliced.idx (i32) 3 sliced.val (f64) 4 sorted.sub_ptr (u32)
// 5 sorted.cnt (u8) 6 sorted.pack (u32) 7 sorted.val (f64) 8 plain row-wise ptr (i32) 9 idx (i32) 10 val (f64).
// Copies min(cap, size) bytes to `out` (host), *nbytes = size of the array in bytes.
int ipxk_layout_array(ipxk_context* c, int which, int array, void* out, ipxint cap, ipxint* nbytes) {
    return guarded([&] {
        IPXK_REQUIRE(c && nbytes && (which == 0 || which == 1), "bad argument");
        bind_device(c);
        const GatherMatrix& M = which == 0 ? c->Acols : c->Arows;
        const void* src = nullptr;
        size_t bytes = 0;
        const size_t ntiles = (size_t)M.sliced.nrb * M.sliced.nslices, nsubs = (size_t)M.sorted.nrb * std::max(M.sorted.nslices, 1) * std::max(M.sorted.nsub, 1);
        const size_t nz = (size_t)M.nnz;
        switch (array) {
            case 0: src = M.sliced.tile_ptr.get(); bytes = M.sliced.built ? (ntiles + 1) * 4 : 0; break;
            case 1: src = M.sliced.cnt.get(); bytes = M.sliced.built ? ntiles * M.sliced.R : 0; break;
            case 2: src = M.sliced.idx.get(); bytes = M.sliced.built ? nz * 4 : 0; break;
            case 3: src = M.sliced.val.get(); bytes = M.sliced.built ? nz * 8 : 0; break;
            case 4: src = M.sorted.sub_ptr.get(); bytes = M.sorted.built ? (nsubs + 1) * 4 : 0; break;
            case 5: src = M.sorted.cnt.get(); bytes = M.sorted.built ? nsubs * M.sorted.RB : 0; break;
            case 6: src = M.sorted.pack.get(); bytes = M.sorted.built ? nz * 4 : 0; break;
            case 7: src = M.sorted.val.get(); bytes = M.sorted.built ? nz * 8 : 0; break;
            case 8: src = c->pl_Tp.get(); bytes = ((size_t)c->m + 1) * 4; break;
            case 9: src = c->pl_Ti.get(); bytes = (size_t)c->nnz * 4; break;
            case 10: src = c->pl_Tx.get(); bytes = (size_t)c->nnz * 8; break;
            case 11: src = M.acc.tile_batch.get(); bytes = M.acc.built ? ((size_t)M.acc.nrb * M.acc.nslices + 1) * 4 : 0; break;
            case 12: src = M.acc.bptr.get(); bytes = M.acc.built ? ((size_t)M.acc.nbatches + 1) * 4 : 0; break;
            case 13: src = M.acc.pack.get(); bytes = M.acc.built ? nz * 4 : 0; break;
            case 14: src = M.acc.val.get(); bytes = M.acc.built ? nz * 8 : 0; break;
            case 15: src = M.accf.tile_batch.get(); bytes = M.accf.built ? ((size_t)M.accf.nrb + 1) * 4 : 0; break;
            case 16: src = M.accf.bptr.get(); bytes = M.accf.built ? ((size_t)M.accf.nbatches + 1) * 4 : 0; break;
            case 17: src = M.accf.pack.get(); bytes = M.accf.built ? nz * 4 : 0; break;
            case 18: src = M.accf.val.get(); bytes = M.accf.built ? nz * 8 : 0; break;
            case 19: src = M.accf.xmin.get(); bytes = M.accf.built ? (size_t)M.accf.nrb * 4 : 0; break;
            case 20: src = M.sorted.xmin.get(); bytes = M.sorted.built && M.sorted.fused ? (size_t)M.sorted.nrb * 4 : 0; break;
            default: IPXK_REQUIRE(false, "unknown array");
        }
        if (M.nlong > 0 && (array < 8 || array > 10)) bytes = 0;       // long rows: the tiles hold fewer entries than nnz; not inspected
        *nbytes = (ipxint)bytes;
        const size_t ncopy = std::min<size_t>(bytes, cap > 0 ? (size_t)cap : 0);
        if (out && ncopy > 0) staged_d2h(out, src, ncopy, c->stream);
    });
}

ipxint ipxk_normal_apply_bytes(const ipxk_context* c) {
    if (!c) return 0;
    const ipxint wi = 4;
    return 2 * c->nnz * (wi + 8) + (c->n + c->m + 2) * wi + 8 * (3 * c->n + 4 * c->m);
}

int ipxk_spmv_layout(const ipxk_context* c, int layout[2], double us[6]) {
    return guarded([&] {
        IPXK_REQUIRE(c && layout, "bad argument");
        auto code = [](const GatherMatrix& M) { return M.use_acc_fused ? 7 : M.use_plain ? 6 : M.use_acc ? 5 : M.use_sorted_fused ? 4 : !M.use_sliced ? 0 : M.sliced.nslices == 1 ? 2 : M.use_sorted ? 3 : 1; };
        layout[0] = code(c->Acols);
        layout[1] = code(c->Arows);
        if (us) {
            us[0] = c->Acols.tuned_us_phased; us[1] = c->Acols.use_acc ? c->Acols.tuned_us_acc : c->Acols.use_sorted ? c->Acols.tuned_us_sorted : c->Acols.tuned_us_sliced;
            us[2] = c->Acols.use_acc_fused ? c->Acols.tuned_us_acc_fused : c->Acols.use_plain ? c->Acols.tuned_us_plain : c->Acols.use_sorted_fused ? c->Acols.tuned_us_sorted_fused : c->Acols.tuned_us_fused;
            us[3] = c->Arows.tuned_us_phased; us[4] = c->Arows.use_acc ? c->Arows.tuned_us_acc : c->Arows.use_sorted ? c->Arows.tuned_us_sorted : c->Arows.tuned_us_sliced;
            us[5] = c->Arows.use_acc_fused ? c->Arows.tuned_us_acc_fused : c->Arows.use_plain ? c->Arows.tuned_us_plain : c->Arows.use_sorted_fused ? c->Arows.tuned_us_sorted_fused : c->Arows.tuned_us_fused;
        }
    });
}

int ipxk_dev_alloc(ipxk_context* c, ipxint bytes, void** ptr) {
    return guarded([&] {
        IPXK_REQUIRE(c && ptr && bytes >= 0, "bad argument");
        bind_device(c);
        *ptr = nullptr;
        IPXK_HIP(hipMalloc(ptr, bytes > 0 ? (size_t)bytes : 8));
    });
}
int ipxk_dev_free(ipxk_context* c, void* ptr) {
    return guarded([&] {
        IPXK_REQUIRE(c != nullptr, "ctx is NULL");
        bind_device(c);
        if (ptr) IPXK_HIP(hipFree(ptr));
    });
}
int ipxk_dev_upload(ipxk_context* c, void* dst_dev, const void* src_host, ipxint bytes) {
    return guarded([&] {
        IPXK_REQUIRE(c && dst_dev && src_host && bytes >= 0, "bad argument");
        bind_device(c);
        staged_h2d(dst_dev, src_host, (size_t)bytes, c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    });
}
int ipxk_dev_download(ipxk_context* c, void* dst_host, const void* src_dev, ipxint bytes) {
    return guarded([&] {
        IPXK_REQUIRE(c && dst_host && src_dev && bytes >= 0, "bad argument");
        bind_device(c);
        staged_d2h(dst_host, src_dev, (size_t)bytes, c->stream);
    });
}

}  // extern "C"
