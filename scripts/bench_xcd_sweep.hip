// Microbenchmark for the level-scheduled triangular sweeps (prototype of trisolve.hip's kernels): cost of a
// whole sweep over a given level-width profile when levels are executed as
//   L  one launch per level (packed ELL records: record + entries in ONE round trip, gathers in a second)
//   S  sync-free runs: ONE launch for a run of levels, value-as-flag hand-off (the result vector is pre-filled
//      with a sentinel; every unknown is stored once with a write-through (sc1) store; consumers poll with
//      L1-bypassing loads), every wave owns chunks of 64 level-ordered positions in a fixed round-robin
//   X  the same confined to ONE XCD (workgroups with blockIdx % 8 == 0 under round-robin dispatch) with plain
//      stores: the hand-off goes through that XCD's L2
// A plan assigns each level by its width: width <= T1 -> X, width > T2 -> L, else S; consecutive levels of one
// kind form one launch.  Synthetic DAG: 3 dependencies per unknown (2 in the previous level, 1 anywhere
// earlier), unknowns scattered over the vector.  Every result is compared bit for bit with a sequential host
// solve; every spin is bounded (abort flag).
//   build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o bench_xcd_sweep bench_xcd_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr unsigned long long kSentinel = 0x7FF8DEAD5EEDBEEFull;
constexpr int W = 4;            // ELL width of the synthetic chunks (3 entries used)
constexpr int kSpinLimit = 1 << 18;

struct Packed {
    const int* order;           // [npos] unknown of position (-1: padding)
    const double* diag;         // [npos]
    const unsigned char* len;   // [npos]
    const int* idx;             // [nchunks][W][64]
    const double* val;
};

__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// L: one level per launch
__global__ __launch_bounds__(256) void level_kernel(Packed S, int pos0, int npos, const double* xin, double* xout,
                                                    const int* done) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= npos) return;
    const int pos = pos0 + t, lane = threadIdx.x & 63;
    const int chunk = pos >> 6;
    const int dn = done ? *done : 0;      // not on the critical path: only predicates the store
    const int r = S.order[pos];
    const double dg = S.diag[pos];
    const int len = S.len[pos];
    int j[W]; double a[W];
#pragma unroll
    for (int e = 0; e < W; e++) {
        j[e] = S.idx[(size_t)chunk * (W * 64) + e * 64 + lane];
        a[e] = S.val[(size_t)chunk * (W * 64) + e * 64 + lane];
    }
    if (r < 0) return;
    const double xr = xin[r];
    double v[W];
#pragma unroll
    for (int e = 0; e < W; e++) v[e] = e < len ? xout[j[e]] : 0.0;
    double acc = xr;
#pragma unroll
    for (int e = 0; e < W; e++) if (e < len) acc -= a[e] * v[e];
    if (!dn) xout[r] = acc / dg;
}

struct Rec {
    int r, len; double dg, xr; int j[W]; double a[W];
};
__device__ __forceinline__ void load_rec(Rec& R, const Packed& S, int c, int lane, const double* xin) {
    const int pos = c * 64 + lane;
    R.r = S.order[pos];
    R.dg = S.diag[pos];
    R.len = S.len[pos];
#pragma unroll
    for (int e = 0; e < W; e++) {
        R.j[e] = S.idx[(size_t)c * (W * 64) + e * 64 + lane];
        R.a[e] = S.val[(size_t)c * (W * 64) + e * 64 + lane];
    }
    R.xr = R.r >= 0 ? xin[R.r] : 0.0;
}

// S / X: a run of levels in one launch
template <bool SC1_STORE, int SLEEP>
__global__ __launch_bounds__(256) void run_kernel(Packed S, int chunk0, int nchunks, const double* xin, double* xout,
                                                  int xcd_only, int* abort_flag, unsigned* xcc_seen) {
    if (xcd_only && (blockIdx.x & 7) != 0) return;
    const int part = xcd_only ? blockIdx.x >> 3 : blockIdx.x;
    const int nparts = xcd_only ? (gridDim.x + 7) >> 3 : gridDim.x;
    const int lane = threadIdx.x & 63;
    const int gw = part * 4 + (threadIdx.x >> 6), NW = nparts * 4;
    if (lane == 0 && xcc_seen) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_seen[gw] = id & 0xf;
    }
    const unsigned long long* xo = reinterpret_cast<const unsigned long long*>(xout);
    int c = chunk0 + gw;
    const int cend = chunk0 + nchunks;
    if (c >= cend) return;
    Rec A;
    load_rec(A, S, c, lane, xin);
    for (;;) {
        unsigned long long bits[W];
#pragma unroll
        for (int e = 0; e < W; e++) bits[e] = (A.r >= 0 && e < A.len) ? ld_sc1(xo + A.j[e]) : 0ull;
        const int cn = c + NW;
        Rec B;
        if (cn < cend) load_rec(B, S, cn, lane, xin);     // in flight while this chunk waits for its dependencies
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int e = 0; e < W; e++) ok &= bits[e] != kSentinel;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
            for (int e = 0; e < W; e++) if (bits[e] == kSentinel) bits[e] = ld_sc1(xo + A.j[e]);
            if (++spins > kSpinLimit || ((spins & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        double acc = A.xr;
#pragma unroll
        for (int e = 0; e < W; e++) if (e < A.len) acc -= A.a[e] * __longlong_as_double((long long)bits[e]);
        if (A.r >= 0) {
            const unsigned long long out = (unsigned long long)__double_as_longlong(acc / A.dg);
            if (SC1_STORE) __hip_atomic_store(reinterpret_cast<unsigned long long*>(xout) + A.r, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(reinterpret_cast<unsigned long long*>(xout) + A.r, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (cn >= cend) break;
        A = B;
        c = cn;
    }
}

__global__ void fill_kernel(int n, unsigned long long* x) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = kSentinel;
}

static std::vector<int> profile(const std::string& name) {
    if (name == "ut") return {1, 1, 1, 1, 3, 4, 6, 7, 9, 10, 8, 11, 6, 8, 9, 14, 16, 17, 29, 37, 39, 56, 76, 94, 98, 120, 134, 151, 157, 173, 230, 264, 307, 379, 413, 448, 507, 596, 676, 724, 814, 928, 1125, 1290, 1571, 1788, 2101, 2453, 2867, 3220, 3869, 4570, 5309, 6205, 7321, 8621, 10003, 11578, 13459, 15707, 18100, 20932, 24087, 27987, 32441, 37164, 42632, 48282, 53737, 58865, 63545, 66557, 67938, 66744, 62207, 54663, 45698, 35387, 25222, 16509, 10163, 5342, 2754, 1365, 616, 259, 111, 36, 14, 4};
    if (name == "lf") return {250698, 148516, 104419, 79805, 63445, 51768, 42705, 35636, 30367, 25986, 22123, 18815, 16221, 14117, 12182, 10671, 9191, 7905, 6984, 6046, 5226, 4482, 3920, 3434, 3024, 2697, 2354, 2100, 1833, 1588, 1387, 1219, 1052, 893, 821, 732, 687, 593, 523, 467, 383, 351, 318, 272, 244, 207, 189, 176, 169, 129, 124, 116, 104, 79, 67, 57, 59, 44, 29, 27, 17, 23, 23, 17, 18, 13, 16, 17, 10, 10, 14, 10, 7, 2, 2, 5, 3, 3, 4, 3, 1, 2, 1, 1, 1, 1};
    // const:WIDTH:LEVELS
    int w = 1024, l = 64;
    sscanf(name.c_str(), "const:%d:%d", &w, &l);
    return std::vector<int>(l, w);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
    int* abort_flag; CHECK(hipMalloc(&abort_flag, 4));
    unsigned* xcc_seen; CHECK(hipMalloc(&xcc_seen, 65536 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    std::vector<std::string> names;
    for (int i = 1; i < argc; i++) names.push_back(argv[i]);
    if (names.empty()) names = {"const:64:64", "const:1024:64", "const:4096:64", "const:16384:32", "const:65536:16", "const:262144:6", "ut", "lf"};
    for (const std::string& name : names) {
        const std::vector<int> widths = profile(name);
        const int levels = (int)widths.size();
        std::vector<int> lptr(levels + 1, 0);
        for (int l = 0; l < levels; l++) lptr[l + 1] = lptr[l] + (widths[l] + 63) / 64 * 64;
        const int npos = lptr[levels], nchunks = npos / 64;
        int M = 1 << 20; while (M < npos) M <<= 1;
        std::mt19937_64 rng(12345);
        std::vector<int> perm(M);
        for (int i = 0; i < M; i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<int> order(npos, -1), idx((size_t)nchunks * W * 64, 0);
        std::vector<double> diag(npos, 1.0), val((size_t)nchunks * W * 64, 0.0), xin(M), ref(M, 0.0);
        std::vector<unsigned char> len(npos, 0);
        std::vector<int> real;      // real (non-padding) positions in order
        std::uniform_real_distribution<double> U(0.1, 0.6), D(0.5, 2.0), R(-1.0, 1.0);
        for (int i = 0; i < M; i++) xin[i] = R(rng);
        int nreal = 0;
        std::vector<int> real_before(levels + 1, 0);
        for (int l = 0; l < levels; l++) {
            for (int t = 0; t < widths[l]; t++) {
                const int p = lptr[l] + t;
                order[p] = perm[nreal + t];
                diag[p] = D(rng);
                len[p] = l == 0 ? 0 : 3;
                const int c = p >> 6, lane = p & 63;
                for (int e = 0; e < (int)len[p]; e++) {
                    int q;    // index into perm of the dependency
                    if (e < 2) q = real_before[l - 1] + (int)(rng() % widths[l - 1]);
                    else q = (int)(rng() % (size_t)real_before[l]);
                    idx[(size_t)c * W * 64 + e * 64 + lane] = perm[q];
                    val[(size_t)c * W * 64 + e * 64 + lane] = U(rng);
                }
            }
            nreal += widths[l];
            real_before[l + 1] = nreal;
        }
        for (int p = 0; p < npos; p++) {
            if (order[p] < 0) continue;
            const int c = p >> 6, lane = p & 63;
            double acc = xin[order[p]];
            for (int e = 0; e < (int)len[p]; e++)
                acc -= val[(size_t)c * W * 64 + e * 64 + lane] * ref[idx[(size_t)c * W * 64 + e * 64 + lane]];
            ref[order[p]] = acc / diag[p];
        }
        int *d_order, *d_idx; double *d_diag, *d_val, *d_xin, *d_xout; unsigned char* d_len;
        CHECK(hipMalloc(&d_order, npos * 4)); CHECK(hipMalloc(&d_idx, idx.size() * 4)); CHECK(hipMalloc(&d_diag, (size_t)npos * 8));
        CHECK(hipMalloc(&d_val, val.size() * 8)); CHECK(hipMalloc(&d_xin, (size_t)M * 8)); CHECK(hipMalloc(&d_xout, (size_t)M * 8));
        CHECK(hipMalloc(&d_len, npos));
        CHECK(hipMemcpy(d_order, order.data(), npos * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_diag, diag.data(), (size_t)npos * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_xin, xin.data(), (size_t)M * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_len, len.data(), npos, hipMemcpyHostToDevice));
        Packed S{d_order, d_diag, d_len, d_idx, d_val};
        std::vector<double> got(M);
        auto check = [&](const char* what) {
            CHECK(hipMemcpy(got.data(), d_xout, (size_t)M * 8, hipMemcpyDeviceToHost));
            int ab; CHECK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int p = 0; p < npos; p++) if (order[p] >= 0) bad += memcmp(&got[order[p]], &ref[order[p]], 8) != 0;
            if (ab || bad) printf("    %s: %s, %zu of %d values differ\n", what, ab ? "ABORTED" : "completed", bad, nreal);
            return !ab && !bad;
        };
        printf("== profile %s: %d levels, %d unknowns, %d chunks\n", name.c_str(), levels, nreal, nchunks);
        struct Plan { int T1, T2, wgsX, gridS, sleep; };
        std::vector<Plan> plans = {
            {0, 0, 0, 0, 1},                    // all launches
            {0, 1 << 30, 0, 256, 1}, {0, 1 << 30, 0, 512, 1}, {0, 1 << 30, 0, 1024, 1}, {0, 1 << 30, 0, 512, 4},
            {0, 32768, 0, 512, 1}, {0, 100000, 0, 512, 1},
            {512, 1 << 30, 16, 512, 1}, {2048, 1 << 30, 32, 512, 1}, {2048, 100000, 32, 512, 1}, {4096, 100000, 32, 512, 1},
            {1 << 30, 1 << 30, 32, 0, 1},       // everything on one XCD
        };
        for (const Plan& P : plans) {
            float best = 1e30f; bool ok = true; int nlaunch = 0, nL = 0, nS = 0, nX = 0;
            for (int rep = 0; rep < 4 && ok; rep++) {
                CHECK(hipMemsetAsync(abort_flag, 0, 4, s));
                fill_kernel<<<256, 256, 0, s>>>(M, (unsigned long long*)d_xout);
                CHECK(hipEventRecord(e0, s));
                nlaunch = nL = nS = nX = 0;
                auto kind = [&](int l) { return widths[l] <= P.T1 ? 2 : widths[l] > P.T2 ? 0 : 1; };
                for (int l = 0; l < levels;) {
                    const int k = kind(l);
                    int b = l + 1;
                    if (k != 0) while (b < levels && kind(b) == k) b++;
                    const int c0 = lptr[l] / 64, nc = (lptr[b] - lptr[l]) / 64;
                    if (k == 0) {
                        level_kernel<<<(lptr[b] - lptr[l] + 255) / 256, 256, 0, s>>>(S, lptr[l], lptr[b] - lptr[l], d_xin, d_xout, abort_flag);
                        nL++;
                    } else if (k == 1) {
                        if (P.sleep == 1) run_kernel<true, 1><<<P.gridS, 256, 0, s>>>(S, c0, nc, d_xin, d_xout, 0, abort_flag, xcc_seen);
                        else run_kernel<true, 4><<<P.gridS, 256, 0, s>>>(S, c0, nc, d_xin, d_xout, 0, abort_flag, xcc_seen);
                        nS++;
                    } else {
                        run_kernel<false, 1><<<P.wgsX * 8, 256, 0, s>>>(S, c0, nc, d_xin, d_xout, 1, abort_flag, xcc_seen);
                        nX++;
                    }
                    nlaunch++;
                    l = b;
                }
                CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
                ok = check("plan");
            }
            printf("  T1 %10d T2 %10d wgsX %3d gridS %4d sleep %d: %8.1f us  (%.2f us per level; launches: %d level, %d sync-free, %d xcd) %s\n",
                   P.T1, P.T2, P.wgsX, P.gridS, P.sleep, best * 1e3, best * 1e3 / levels, nL, nS, nX, ok ? "" : "(WRONG)");
        }
        (void)hipFree(d_order); (void)hipFree(d_idx); (void)hipFree(d_diag); (void)hipFree(d_val); (void)hipFree(d_xin); (void)hipFree(d_xout); (void)hipFree(d_len);
    }
    return 0;
}
