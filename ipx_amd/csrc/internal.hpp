// Internal declarations shared by the translation units of libipx_kkt_hip.so.
// Nothing here is part of the ABI (see include/ipx_kkt_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ipx_kkt_hip.h"

namespace ipxk {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

void set_last_error(const std::string& msg);

#define IPXK_HIP(expr)                                                         \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            char buf_[512];                                                    \
            snprintf(buf_, sizeof buf_, "%s:%d: %s failed: %s", __FILE__,      \
                     __LINE__, #expr, hipGetErrorString(e_));                  \
            throw ::ipxk::Error(e_ == hipErrorOutOfMemory ? IPXK_E_ALLOC       \
                                                          : IPXK_E_HIP, buf_); \
        }                                                                      \
    } while (0)

#define IPXK_REQUIRE(cond, msg)                                                \
    do {                                                                       \
        if (!(cond))                                                           \
            throw ::ipxk::Error(IPXK_E_ARGUMENT, std::string(__func__) + ": " + (msg)); \
    } while (0)

// ---------------------------------------------------------------------------
// host <-> device copies of pageable memory
// ---------------------------------------------------------------------------
// hipMemcpyAsync on pageable memory pins the pages on demand: measured 8-36 ms per call on the
// MI355X box even for a few MB (profiles/, HIP API trace of the host-pointer path).  All copies of
// caller-owned memory therefore go through two pinned staging buffers owned by the library
// (memcpy into pinned memory, DMA from there, double-buffered in 8 MiB chunks).
// On return the host data has been fully consumed (h2d) / is complete (d2h).
void staged_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);
void staged_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s);

// ---------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------
template <class T>
class DevBuf {
public:
    DevBuf() = default;
    explicit DevBuf(size_t n) { resize(n); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void resize(size_t n) {
        if (n == n_) return;
        release();
        if (n > 0) IPXK_HIP(hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T)));
        // IPXK_POISON=1 (debugging aid): fresh buffers of doubles hold NaNs, so that a read of something never written shows
        if (n > 0 && std::is_same<T, double>::value) {
            static const bool poison = getenv("IPXK_POISON") != nullptr;
            if (poison) { IPXK_HIP(hipMemset(p_, 0xff, n * sizeof(T))); IPXK_HIP(hipDeviceSynchronize()); }   // (before anybody's stream writes it)
        }
        n_ = n;
    }
    // grow-only variant of resize: keeps the allocation when it is already large enough
    void ensure(size_t n) { if (n > n_) resize(n); }
    void release() {
        if (p_) (void)hipFree(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T* get() const { return p_; }
    size_t size() const { return n_; }
    void upload(const T* host, size_t n, hipStream_t s) {
        if (n > n_) resize(n);
        if (n) staged_h2d(p_, host, n * sizeof(T), s);
    }
    void upload(const std::vector<T>& v, hipStream_t s) { upload(v.data(), v.size(), s); }
    void download(T* host, size_t n, hipStream_t s) const {
        if (n) staged_d2h(host, p_, n * sizeof(T), s);
    }
private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

// ---------------------------------------------------------------------------
// "gather matrix": a sparse matrix stored by output row, out[r] = f(sum_p
// x[idx[p]] * val[p]).  The CSC of A is the gather matrix of A' (pass 1 of
// A W A'), the row-wise copy of A is the gather matrix of A (pass 2).
// 32-bit indices on the device (4 B per index in the roofline model).
//
// Layout ("time-tiled, LDS-staged partial rows").  Measured on MI355X
// (profiles/r01_gather_microbench.txt): random 8-byte gathers run at ~190 G/s
// when the gathered vector slice is L2-resident (<= 2 MB) but only 70-98 G/s from
// a 16 / 8 MB vector, while the (idx,val) stream alone runs at 6.6 TB/s.  So the
// entries are re-ordered such that, at any time, every workgroup gathers from the
// same slice of x:
//   * the gathered index space is cut into P "phases" of kSliceElems entries;
//   * G co-resident workgroups each own kBlock*RT consecutive output rows per
//     round (thread t owns RT consecutive rows, accumulators live in registers);
//   * storage is ordered (round, phase, workgroup): in step (q,p,w) workgroup w
//     streams -- with fully coalesced loads -- the entries of its rows whose index
//     falls into phase p, stages the products in LDS and each thread adds its
//     rows' products in storage order.  Per (row, phase) only an 8-bit count is
//     stored (no row pointers).
// Because phases ascend with the index, a row is still summed in ascending index
// order, i.e. in the reference's order (bit-exact row sums).
// Rows with more than kMaxRowLen entries ("long rows": dense columns) are cut into
// segments of kLongSeg entries that separate workgroups sum into `long_partials`,
// combined in segment order by a fix-up kernel (deterministic, atomic-free).
// ---------------------------------------------------------------------------
constexpr int kBlock = 256;        // threads per workgroup (4 waves)
constexpr int kChunkNnz = 1024;    // products staged in LDS at a time (4 per thread)
constexpr int kMaxRowLen = 255;    // longer rows take the long-row path
constexpr int kLongSeg = 2048;     // nonzeros per segment of a long row
constexpr int kMaxWorkgroups = 1024; // 4 per CU: all co-resident (16 waves/CU)
constexpr int kMaxPartials = 2048; // upper bound on grid size of reducing kernels
constexpr int kMaxRT = 8;          // rows per thread per round

struct GatherView {                // passed to kernels by value
    int nrows, ncols;
    int P, G, RT, Q;               // phases, workgroups, rows/thread, rounds
    int RWrows;                    // rows owned by a workgroup per round (<= kBlock*RT)
    const int* step_ptr;           // [Q*P*G+1] first entry of each step
    // per-workgroup chunk table: the non-empty steps of (round q, workgroup w) cut into
    // chunks of at most kChunkNnz entries, in phase order
    const int* wg_chunk_ptr;       // [Q*G+1]
    const int* chunk_start;        // first entry of the chunk
    const int* chunk_info;         // # entries | (first chunk of its step) << 30
    const int* chunk_step;         // step index (addresses the counts)
    const unsigned char* counts;   // [Q*P*G*kBlock*RT] entries per (row, phase)
    const int* idx;                // [nnz_short] gathered index
    const double* val;             // [nnz_short]
    const unsigned char* row_long; // [nrows] 1 for long rows, nullptr if there are none
    // long rows
    int nseg;                      // # segments (= workgroups of the long kernel)
    const int* seg_p0;             // [nseg] first entry in lidx/lval
    const int* seg_p1;
    const int* lidx;
    const double* lval;
    int nlong;                     // # long rows
    const int* long_row;           // [nlong] row index
    const int* long_slot;          // [nlong+1] segment range of each long row
    double* long_partials;         // [nseg]
    unsigned long long* stamps;    // tuning aid: wall clock at the start of each step (or nullptr)
    int masked;                    // val / lval are masked copies: entries with value zero are not gathered
};

// XCD-sliced tile layout (alternative to the phased layout for large gathered vectors).  The
// columns are cut into `nslices` contiguous slices of at most ~2 MiB of x, the rows into blocks of
// kSlicedRows.  Tile (rb, s) holds the entries of row block rb whose column lies in slice s, row by
// row in storage order; it is processed by workgroup rb*nslices + s, which under the round-robin
// workgroup -> XCD dispatch of gfx950 runs on XCD (rb*nslices + s) % 8: an XCD only ever gathers
// from the slice(s) congruent to its index, which therefore stay resident in its 4 MiB L2 no
// matter how far workgroups drift apart -- x is fetched from HBM about once instead of once per
// XCD.  (Placement affects speed only; the result does not depend on it.)  A tile writes one
// partial sum per row; a second streaming kernel adds the slices' partials in slice order and
// applies the epilogue.  Within a slice a row is summed in storage order, so results differ from
// the phased layout only by the association across slices (~1 ulp).
constexpr int kSlicedRows = 1024;      // rows per tile: 4 per thread, or 2 / 1 when tiles would not fit
constexpr int kSlicedMaxTile = 7680;   // entries per tile that fit the LDS staging buffer (62 KB)

struct SlicedView {
    int nrows, nrows_pad, nslices, nrb;
    int R;                             // rows per tile (kBlock * 1, 2 or 4)
    const unsigned* tile_ptr;          // [nrb*nslices + 1] first entry of each tile
    const unsigned char* cnt;          // [nrb*nslices][R] entries per row of the tile
    const int* idx;
    const double* val;
    double* partial;                   // [nslices][nrows_pad]
    const unsigned char* row_long;     // [nrows] 1 for long rows (handled by the long-row kernels), or nullptr
    int masked;                        // val is a masked copy: entries with value zero are not gathered
};

struct SlicedMatrix {
    bool built = false;
    int nslices = 0, nrb = 0, nrows_pad = 0, max_tile = 0, R = kSlicedRows;
    double dominant_fraction = 1.0;    // share of the entries that lie in their row block's fullest slice
    DevBuf<unsigned> tile_ptr;
    DevBuf<unsigned char> cnt;
    DevBuf<int> idx;
    DevBuf<double> val, partial;
};

// Sorted sub-tiles: the sliced layout with the gathers of a tile issued in ADDRESS order.
// Measured (profiles/r03_gather_sorted_microbench.txt): random 8-byte gathers from an L2-resident slice cost
// ~41 us per 16M for issue + ~48 us per 16M L1->L2 line requests; lanes of one instruction (and consecutive
// instructions of a CU) that fall on the same 128-byte line share a request, so a tile whose entries are sorted by
// gathered index needs only as many requests as it touches lines: 0.63 of its entries when it holds as many
// entries as its slice has lines (72 instead of 88 us), 0.43 at two entries per line (60 us).  Layout:
//   * row blocks of RB = 256 * RPT rows (8192 at C3), slices as in the sliced layout, every slice cut into `nsub`
//     sub-slices; sub-tile (rb, s, h) holds the entries of the block in sub-slice h of slice s SORTED by gathered
//     index, at most kSortedMaxSub of them;
//   * an entry is one 32-bit word (slot << 18 | index - first index of the slice) + its value: slot = the place of
//     the entry when the sub-tile's entries are listed row by row in storage order -- the product goes to that LDS
//     slot, and after one barrier every thread adds up the products of its RPT consecutive rows from consecutive
//     slots, continuing the running sums of the previous sub-tile: a row's sum is formed slice by slice in storage
//     order exactly as in the sliced layout (bit-identical partial sums), one workgroup per (rb, s) -- few, large
//     workgroups that stream their entries batch after batch without a barrier in between (measured: two
//     workgroups of 1024 threads per CU that process a sub-tile in one batch between two barriers lose more to
//     the exposed load -> gather -> stage chain than the sorted gathers win);
//   * partial vectors and the combine kernel are the sliced layout's.
constexpr int kSortedThreads = 256;
constexpr int kSortedMaxSub = 8192;    // entries per sub-tile: 13 bits of slot
constexpr int kSortedOffBits = 18;     // index inside a slice of at most 2 MiB of x
//   * FUSED form (one slice = all of x, one sub-tile per row block; for matrices whose gathers have locality): the
//     gathered index is stored relative to the tile's smallest index (`xmin`, the tile's window must span less
//     than 2^18 entries), the thread starts each row from the epilogue's initial value, adds the row's products
//     in storage order and applies the epilogue itself -- bit-identical to the phased and fused layouts; a banded
//     tile touches a few hundred lines with thousands of entries, so almost all its gathers share requests.
struct SortedView {
    int nrows, nrows_pad, nslices, nsub, nrb, RB, slice_elems;
    const int* xmin;                   // FUSED: [nrb] smallest gathered index of the tile; else nullptr
    const unsigned char* row_long;     // FUSED: rows left to the long-row kernels, or nullptr
    const unsigned* sub_ptr;           // [nrb*nslices*nsub + 1]
    const unsigned char* cnt;          // [nrb*nslices*nsub][RB] entries per row of the sub-tile
    const unsigned* pack;
    const double* val;
    double* partial;                   // [nslices][nrows_pad]
};
struct SortedMatrix {
    bool built = false;
    int nslices = 0, nsub = 0, nrb = 0, RB = 0, nrows_pad = 0, max_sub = 0, slice_elems = 0;
    DevBuf<unsigned> sub_ptr, pack;
    DevBuf<unsigned char> cnt;
    DevBuf<double> val, partial;
    bool fused = false;                // the FUSED form
    DevBuf<int> xmin;
};

// Accumulated tiles (round 4; the headline's layout): the sliced layout's slices with the ROW SUMS of a row block kept
// in LDS.  Measured first (scripts/bench_ldsacc.hip, profiles/r04_ldsacc_microbench.txt): what the sorted sub-tiles pay
// for staging every product in an LDS slot (a byte count per row and sub-tile, a scan, a row-by-row sum phase, running
// sums in registers that cap the row block at 8192 rows) costs more than the gathers; with nothing but RB doubles of
// LDS per workgroup the row block grows to 16384 rows, a tile holds two entries per 128-byte line of its slice, and
// the sum phase disappears:
//   * tile (rb, s) = the entries of RB rows whose gathered index lies in slice s, in ascending order of the gathered
//     ADDRESS over the whole slice (ties in storage order), cut into BATCHES of at most kAccBatch entries;
//   * a batch holds at most ONE entry of a row: walking the entries in address order, an entry whose row already has
//     one in the current batch waits for the next batch (it goes first there), and a batch is closed when it is full
//     or nothing else is left -- so the ds_add_f64 of a batch hit distinct addresses, one workgroup barrier separates
//     two batches, and a row's products are added in ascending address order whatever the thread timing:
//     deterministic, and equal to the storage order (hence to the sliced layout's partial sums, bit for bit) when the
//     rows are stored with ascending indices, as the reference's Transpose and its sorted CSC give them;
//   * an entry is one 32-bit word (row in block << 18 | index - first index of the slice) + its value; batch q of
//     tile t is entries [bptr[tb[t] + q], bptr[tb[t] + q + 1]);
//   * partial vectors and the combine kernel are the sliced layout's (folding the partials into the last workgroup of
//     a row block to finish measured SLOWER than the combine launch: 109-145 against 95-101 us per pass).
constexpr int kAccThreads = 512;
constexpr int kAccPerThread = 4;
constexpr int kAccBatch = kAccThreads * kAccPerThread;     // 2048 entries between two barriers
constexpr int kAccMaxRows = 16384;                         // 128 KB of row sums; 14 bits of row in the entry word
//   * FUSED form (one slice = all of x, for matrices whose gathers have locality; round 4): a tile is a row block of RB rows
//     (as many as give every CU two tiles), the gathered index is stored relative to the tile's smallest one (`xmin`; the
//     tile's window of x must span less than 2^18 entries), the row sums start from the epilogue's initial value and the
//     tile kernel applies the epilogue itself: no partial vectors, no combine launch.  Rows must be stored with ascending
//     indices (then a row is summed in storage order: bit-identical to the phased, fused and sorted fused layouts, so the
//     timing at ipxk_create may choose among them).
struct AccView {
    int nrows, nrows_pad, nslices, nrb, RB, slice_elems;
    const int* xmin;                   // FUSED: [nrb] smallest gathered index of the tile; else nullptr
    const unsigned* tile_batch;        // [nrb*nslices + 1] first batch of each tile
    const unsigned* bptr;              // [# batches + 1] first entry of each batch
    const unsigned* pack;
    const double* val;
    double* partial;                   // [nslices][nrows_pad]
};
struct AccMatrix {
    bool built = false;
    int nslices = 0, nrb = 0, RB = 0, nrows_pad = 0, slice_elems = 0;
    int64_t nbatches = 0, deferred = 0;
    DevBuf<unsigned> tile_batch, bptr, pack;
    DevBuf<double> val, partial;
    bool fused = false;                // the FUSED form
    DevBuf<int> xmin;
};

struct LayoutScratch;                 // layout_device.hip

struct GatherMatrix {
    int nrows = 0, ncols = 0;
    int64_t nnz = 0;
    int P = 1, G = 1, RT = 1, Q = 1, RWrows = kBlock;
    DevBuf<int> step_ptr, idx, wg_chunk_ptr, chunk_start, chunk_info, chunk_step;
    DevBuf<unsigned char> counts, row_long;
    DevBuf<double> val;
    int nseg = 0, nlong = 0;
    DevBuf<int> seg_p0, seg_p1, lidx, long_row, long_slot;
    DevBuf<double> lval, long_partials;
    DevBuf<unsigned long long> stamps;   // allocated only when IPXK_STAMPS=1
    // how much of the layout choice build() pays for: 2 = every layout built and timed (the model matrix: once per
    // model), 1 = phased against fused only, 0 = the phased layout, no timing (auxiliary matrices built inside a
    // Factorize: the dense columns' gather matrices)
    int tune_level = 2;
    // optional copy in plain row order (ptr/idx/val) for callers that address single rows
    bool keep_plain = false;
    std::vector<int> h_plain_ptr;
    DevBuf<int> plain_idx;
    DevBuf<double> plain_val;

    // Builds from host arrays with 64-bit indices (ptr has nrows+1 entries).
    void build(int64_t nrows_, int64_t ncols_, const ipxint* hptr, const ipxint* hidx,
               const double* hval, hipStream_t s);
    // Builds from the plain device copy (ptr / idx 32 bits) with radix sorts on the device (layout_device.hip): the sliced
    // and the sorted layout, every array equal to build()'s.  Returns false -- nothing built -- when the matrix is not
    // one of those the device path covers (long rows, a gathered vector that fits an XCD's L2, gathers that do not
    // spread over the slices): the caller then runs build() on host arrays.
    bool build_device(LayoutScratch& S, int64_t nrows_, int64_t ncols_, int64_t nnz_, const int* dptr, const int* didx,
                      const double* dval, hipStream_t s);
    bool build_device_local(LayoutScratch& S, int64_t nrows_, int64_t ncols_, int64_t nnz_, const int* dptr, const int* didx,
                            const double* dval, double share, hipStream_t s);
    void set_geometry(int64_t nrows_, int64_t ncols_);      // P, G, RT, Q, RWrows of the phased layout
    // optional second layout and the choice between the two (IPXK_SPMV_LAYOUT=phased|sliced|auto;
    // auto times both once at build time on this matrix and keeps the faster one)
    SlicedMatrix sliced;
    bool use_sliced = false;
    // the sliced layout's tiles with sorted gathers (bit-identical results: a timing may choose between the two)
    SortedMatrix sorted;
    bool use_sorted = false;           // only with use_sliced and sliced.nslices > 1
    void build_sorted(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s);
    // accumulated tiles (the sliced layout's slices, row sums in LDS); use_acc: the layout in use for unmasked products
    AccMatrix acc;
    bool use_acc = false;
    float tuned_us_acc = 0.f;
    void build_acc(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s);     // host builder (test reference)
    void build_acc_fused(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s);
    AccMatrix accf;                    // the FUSED form (an overlay for the unmasked launches, like the sorted fused tiles)
    bool use_acc_fused = false;
    float tuned_us_acc_fused = 0.f;
    AccView acc_fused_view() const;
    AccView acc_view() const;
    // plain rows (small matrices): the device's plain copy of the matrix itself, 8 lanes per row (spmv_rowgroup_kernel); an
    // overlay for the unmasked launches like the sorted fused tiles; set by the caller of build() (the model matrices)
    const int* csr_ptr = nullptr;
    const int* csr_idx = nullptr;
    const double* csr_val = nullptr;
    bool use_plain = false;
    float tuned_us_plain = 0.f;
    int plain_grid() const { return (int)std::min<int64_t>(kMaxPartials, std::max<int64_t>(1, ((int64_t)nrows * 8 + kBlock - 1) / kBlock)); }
    // the FUSED form (independent of the sliced layout); use_sorted_fused: it is the layout in use
    void build_sorted_fused(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s);
    bool use_sorted_fused = false;
    float tuned_us_sorted_fused = 0.f;
    int sorted_fused_grid() const {
        static const int cap = [] { const char* e = getenv("IPXK_SF_GRID"); return e && atoi(e) > 0 ? std::min(atoi(e), kMaxPartials) : kMaxPartials; }();
        return std::min(sorted.nrb, cap);
    }
    SortedView sorted_view() const;
    float tuned_us_phased = 0.f, tuned_us_sliced = 0.f, tuned_us_fused = 0.f, tuned_us_sorted = 0.f;
    // ns_request: 0 = as many slices as x needs (>= 2), 1 = the fused single-slice variant
    void build_sliced(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s, int ns_request);
    // which: 0 = the matrix, 1 = the masked value array (mask_values), 2 = the compacted copy (compact_tiles)
    SlicedView sliced_view(int which = 0) const;
    int fused_grid() const { return std::min(sliced.nrb, kMaxPartials); }
    int combine_grid() const { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, ((int64_t)nrows + kBlock - 1) / kBlock)); }

    GatherView view(bool masked = false) const;
    int grid() const { return G; }
    // Masked products (the basis path's N N' on the model matrix: entries of BASIC / fixed columns count for
    // nothing): a second value array of the layout in use in which those entries are zero -- the kernels issue no
    // gather for an entry whose value is zero.  mask_values() fills it from a weight per ROW of the gather matrix
    // (by_row) or per GATHERED index; view(true) / sliced_view(true) show it (launch_spmv<Epi, true>).
    DevBuf<double> valM, lvalM;
    // Compacted copy of the tile layout (sliced / fused): only the entries whose weight is nonzero, tile by tile
    // in the same order, with their own tile pointers and per-row counts -- the basis path's N as a matrix of its
    // own (SplittedNormalMatrix::Prepare copies N, src/splitted_normal_matrix.cc:42-55; here the copy is a stream
    // compaction of the resident tiles on the device).  Same arithmetic as the masked form (the entries left out
    // contributed exact zeros), but the kernels no longer stream them.  Long rows keep their masked values.
    struct CompactTiles {
        bool valid = false;
        DevBuf<unsigned> tile_ptr, tile_kept;
        DevBuf<unsigned char> cnt;
        DevBuf<int> idx;
        DevBuf<double> val;
    } compact;
    void compact_tiles(const double* weight, bool by_row, hipStream_t s);
    DevBuf<int> rowof;                  // row of every stored short entry (built on first use by mask_values(by_row))
    void mask_values(const double* weight, bool by_row, hipStream_t s);
    // # dot partials a launch of the phased / sliced / fused layouts produces (the sorted-fused overlay: launch_spmv)
    int num_partials() const {
        const int extra = nlong > 0 ? 1 : 0;      // the long-row fix-up kernel adds one
        if (!use_sliced) return G + extra;
        return (sliced.nslices == 1 ? fused_grid() : combine_grid()) + extra;   // fused: one dot partial per workgroup
    }
    std::vector<unsigned char> h_row_long;   // host copy of row_long (empty: no long rows)
};

// elements of the gathered vector per phase (IPXK_SLICE_KB overrides, default 1 MiB)
int slice_elems();

// ---------------------------------------------------------------------------
// CR loop state kept on the device (see cr.hip)
// ---------------------------------------------------------------------------
struct CrState {
    double tol;
    long long maxiter;
    long long k_started;     // written by the control kernel, read by the others
    long long k_finished;    // written by the direction-update kernel
    double cdot[2];          // indexed by iteration parity
    double rps[2];           // resnorm_precond_system, indexed by (iter/5) parity
    double resnorm;          // residual norm at the last loop head
    long long iter;          // result: # iterations
    int errflag;             // result
    int done;
    long long hist_cap;
    double diag_rps_old, diag_rps_new;   // errflag 204: the two values of the monotonicity check
    int mode;                            // CrMode of the run (cr.hip)
};

struct Context;

}  // namespace ipxk
