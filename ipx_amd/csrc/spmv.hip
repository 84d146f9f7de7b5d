// Gather-matrix construction (host) and the NormalMatrix product
//   lhs = AI * W * AI' * rhs                 reference src/normal_matrix.cc:45-126
// as two row-gather SpMVs:  t = Ws .* (A' rhs);  lhs = W_I .* rhs + A t  with the
// dot product rhs'lhs fused into the second pass (src/normal_matrix.cc:123-124).
#include <algorithm>
#include <chrono>
#include <cstdlib>

#include <cstdio>
#include <string>

#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

// ---------------------------------------------------------------------------
// GatherMatrix
// ---------------------------------------------------------------------------
int slice_elems() {
    static int cached = 0;
    if (!cached) {
        int kb = 1024;
        if (const char* e = getenv("IPXK_SLICE_KB")) kb = atoi(e) > 0 ? atoi(e) : kb;
        cached = kb * 128;   // doubles
    }
    return cached;
}

// elements of the gathered vector per phase of the phased layout: ~1 MiB of x; at most kMaxPhases phases (the
// per-(row,phase) count table grows with P), so very long vectors get proportionally larger slices
static int64_t phase_slice(int64_t ncols_) {
    constexpr int kMaxPhases = 64;
    int64_t slice = slice_elems();
    if ((ncols_ + slice - 1) / slice > kMaxPhases) slice = (ncols_ + kMaxPhases - 1) / kMaxPhases;
    return slice;
}

void GatherMatrix::set_geometry(int64_t nrows_, int64_t ncols_) {
    const int64_t slice = phase_slice(ncols_);
    P = (int)std::max<int64_t>(1, (ncols_ + slice - 1) / slice);
    int maxwg = kMaxWorkgroups;
    if (const char* e = getenv("IPXK_MAX_WG")) maxwg = atoi(e) > 0 ? atoi(e) : maxwg;
    G = (int)std::min<int64_t>(maxwg, std::max<int64_t>(1, (nrows_ + kBlock - 1) / kBlock));
    RT = 1;
    while (RT < kMaxRT && (int64_t)G * kBlock * RT < nrows_) RT *= 2;
    const int64_t RW = (int64_t)kBlock * RT;     // count slots per step
    // rows a workgroup owns per round: all RW slots, or -- for matrices too small to give
    // every CU several workgroups that way -- fewer (threads without a row still stream)
    RWrows = (int)RW;
    if (RT == 1 && nrows_ < (int64_t)kBlock * maxwg) {
        const int64_t want = (nrows_ + maxwg - 1) / maxwg;
        RWrows = (int)std::min<int64_t>(kBlock, std::max<int64_t>(32, (want + 31) / 32 * 32));
    }
    G = (int)std::max<int64_t>(1, std::min<int64_t>(maxwg, (nrows_ + RWrows - 1) / RWrows));
    Q = (int)std::max<int64_t>(1, (nrows_ + (int64_t)G * RWrows - 1) / ((int64_t)G * RWrows));
}

void GatherMatrix::build(int64_t nrows_, int64_t ncols_, const ipxint* hptr, const ipxint* hidx,
                         const double* hval, hipStream_t s) {
    IPXK_REQUIRE(nrows_ >= 0 && ncols_ >= 0, "negative dimension");
    IPXK_REQUIRE(nrows_ < (int64_t(1) << 31) - 1 && ncols_ < (int64_t(1) << 31) - 1,
                 "dimension exceeds 32-bit device indices");
    const int64_t nz = hptr[nrows_];
    IPXK_REQUIRE(nz < (int64_t(1) << 31) - kLongSeg, "nnz exceeds 32-bit device indices");
    nrows = (int)nrows_;
    ncols = (int)ncols_;
    nnz = nz;

    set_geometry(nrows_, ncols_);
    const int64_t slice = phase_slice(ncols_);
    const int64_t RW = (int64_t)kBlock * RT;     // count slots per step
    const int64_t nsteps = (int64_t)Q * P * G;
    IPXK_REQUIRE(nsteps * RW < (int64_t(1) << 40), "matrix too large for the phased layout");

    // long rows
    std::vector<unsigned char> rlong;
    std::vector<int> sp0, sp1, lrow, lslot, li;
    std::vector<double> lv;
    for (int r = 0; r < nrows; r++) {
        const int64_t len = hptr[r + 1] - hptr[r];
        if (len <= kMaxRowLen) continue;
        if (rlong.empty()) rlong.assign(nrows, 0);
        rlong[r] = 1;
        lrow.push_back(r);
        lslot.push_back((int)sp0.size());
        for (int64_t q0 = hptr[r]; q0 < hptr[r + 1]; q0 += kLongSeg) {
            const int64_t q1 = std::min<int64_t>(q0 + kLongSeg, hptr[r + 1]);
            sp0.push_back((int)li.size());
            for (int64_t p = q0; p < q1; p++) { li.push_back((int)hidx[p]); lv.push_back(hval[p]); }
            sp1.push_back((int)li.size());
        }
    }
    lslot.push_back((int)sp0.size());
    nlong = (int)lrow.size();
    nseg = (int)sp0.size();
    h_row_long = rlong;

    // counts per (row, phase) and step sizes
    std::vector<unsigned char> cnt((size_t)nsteps * RW, 0);
    std::vector<int> sptr(nsteps + 1, 0);
    auto step_of = [&](int r, int p, int64_t& lr) {
        const int64_t per = (int64_t)G * RWrows;
        const int64_t q = r / per, rem = r % per;
        const int64_t w = rem / RWrows;
        lr = rem % RWrows;
        return (q * P + p) * G + w;
    };
    for (int r = 0; r < nrows; r++) {
        if (!rlong.empty() && rlong[r]) continue;
        for (int64_t p = hptr[r]; p < hptr[r + 1]; p++) {
            int64_t lr;
            const int64_t st = step_of(r, (int)(hidx[p] / slice), lr);
            cnt[(size_t)st * RW + lr]++;
            sptr[st + 1]++;
        }
    }
    for (int64_t st = 0; st < nsteps; st++) sptr[st + 1] += sptr[st];
    const int64_t nshort = sptr[nsteps];
    std::vector<int> i32((size_t)std::max<int64_t>(nshort, 1));
    std::vector<double> v64((size_t)std::max<int64_t>(nshort, 1));
    {
        std::vector<int> cursor(sptr.begin(), sptr.end() - 1);
        for (int r = 0; r < nrows; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            for (int64_t p = hptr[r]; p < hptr[r + 1]; p++) {
                int64_t lr;
                const int64_t st = step_of(r, (int)(hidx[p] / slice), lr);
                const int put = cursor[st]++;
                i32[put] = (int)hidx[p];
                v64[put] = hval[p];
            }
        }
    }

    // chunk table
    std::vector<int> wcp((size_t)Q * G + 1, 0), cst, cinf, cstep;
    for (int q = 0; q < Q; q++)
        for (int w = 0; w < G; w++) {
            for (int p = 0; p < P; p++) {
                const int64_t st = ((int64_t)q * P + p) * G + w;
                for (int c0 = sptr[st]; c0 < sptr[st + 1]; c0 += kChunkNnz) {
                    cst.push_back(c0);
                    cinf.push_back(std::min(kChunkNnz, sptr[st + 1] - c0) | (c0 == sptr[st] ? (1 << 30) : 0));
                    cstep.push_back((int)st);
                }
            }
            wcp[(size_t)q * G + w + 1] = (int)cst.size();
        }
    for (int pad = 0; pad < 4; pad++) { cst.push_back(0); cinf.push_back(0); cstep.push_back(0); }
    wg_chunk_ptr.upload(wcp, s);
    chunk_start.upload(cst, s);
    chunk_info.upload(cinf, s);
    chunk_step.upload(cstep, s);
    step_ptr.upload(sptr, s);
    counts.upload(cnt, s);
    idx.upload(i32, s);
    val.upload(v64, s);
    if (nlong > 0) {
        row_long.upload(rlong, s);
        seg_p0.upload(sp0, s);
        seg_p1.upload(sp1, s);
        lidx.upload(li, s);
        lval.upload(lv, s);
        long_row.upload(lrow, s);
        long_slot.upload(lslot, s);
    }
    long_partials.resize(nseg > 0 ? nseg : 1);
    if (getenv("IPXK_STAMPS")) stamps.resize((size_t)nsteps + G);
    if (keep_plain) {
        h_plain_ptr.resize(nrows + 1);
        std::vector<int> pi((size_t)std::max<int64_t>(nz, 1));
        for (int r = 0; r <= nrows; r++) h_plain_ptr[r] = (int)hptr[r];
        for (int64_t p = 0; p < nz; p++) pi[p] = (int)hidx[p];
        plain_idx.upload(pi, s);
        plain_val.upload(hval, (size_t)nz, s);
    }
    IPXK_HIP(hipStreamSynchronize(s));  // host vectors go out of scope

    // alternative layouts + choice
    use_sliced = false;
    sliced = SlicedMatrix();
    std::string layout = "auto";
    if (const char* e = getenv("IPXK_SPMV_LAYOUT")) layout = e;
    if (layout == "phased") return;
    use_sorted = false;
    use_sorted_fused = false;
    sorted = SortedMatrix();
    use_acc = false;
    acc = AccMatrix();
    use_plain = false;
    use_acc_fused = false;
    accf = AccMatrix();
    if (layout == "plain") { use_plain = csr_ptr != nullptr; return; }
    if (layout == "accfused") { build_acc_fused(hptr, hidx, hval, s); use_acc_fused = accf.built; return; }
    if (layout == "acc") {
        build_sliced(hptr, hidx, hval, s, 0);
        use_sliced = sliced.built;
        build_acc(hptr, hidx, hval, s);
        use_acc = acc.built;
        return;
    }
    if (layout == "sortedfused") {
        build_sorted_fused(hptr, hidx, hval, s);
        use_sorted_fused = sorted.built;
        return;
    }
    if (layout == "sliced" || layout == "fused" || layout == "sorted") {
        build_sliced(hptr, hidx, hval, s, layout == "fused" ? 1 : 0);
        use_sliced = sliced.built;
        if (layout == "sorted") { build_sorted(hptr, hidx, hval, s); use_sorted = sorted.built; }
        return;
    }
    // auto.  The phased and the fused layout add a row's products in the same (the reference's) order and give
    // bit-identical results, so a timing on the spot may choose between them (the phased layout wins ties,
    // 5 % margin).  The sliced layout associates a row's sum per slice, hence whether it is used must not
    // depend on a timing: it is chosen by a property of the matrix alone -- x does not fit an XCD's L2 and
    // the gathers of a row block spread over the slices (share of its fullest slice <= 1.5 / #slices: that
    // is where confining every XCD to one slice of x pays; measured, C3-sized: uniformly random indices 256
    // against 300 us per apply, banded ones 240-400 against 190-220).  Timings of all three are recorded.
    DevBuf<double> tx((size_t)std::max(ncols, 1)), tout((size_t)std::max(nrows, 1));
    IPXK_HIP(hipMemsetAsync(tx.get(), 0, tx.size() * sizeof(double), s));
    hipEvent_t e0, e1;
    IPXK_HIP(hipEventCreate(&e0));
    IPXK_HIP(hipEventCreate(&e1));
    EpiScale epi{{}, nullptr, tout.get()};
    auto time_current = [&]() {
        const int reps = 5;
        for (int w = 0; w < 2; w++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e1, s));
        IPXK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    // small matrices: a few microseconds either way, not worth two more copies of the matrix
    if (nnz >= (1 << 16) && tune_level > 0) {
        tuned_us_phased = time_current();
        SlicedMatrix fusedm, slicedm;
        build_sliced(hptr, hidx, hval, s, 1);
        if (sliced.built) { use_sliced = true; tuned_us_fused = time_current(); use_sliced = false; fusedm = std::move(sliced); }
        sliced = SlicedMatrix();
        if (tune_level > 1) build_sliced(hptr, hidx, hval, s, 0);
        if (sliced.built) { use_sliced = true; tuned_us_sliced = time_current(); use_sliced = false; slicedm = std::move(sliced); }
        sliced = SlicedMatrix();
        const double share = slicedm.built ? slicedm.dominant_fraction : 1.0;
        const bool spread = slicedm.built && share <= 1.5 / slicedm.nslices;
        if (spread) sliced = std::move(slicedm);
        else if (fusedm.built && tuned_us_fused < 0.95f * tuned_us_phased) sliced = std::move(fusedm);
        use_sliced = sliced.built;
        if (!spread && tune_level > 1 && !(getenv("IPXK_SPMV_SORTED") && getenv("IPXK_SPMV_SORTED")[0] == '0')) {
            // gathers with locality: the fused tiles with the gathers in address order (bit-identical to the phased
            // and fused layouts, so a timing may choose); kept if it beats what the timing chose so far
            // (an overlay: masked products -- the basis path's N N' -- keep using the layout chosen above)
            build_sorted_fused(hptr, hidx, hval, s);
            if (sorted.built) {
                use_sorted_fused = true;
                tuned_us_sorted_fused = time_current();
                const float best = use_sliced ? tuned_us_fused : tuned_us_phased;
                if (!(tuned_us_sorted_fused < 0.95f * best)) { use_sorted_fused = false; sorted = SortedMatrix(); }
            }
        }
        if (spread && !(getenv("IPXK_SPMV_SORTED") && getenv("IPXK_SPMV_SORTED")[0] == '0')) {
            // the same slices with the gathers of a tile in address order: bit-identical partial sums, so the faster
            // of the two is kept (the sliced arrays stay: the basis path compacts them)
            build_sorted(hptr, hidx, hval, s);
            if (sorted.built) {
                use_sorted = true;
                tuned_us_sorted = time_current();
                use_sorted = tuned_us_sorted < tuned_us_sliced || (getenv("IPXK_SPMV_SORTED") && getenv("IPXK_SPMV_SORTED")[0] == '1');
                if (!use_sorted && !getenv("IPXK_BUILD_ALL_LAYOUTS")) sorted = SortedMatrix();
            }
        }
        if (!spread && tune_level > 1 && !(getenv("IPXK_SPMV_ACC") && getenv("IPXK_SPMV_ACC")[0] == '0')) {
            // gathers with locality, second candidate: the fused accumulated tiles (bit-identical to the layouts timed above for
            // rows stored with ascending indices -- build_acc_fused checks that); kept if it beats what was chosen so far
            build_acc_fused(hptr, hidx, hval, s);
            if (accf.built) {
                use_acc_fused = true;
                tuned_us_acc_fused = time_current();
                const float best = use_sorted_fused ? tuned_us_sorted_fused : use_sliced ? tuned_us_fused : tuned_us_phased;
                if (!(tuned_us_acc_fused < 0.95f * best)) { use_acc_fused = false; accf = AccMatrix(); }
                else if (use_sorted_fused) { use_sorted_fused = false; sorted = SortedMatrix(); }
            }
        }
        if (!spread && !use_acc_fused && csr_ptr && nnz <= (int64_t(4) << 20) && !(getenv("IPXK_SPMV_PLAIN") && getenv("IPXK_SPMV_PLAIN")[0] == '0')) {
            // small matrices: the plain rows, 8 lanes each (bit-identical to the three layouts timed above); kept if it beats them
            use_plain = true;
            tuned_us_plain = time_current();
            const float best = use_sorted_fused ? tuned_us_sorted_fused : use_sliced ? tuned_us_fused : tuned_us_phased;
            if (!(tuned_us_plain < 0.95f * best)) use_plain = false;
        }
        if (spread && !(getenv("IPXK_SPMV_ACC") && getenv("IPXK_SPMV_ACC")[0] == '0')) {
            // accumulated tiles: used whenever they can be built (never a timing decision: see build_device)
            build_acc(hptr, hidx, hval, s);
            if (acc.built) {
                use_acc = true;
                tuned_us_acc = time_current();
                if (!getenv("IPXK_BUILD_ALL_LAYOUTS")) { use_sorted = false; sorted = SortedMatrix(); }
            }
        }
        if (use_sliced) {        // the phased copy of the entries is not needed any more
            idx.release(); val.release(); counts.release(); step_ptr.release();
            wg_chunk_ptr.release(); chunk_start.release(); chunk_info.release(); chunk_step.release();
        }
        if (getenv("IPXK_VERBOSE"))
            fprintf(stderr, "ipxk: gather matrix %d x %d nnz %lld: phased %.1f us, fused %.1f us, sliced %.1f us, sorted %.1f us, plain rows %.1f us (fullest-slice share %.2f) -> %s\n",
                    nrows, ncols, (long long)nnz, tuned_us_phased, tuned_us_fused, tuned_us_sliced, tuned_us_sorted, tuned_us_plain, share,
                    use_acc_fused ? "accumulated-fused" : use_plain ? "plain rows" : use_sorted_fused ? "sorted-fused" : !use_sliced ? "phased" : sliced.nslices == 1 ? "fused" : use_sorted ? "sorted" : "sliced");
        if (getenv("IPXK_VERBOSE") && tuned_us_sorted_fused > 0.f) fprintf(stderr, "ipxk:   sorted-fused %.1f us\n", tuned_us_sorted_fused);
        if (getenv("IPXK_VERBOSE") && tuned_us_acc_fused > 0.f)
            fprintf(stderr, "ipxk:   accumulated-fused %.1f us (%d rows per tile, %lld batches, %.1f%% of the entries waited)\n", tuned_us_acc_fused, accf.RB,
                    (long long)accf.nbatches, accf.built ? 100.0 * (double)accf.deferred / (double)nnz : 0.0);
    }
    IPXK_HIP(hipEventDestroy(e0));
    IPXK_HIP(hipEventDestroy(e1));
}

// The device path for matrices whose gathers have locality, or whose gathered vector fits an XCD's L2 (round 4): the fused tiles
// (one slice; also what the masked products of the basis path use), the fused sorted tiles and the fused accumulated tiles, all
// built on the device and all bit-identical to the phased layout (the accumulated ones for sorted rows only: device_build_acc_fused
// checks), so a timing chooses among them; the plain rows join for small matrices.  The phased layout is not built on this path.
bool GatherMatrix::build_device_local(LayoutScratch& S, int64_t nrows_, int64_t ncols_, int64_t nnz_, const int* dptr, const int* didx,
                                      const double* dval, double share, hipStream_t s) {
    if (getenv("IPXK_LAYOUT_LOCAL") && getenv("IPXK_LAYOUT_LOCAL")[0] == 'h') return false;      // host builders for these matrices
    SlicedMatrix fu;
    if (!device_build_sliced(S, fu, (int)nrows_, (int)ncols_, nnz_, dptr, didx, dval, s, 1)) return false;
    nrows = (int)nrows_; ncols = (int)ncols_; nnz = nnz_;
    set_geometry(nrows_, ncols_);
    sliced = std::move(fu);                       // (nlong, nseg, ... : set by build_device)
    sliced.dominant_fraction = share;
    use_sliced = true;
    use_sorted = false; use_sorted_fused = false; use_acc = false; use_acc_fused = false; use_plain = false;
    sorted = SortedMatrix(); acc = AccMatrix(); accf = AccMatrix();
    DevBuf<double> tx((size_t)std::max(ncols, 1)), tout((size_t)std::max(nrows, 1));
    IPXK_HIP(hipMemsetAsync(tx.get(), 0, tx.size() * sizeof(double), s));
    hipEvent_t e0, e1;
    IPXK_HIP(hipEventCreate(&e0));
    IPXK_HIP(hipEventCreate(&e1));
    EpiScale epi{{}, nullptr, tout.get()};
    auto time_current = [&]() {
        const int reps = 5;
        for (int w = 0; w < 2; w++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e1, s));
        IPXK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    const bool keep_all = getenv("IPXK_BUILD_ALL_LAYOUTS") != nullptr;          // (tests: the layouts the timing discards stay built)
    tuned_us_fused = time_current();
    float best = tuned_us_fused;
    SortedMatrix so;
    if (device_build_sorted_fused(S, so, nrows, ncols, nnz_, dptr, didx, dval, s)) {
        sorted = std::move(so);
        use_sorted_fused = true;
        tuned_us_sorted_fused = time_current();
        if (tuned_us_sorted_fused < 0.95f * best) best = tuned_us_sorted_fused;
        else { use_sorted_fused = false; if (!keep_all) sorted = SortedMatrix(); }
    }
    AccMatrix af;
    if (nlong == 0 && !(getenv("IPXK_SPMV_ACC") && getenv("IPXK_SPMV_ACC")[0] == '0') && device_build_acc_fused(S, af, nrows, ncols, nnz_, dptr, didx, dval, s)) {
        accf = std::move(af);
        use_acc_fused = true;
        tuned_us_acc_fused = time_current();
        if (tuned_us_acc_fused < 0.95f * best) { best = tuned_us_acc_fused; use_sorted_fused = false; if (!keep_all) sorted = SortedMatrix(); }
        else { use_acc_fused = false; if (!keep_all) accf = AccMatrix(); }
    }
    if (!use_acc_fused && csr_ptr && nnz <= (int64_t(4) << 20) && !(getenv("IPXK_SPMV_PLAIN") && getenv("IPXK_SPMV_PLAIN")[0] == '0')) {
        const bool sf = use_sorted_fused;
        use_sorted_fused = false;
        use_plain = true;
        tuned_us_plain = time_current();
        if (tuned_us_plain < 0.95f * best) { best = tuned_us_plain; if (!keep_all) sorted = SortedMatrix(); }
        else { use_plain = false; use_sorted_fused = sf; }
    }
    IPXK_HIP(hipEventDestroy(e0));
    IPXK_HIP(hipEventDestroy(e1));
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: gather matrix %d x %d nnz %lld built on the device (gathers with locality): fused %.1f us, sorted-fused %.1f us, accumulated-fused %.1f us, "
                        "plain rows %.1f us (fullest-slice share %.2f) -> %s\n", nrows, ncols, (long long)nnz, tuned_us_fused, tuned_us_sorted_fused,
                tuned_us_acc_fused, tuned_us_plain, share,
                use_acc_fused ? "accumulated-fused" : use_plain ? "plain rows" : use_sorted_fused ? "sorted-fused" : "fused");
    return true;
}

// The device path (layout_device.hip): sliced + sorted layouts by radix sorts, for matrices whose gathered vector needs
// slicing and whose gathers spread over the slices -- the same decision build() takes from the same property of the
// matrix, so a model gets the same layouts whichever path builds them.
bool GatherMatrix::build_device(LayoutScratch& S, int64_t nrows_, int64_t ncols_, int64_t nnz_, const int* dptr, const int* didx,
                                const double* dval, hipStream_t s) {
    if (const char* e = getenv("IPXK_LAYOUT_BUILD")) if (e[0] == 'h') return false;           // host: the test reference
    if (const char* e = getenv("IPXK_SPMV_LAYOUT")) if (std::string(e) != "auto") return false;
    if (const char* e = getenv("IPXK_SPMV_SORTED")) if (e[0] == '0') return false;
    if (tune_level < 2 || keep_plain || nnz_ < (1 << 16) || getenv("IPXK_STAMPS")) return false;
    if (nrows_ >= (int64_t(1) << 31) - 1 || ncols_ >= (int64_t(1) << 31) - 1 || nnz_ >= (int64_t(1) << 31) - kLongSeg) return false;
    nlong = 0; nseg = 0;
    h_row_long.clear();
    long_partials.resize(1);
    // long rows go to the long-row kernels' arrays; the tile layouts are built from the matrix without them
    DevBuf<int> sptr, sidx;
    DevBuf<double> sval;
    const int64_t nnz_all = nnz_;
    if (device_max_row_length(S, (int)nrows_, dptr, s) > kMaxRowLen) {
        if (getenv("IPXK_LONG_ROWS_HOST")) return false;                                // (tests: the host builders as the reference)
        if (!device_strip_long_rows(S, *this, (int)nrows_, dptr, didx, dval, sptr, sidx, sval, &nnz_, s)) return false;
        dptr = sptr.get(); didx = sidx.get(); dval = sval.get();
        if (nnz_ < (1 << 16)) { nlong = 0; nseg = 0; h_row_long.clear(); return false; }
    }
    SlicedMatrix sl;
    const bool slices = device_build_sliced(S, sl, (int)nrows_, (int)ncols_, nnz_, dptr, didx, dval, s);
    if (!slices || !(sl.dominant_fraction <= 1.5 / sl.nslices)) {
        const bool ok = build_device_local(S, nrows_, ncols_, nnz_, dptr, didx, dval, slices ? sl.dominant_fraction : 1.0, s);
        nnz = nnz_all;
        return ok;
    }
    nrows = (int)nrows_; ncols = (int)ncols_; nnz = nnz_;
    set_geometry(nrows_, ncols_);
    sliced = std::move(sl);
    use_sliced = true;
    use_sorted = false; use_sorted_fused = false;
    sorted = SortedMatrix();
    acc = AccMatrix(); use_acc = false;
    const bool want_acc = !(getenv("IPXK_SPMV_ACC") && getenv("IPXK_SPMV_ACC")[0] == '0');
    AccMatrix ac;
    if (want_acc && device_build_acc(S, ac, sliced, nrows, ncols, nnz_, dptr, didx, dval, s)) acc = std::move(ac);
    SortedMatrix so;
    // (the sorted sub-tiles are the fall-back of the accumulated tiles; both are built only when asked for)
    if ((!acc.built || getenv("IPXK_BUILD_ALL_LAYOUTS")) && device_build_sorted(S, so, sliced, nrows, ncols, nnz_, dptr, didx, dval, s)) sorted = std::move(so);
    // sliced against sorted: bit-identical partial sums, the faster one is kept (as in build())
    DevBuf<double> tx((size_t)std::max(ncols, 1)), tout((size_t)std::max(nrows, 1));
    IPXK_HIP(hipMemsetAsync(tx.get(), 0, tx.size() * sizeof(double), s));
    hipEvent_t e0, e1;
    IPXK_HIP(hipEventCreate(&e0));
    IPXK_HIP(hipEventCreate(&e1));
    EpiScale epi{{}, nullptr, tout.get()};
    auto time_current = [&]() {
        const int reps = 5;
        for (int w = 0; w < 2; w++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) launch_spmv(*this, tx.get(), epi, nullptr, nullptr, s);
        IPXK_HIP(hipEventRecord(e1, s));
        IPXK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    tuned_us_sliced = time_current();
    if (sorted.built) {
        use_sorted = true;
        tuned_us_sorted = time_current();
        use_sorted = tuned_us_sorted < tuned_us_sliced || (getenv("IPXK_SPMV_SORTED") && getenv("IPXK_SPMV_SORTED")[0] == '1');
        if (!use_sorted && !getenv("IPXK_BUILD_ALL_LAYOUTS")) sorted = SortedMatrix();
    }
    if (acc.built) {
        // used whenever it was built: its row sums equal the sliced layout's bit for bit only for rows stored with
        // ascending indices, so the choice is a property of the matrix and the environment, never of a timing
        use_acc = true;
        tuned_us_acc = time_current();
    }
    IPXK_HIP(hipEventDestroy(e0));
    IPXK_HIP(hipEventDestroy(e1));
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: gather matrix %d x %d nnz %lld built on the device: sliced %.1f us, sorted %.1f us, accumulated %.1f us (%lld batches, %.1f%% of the entries waited) (fullest-slice share %.2f) -> %s\n",
                nrows, ncols, (long long)nnz, tuned_us_sliced, tuned_us_sorted, tuned_us_acc, (long long)acc.nbatches,
                acc.built ? 100.0 * (double)acc.deferred / (double)nnz : 0.0, sliced.dominant_fraction,
                use_acc ? "accumulated" : use_sorted ? "sorted" : "sliced");
    nnz = nnz_all;
    return true;
}

// Sliced layout (internal.hpp).  Eligible when x does not fit an XCD's L2, no row is "long" and
// every tile fits the LDS staging buffer.
void GatherMatrix::build_sliced(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s,
                                int ns_request) {
    const int64_t x_bytes = (int64_t)ncols * 8;
    if (nrows == 0 || nnz == 0 || ncols == 0) return;
    // long rows (dense columns) stay with the long-row kernels, the tiles hold everything else
    const std::vector<unsigned char>& rlong = h_row_long;
    int64_t nshort = nnz;
    for (int r = 0; r < nrows && !rlong.empty(); r++) if (rlong[r]) nshort -= hptr[r + 1] - hptr[r];
    if (nshort == 0) return;
    int ns = 1;
    if (ns_request != 1) {
        int64_t slice_bytes = int64_t(2) << 20;          // half of an XCD's L2
        if (const char* e = getenv("IPXK_SLICE_TEST_KB"))   // tests: make small matrices eligible
            if (atoi(e) > 0) slice_bytes = (int64_t)atoi(e) << 10;
        // x fits an XCD's L2: nothing to slice (IPXK_SLICE_FORCE2: experiments with two slices)
        if (x_bytes <= 2 * slice_bytes && !(getenv("IPXK_SLICE_FORCE2") && x_bytes > slice_bytes)) return;
        ns = 2;
        while (ns < 8 && x_bytes > (int64_t)ns * slice_bytes) ns *= 2;
    }
    const int64_t slice = ((ncols + ns - 1) / ns + 15) / 16 * 16;
    // rows per tile: as many as fit the LDS staging buffer (a matrix whose rows concentrate in one
    // slice, e.g. a banded one, needs smaller tiles than a uniformly random one)
    int R = kSlicedRows, nrb = 0, max_tile = 0;
    // small matrices: enough tiles to give every CU several workgroups (C2, 50k x 100k: 98 tiles of 1024 rows kept
    // 98 of the 256 CUs busy with 8192 entries each)
    while (R > kBlock && (nrows + R - 1) / R * (int64_t)ns < 2048) R /= 2;
    int64_t ntiles = 0;
    std::vector<unsigned> tptr;
    std::vector<unsigned char> cnt;
    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    for (;; R /= 2) {
        if (R < kBlock) {
            if (verbose) fprintf(stderr, "ipxk: sliced layout not used for %d x %d: a tile of %d rows holds %d entries\n", nrows, ncols, 2 * R, max_tile);
            return;
        }
        nrb = (nrows + R - 1) / R;
        ntiles = (int64_t)nrb * ns;
        // pass 1: entries per (tile, row)
        tptr.assign((size_t)ntiles + 1, 0);
        cnt.assign((size_t)ntiles * R, 0);
        for (int r = 0; r < nrows; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            const int64_t tile0 = (int64_t)(r / R) * ns;
            const int rr = r % R;
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
                const int64_t tile = tile0 + hidx[p] / slice;
                unsigned char& cc = cnt[(size_t)tile * R + rr];
                if (cc == 255) {                                // count does not fit a byte
                    if (verbose) fprintf(stderr, "ipxk: sliced layout not used for %d x %d: row %d has > 255 entries in one slice\n", nrows, ncols, r);
                    return;
                }
                cc++;
                tptr[tile + 1]++;
            }
        }
        max_tile = 0;
        for (int64_t t = 0; t < ntiles; t++) {
            max_tile = std::max(max_tile, (int)tptr[t + 1]);
            tptr[t + 1] += tptr[t];
        }
        if ((int64_t)tptr[ntiles] != nshort) return;
        if (max_tile <= kSlicedMaxTile) break;
    }
    // how concentrated the gathers of a row block are: share of the entries in the block's fullest slice
    // (1/ns for uniformly spread indices, ~1 for a banded matrix)
    {
        int64_t dom = 0;
        for (int rb = 0; rb < nrb; rb++) {
            unsigned best = 0;
            for (int sl = 0; sl < ns; sl++) best = std::max(best, tptr[(size_t)rb * ns + sl + 1] - tptr[(size_t)rb * ns + sl]);
            dom += best;
        }
        sliced.dominant_fraction = (double)dom / (double)nshort;
    }
    // pass 2: fill, rows in order, a row's entries in storage order (no assumption that the indices
    // of a row are sorted)
    std::vector<int> ti((size_t)nshort);
    std::vector<double> tv((size_t)nshort);
    std::vector<unsigned> cursor(tptr.begin(), tptr.end() - 1);
    for (int r = 0; r < nrows; r++) {
        if (!rlong.empty() && rlong[r]) continue;
        const int64_t tile0 = (int64_t)(r / R) * ns;
        for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
            const unsigned put = cursor[tile0 + hidx[p] / slice]++;
            ti[put] = (int)hidx[p];
            tv[put] = hval[p];
        }
    }
    sliced.R = R;
    sliced.nslices = ns;
    sliced.nrb = nrb;
    sliced.nrows_pad = nrb * R;
    sliced.max_tile = max_tile;
    sliced.tile_ptr.upload(tptr, s);
    sliced.cnt.upload(cnt, s);
    sliced.idx.upload(ti, s);
    sliced.val.upload(tv, s);
    sliced.partial.resize(ns > 1 ? (size_t)ns * sliced.nrows_pad : 1);
    IPXK_HIP(hipStreamSynchronize(s));
    sliced.built = true;
}

// Sorted sub-tiles (internal.hpp): the slices of the sliced layout, which must exist.
void GatherMatrix::build_sorted(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s) {
    sorted = SortedMatrix();
    if (!sliced.built || sliced.nslices < 2) return;
    const int ns = sliced.nslices;
    const int64_t slice = ((ncols + ns - 1) / ns + 15) / 16 * 16;        // as in build_sliced
    if (slice > (int64_t(1) << kSortedOffBits)) return;
    // (sub-slices per slice: 2.  Round 3, measured at C3 with IPXK_SORTED_NSUB: 4 sub-slices let the row block double at the
    // same staging buffer, i.e. twice the entries per line of the gathered window -- but the apply went 238 -> 276 us, 8 ->
    // 351 us: two more barriers, a scan and a count word per row and sub-tile cost more than the shared requests save)
    static const int nsub_env = [] { const char* e = getenv("IPXK_SORTED_NSUB"); return e && atoi(e) > 0 ? std::min(atoi(e), 16) : 2; }();
    const int nsub = nsub_env;
    const int64_t half = (slice / nsub + 15) / 16 * 16;
    const std::vector<unsigned char>& rlong = h_row_long;
    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    int RB = 32 * kSortedThreads, nrb = 0, max_sub = 0;
    int64_t nsubs = 0, nshort = 0;
    std::vector<unsigned> sptr;
    std::vector<unsigned char> cnt;
    for (;; RB /= 2) {
        if (RB < 4 * kSortedThreads) {
            if (verbose) fprintf(stderr, "ipxk: sorted layout not used for %d x %d: a sub-tile of %d rows holds %d entries\n", nrows, ncols, 2 * RB, max_sub);
            return;
        }
        nrb = (nrows + RB - 1) / RB;
        nsubs = (int64_t)nrb * ns * nsub;
        sptr.assign((size_t)nsubs + 1, 0);
        cnt.assign((size_t)nsubs * RB, 0);
        bool ok = true;
        for (int r = 0; r < nrows && ok; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            const int64_t tile0 = (int64_t)(r / RB) * ns;
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
                const int64_t sl = hidx[p] / slice, off = hidx[p] - sl * slice;
                const int64_t sub = (tile0 + sl) * nsub + std::min<int64_t>(off / half, nsub - 1);
                unsigned char& cc = cnt[(size_t)sub * RB + r % RB];
                if (cc == 255) { ok = false; break; }
                cc++;
                sptr[sub + 1]++;
            }
        }
        if (!ok) return;                 // a row with > 255 entries in one sub-slice
        max_sub = 0;
        for (int64_t t = 0; t < nsubs; t++) { max_sub = std::max(max_sub, (int)sptr[t + 1]); sptr[t + 1] += sptr[t]; }
        nshort = sptr[nsubs];
        if (max_sub <= kSortedMaxSub) break;
    }
    if (nshort == 0) return;
    // entries row by row (slot = place in that order), then every sub-tile sorted by gathered index
    std::vector<unsigned> pk((size_t)nshort);
    std::vector<double> tv((size_t)nshort);
    {
        std::vector<unsigned> cursor(sptr.begin(), sptr.end() - 1);
        for (int r = 0; r < nrows; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            const int64_t tile0 = (int64_t)(r / RB) * ns;
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
                const int64_t sl = hidx[p] / slice, off = hidx[p] - sl * slice;
                const int64_t sub = (tile0 + sl) * nsub + std::min<int64_t>(off / half, nsub - 1);
                const unsigned put = cursor[sub]++;
                pk[put] = ((put - sptr[sub]) << kSortedOffBits) | (unsigned)off;
                tv[put] = hval[p];
            }
        }
        std::vector<std::pair<unsigned, double>> tmp;
        const unsigned mask = (1u << kSortedOffBits) - 1u;
        for (int64_t t = 0; t < nsubs; t++) {
            const unsigned a = sptr[t], b = sptr[t + 1];
            if (b - a < 2) continue;
            tmp.resize(b - a);
            for (unsigned e = a; e < b; e++) tmp[e - a] = {pk[e], tv[e]};
            std::sort(tmp.begin(), tmp.end(), [&](const std::pair<unsigned, double>& x, const std::pair<unsigned, double>& y) {
                const unsigned ox = x.first & mask, oy = y.first & mask;
                return ox != oy ? ox < oy : x.first < y.first;
            });
            for (unsigned e = a; e < b; e++) { pk[e] = tmp[e - a].first; tv[e] = tmp[e - a].second; }
        }
    }
    sorted.nslices = ns; sorted.nsub = nsub; sorted.nrb = nrb; sorted.RB = RB; sorted.nrows_pad = nrb * RB;
    sorted.max_sub = max_sub; sorted.slice_elems = (int)slice;
    sorted.sub_ptr.upload(sptr, s);
    sorted.cnt.upload(cnt, s);
    sorted.pack.upload(pk, s);
    sorted.val.upload(tv, s);
    sorted.partial.resize((size_t)ns * sorted.nrows_pad);
    IPXK_HIP(hipStreamSynchronize(s));
    sorted.built = true;
}

// Accumulated tiles (internal.hpp), host builder: the reference the device builder (layout_device.hip) is tested against.
void GatherMatrix::build_acc(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s) {
    acc = AccMatrix();
    if (!sliced.built || sliced.nslices < 2 || nlong > 0 || nnz == 0) return;
    const int ns = sliced.nslices;
    const int64_t slice = ((ncols + ns - 1) / ns + 15) / 16 * 16;        // as in build_sliced
    if (slice > (int64_t(1) << kSortedOffBits)) return;
    const int RB = acc_rows_per_block(nrows, ns);
    const int nrb = (nrows + RB - 1) / RB;
    const int64_t ntiles = (int64_t)nrb * ns;
    struct E { unsigned off, row; double v; };
    std::vector<unsigned> tptr((size_t)ntiles + 1, 0);
    for (int r = 0; r < nrows; r++)
        for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) tptr[(size_t)(r / RB) * ns + hidx[p] / slice + 1]++;
    for (int64_t t = 0; t < ntiles; t++) tptr[t + 1] += tptr[t];
    std::vector<E> all((size_t)nnz);
    {
        std::vector<unsigned> cursor(tptr.begin(), tptr.end() - 1);
        for (int r = 0; r < nrows; r++)
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
                const int64_t sl = hidx[p] / slice;
                all[cursor[(size_t)(r / RB) * ns + sl]++] = E{(unsigned)(hidx[p] - sl * slice), (unsigned)(r % RB), hval[p]};
            }
    }
    std::vector<unsigned> pk((size_t)nnz), bp, tb((size_t)ntiles + 1, 0);
    std::vector<double> tv((size_t)nnz);
    std::vector<int> stamp((size_t)RB), pend, newpend;
    int64_t deferred = 0;
    for (int64_t t = 0; t < ntiles; t++) {
        E* a = all.data() + tptr[t];
        const int ne = (int)(tptr[t + 1] - tptr[t]);
        tb[t] = (unsigned)bp.size();
        std::stable_sort(a, a + ne, [](const E& x, const E& y) { return x.off < y.off; });
        std::fill(stamp.begin(), stamp.end(), -1);
        pend.clear();
        int cursor = 0, put = 0, batch = 0;
        while (put < ne) {
            bp.push_back(tptr[t] + (unsigned)put);
            newpend.clear();
            int fill = 0;
            auto offer = [&](int i) {
                if (stamp[a[i].row] == batch || fill == kAccBatch) { newpend.push_back(i); deferred++; return; }
                stamp[a[i].row] = batch;
                pk[tptr[t] + put] = (a[i].row << kSortedOffBits) | a[i].off;
                tv[tptr[t] + put] = a[i].v;
                put++; fill++;
            };
            for (int i : pend) offer(i);                                   // whoever waited goes first, in order
            while (fill < kAccBatch && cursor < ne) offer(cursor++);      // then the stream
            pend.swap(newpend);
            batch++;
        }
    }
    tb[ntiles] = (unsigned)bp.size();
    bp.push_back((unsigned)nnz);
    acc.nslices = ns; acc.nrb = nrb; acc.RB = RB; acc.nrows_pad = nrb * RB; acc.slice_elems = (int)slice;
    acc.nbatches = (int64_t)bp.size() - 1; acc.deferred = deferred;
    acc.tile_batch.upload(tb, s);
    acc.bptr.upload(bp, s);
    acc.pack.upload(pk, s);
    acc.val.upload(tv, s);
    acc.partial.resize((size_t)ns * acc.nrows_pad);
    IPXK_HIP(hipStreamSynchronize(s));
    acc.built = true;
}

AccView GatherMatrix::acc_view() const {
    AccView V;
    V.nrows = nrows; V.nrows_pad = acc.nrows_pad; V.nslices = acc.nslices; V.nrb = acc.nrb; V.RB = acc.RB; V.slice_elems = acc.slice_elems;
    V.xmin = nullptr;
    V.tile_batch = acc.tile_batch.get(); V.bptr = acc.bptr.get(); V.pack = acc.pack.get(); V.val = acc.val.get(); V.partial = acc.partial.get();
    return V;
}

// FUSED accumulated tiles (internal.hpp): one slice, the epilogue in the tile kernel.  Only for matrices without long rows whose
// rows are stored with ascending indices and whose row blocks gather from windows of less than 2^18 entries.
void GatherMatrix::build_acc_fused(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s) {
    accf = AccMatrix();
    if (nrows == 0 || nnz == 0 || ncols == 0 || nlong > 0) return;
    for (int r = 0; r < nrows; r++)
        for (ipxint p = hptr[r] + 1; p < hptr[r + 1]; p++)
            if (hidx[p] <= hidx[p - 1]) return;                    // unsorted row: the sum would not be in storage order
    // rows per tile: as many as a batch has entries (a batch takes one entry per row: with fewer rows its batches could not fill,
    // with more the greedy leaves more tail batches), doubled until there are at most kMaxPartials tiles.  Measured on the banded
    // probe, 8-entry rows: 2048 rows 43.9 us per pass (7816 batches for 16 M entries), 4096 rows 46.0 (9998); 16-entry rows: 1024
    // rows 112, 2048 rows 83 -- those keep the sorted fused tiles (58.6), the timing decides.
    int RB = kAccBatch;
    while (((int64_t)nrows + RB - 1) / RB > kMaxPartials) RB *= 2;
    if (RB > kAccMaxRows) return;
    const int nrb = (nrows + RB - 1) / RB;
    struct E { unsigned off, row; double v; };
    std::vector<E> a;
    std::vector<unsigned> pk((size_t)nnz), bp, tb((size_t)nrb + 1, 0);
    std::vector<double> tv((size_t)nnz);
    std::vector<int> xmin((size_t)nrb, 0), stamp((size_t)RB), pend, newpend;
    int64_t deferred = 0;
    unsigned base = 0;
    for (int t = 0; t < nrb; t++) {
        const int r1 = std::min(nrows, (t + 1) * RB);
        ipxint lo = ncols, hi = -1;
        for (int r = t * RB; r < r1; r++)
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) { lo = std::min(lo, hidx[p]); hi = std::max(hi, hidx[p]); }
        tb[t] = (unsigned)bp.size();
        if (hi < 0) continue;
        if (hi - lo >= (ipxint(1) << kSortedOffBits)) return;          // the tile's window of x is too wide: no locality to use
        xmin[t] = (int)lo;
        a.clear();
        for (int r = t * RB; r < r1; r++)
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) a.push_back(E{(unsigned)(hidx[p] - lo), (unsigned)(r - t * RB), hval[p]});
        const int ne = (int)a.size();
        std::stable_sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.off < y.off; });
        std::fill(stamp.begin(), stamp.end(), -1);
        pend.clear();
        int cursor = 0, put = 0, batch = 0;
        while (put < ne) {
            bp.push_back(base + (unsigned)put);
            newpend.clear();
            int fill = 0;
            auto offer = [&](int i) {
                if (stamp[a[i].row] == batch || fill == kAccBatch) { newpend.push_back(i); deferred++; return; }
                stamp[a[i].row] = batch;
                pk[base + put] = (a[i].row << kSortedOffBits) | a[i].off;
                tv[base + put] = a[i].v;
                put++; fill++;
            };
            for (int i : pend) offer(i);
            while (fill < kAccBatch && cursor < ne) offer(cursor++);
            pend.swap(newpend);
            batch++;
        }
        base += (unsigned)ne;
    }
    tb[nrb] = (unsigned)bp.size();
    bp.push_back((unsigned)nnz);
    accf.nslices = 1; accf.nrb = nrb; accf.RB = RB; accf.nrows_pad = nrb * RB; accf.slice_elems = 0; accf.fused = true;
    accf.nbatches = (int64_t)bp.size() - 1; accf.deferred = deferred;
    accf.tile_batch.upload(tb, s);
    accf.bptr.upload(bp, s);
    accf.pack.upload(pk, s);
    accf.val.upload(tv, s);
    accf.xmin.upload(xmin, s);
    IPXK_HIP(hipStreamSynchronize(s));
    accf.built = true;
}

AccView GatherMatrix::acc_fused_view() const {
    AccView V;
    V.nrows = nrows; V.nrows_pad = accf.nrows_pad; V.nslices = 1; V.nrb = accf.nrb; V.RB = accf.RB; V.slice_elems = 0;
    V.xmin = accf.xmin.get();
    V.tile_batch = accf.tile_batch.get(); V.bptr = accf.bptr.get(); V.pack = accf.pack.get(); V.val = accf.val.get(); V.partial = nullptr;
    return V;
}

// FUSED sorted tiles (internal.hpp): one slice, the epilogue in the tile kernel.
void GatherMatrix::build_sorted_fused(const ipxint* hptr, const ipxint* hidx, const double* hval, hipStream_t s) {
    sorted = SortedMatrix();
    if (nrows == 0 || nnz == 0 || ncols == 0) return;
    const std::vector<unsigned char>& rlong = h_row_long;
    int RB = 32 * kSortedThreads, nrb = 0, max_sub = 0;
    int64_t nshort = 0;
    std::vector<unsigned> sptr;
    std::vector<unsigned char> cnt;
    for (;; RB /= 2) {
        if (RB < kSortedThreads) return;
        nrb = (nrows + RB - 1) / RB;
        if (RB > kSortedThreads && nrb < 1024) continue;          // enough tiles to fill the chip
        sptr.assign((size_t)nrb + 1, 0);
        cnt.assign((size_t)nrb * RB, 0);
        bool ok = true;
        for (int r = 0; r < nrows && ok; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            const int64_t len = hptr[r + 1] - hptr[r];
            if (len > 255) { ok = false; break; }
            cnt[(size_t)(r / RB) * RB + r % RB] = (unsigned char)len;
            sptr[r / RB + 1] += (unsigned)len;
        }
        if (!ok) return;
        max_sub = 0;
        for (int t = 0; t < nrb; t++) { max_sub = std::max(max_sub, (int)sptr[t + 1]); sptr[t + 1] += sptr[t]; }
        nshort = sptr[nrb];
        static const int cap = [] { const char* e = getenv("IPXK_SF_MAXSUB"); return e && atoi(e) >= 256 ? std::min(atoi(e), kSortedMaxSub) : kSortedMaxSub; }();
        if (max_sub <= cap) break;
    }
    if (nshort == 0) return;
    std::vector<int> xmin((size_t)nrb, 0);
    std::vector<unsigned> pk((size_t)nshort);
    std::vector<double> tv((size_t)nshort);
    std::vector<std::pair<ipxint, std::pair<unsigned, double>>> tmp;    // (index, (slot, value))
    for (int t = 0; t < nrb; t++) {
        tmp.clear();
        const int r1 = std::min(nrows, (t + 1) * RB);
        ipxint lo = ncols, hi = -1;
        for (int r = t * RB; r < r1; r++) {
            if (!rlong.empty() && rlong[r]) continue;
            for (ipxint p = hptr[r]; p < hptr[r + 1]; p++) {
                tmp.push_back({hidx[p], {(unsigned)tmp.size(), hval[p]}});
                lo = std::min(lo, hidx[p]); hi = std::max(hi, hidx[p]);
            }
        }
        if (tmp.empty()) continue;
        if (hi - lo >= (ipxint(1) << kSortedOffBits)) return;     // the tile's window of x is too wide: no locality to use
        xmin[t] = (int)lo;
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<ipxint, std::pair<unsigned, double>>& a,
                                             const std::pair<ipxint, std::pair<unsigned, double>>& b) {
            return a.first != b.first ? a.first < b.first : a.second.first < b.second.first;
        });
        for (size_t e = 0; e < tmp.size(); e++) {
            pk[sptr[t] + e] = (tmp[e].second.first << kSortedOffBits) | (unsigned)(tmp[e].first - lo);
            tv[sptr[t] + e] = tmp[e].second.second;
        }
    }
    sorted.nslices = 1; sorted.nsub = 1; sorted.nrb = nrb; sorted.RB = RB; sorted.nrows_pad = nrb * RB;
    sorted.max_sub = max_sub; sorted.slice_elems = 0; sorted.fused = true;
    sorted.sub_ptr.upload(sptr, s);
    sorted.cnt.upload(cnt, s);
    sorted.pack.upload(pk, s);
    sorted.val.upload(tv, s);
    sorted.xmin.upload(xmin, s);
    IPXK_HIP(hipStreamSynchronize(s));
    sorted.built = true;
}

SortedView GatherMatrix::sorted_view() const {
    SortedView V;
    V.xmin = sorted.fused ? sorted.xmin.get() : nullptr;
    V.row_long = sorted.fused && nlong > 0 ? row_long.get() : nullptr;
    V.nrows = nrows; V.nrows_pad = sorted.nrows_pad; V.nslices = sorted.nslices; V.nsub = sorted.nsub; V.nrb = sorted.nrb;
    V.RB = sorted.RB; V.slice_elems = sorted.slice_elems;
    V.sub_ptr = sorted.sub_ptr.get(); V.cnt = sorted.cnt.get(); V.pack = sorted.pack.get(); V.val = sorted.val.get();
    V.partial = sorted.partial.get();
    return V;
}

SlicedView GatherMatrix::sliced_view(int which) const {
    SlicedView V;
    V.nrows = nrows; V.nrows_pad = sliced.nrows_pad; V.nslices = sliced.nslices; V.nrb = sliced.nrb; V.R = sliced.R;
    V.tile_ptr = sliced.tile_ptr.get(); V.cnt = sliced.cnt.get();
    V.idx = sliced.idx.get(); V.val = which == 1 ? valM.get() : sliced.val.get(); V.partial = sliced.partial.get();
    if (which == 2) {
        V.tile_ptr = compact.tile_ptr.get(); V.cnt = compact.cnt.get(); V.idx = compact.idx.get(); V.val = compact.val.get();
    }
    V.row_long = nlong > 0 ? row_long.get() : nullptr;
    V.masked = which == 1 ? 1 : 0;
    return V;
}

GatherView GatherMatrix::view(bool use_masked) const {
    GatherView V;
    V.nrows = nrows; V.ncols = ncols;
    V.P = P; V.G = G; V.RT = RT; V.Q = Q; V.RWrows = RWrows;
    V.step_ptr = step_ptr.get(); V.counts = counts.get();
    V.wg_chunk_ptr = wg_chunk_ptr.get(); V.chunk_start = chunk_start.get();
    V.chunk_info = chunk_info.get(); V.chunk_step = chunk_step.get();
    V.idx = idx.get(); V.val = (use_masked && !use_sliced) ? valM.get() : val.get();
    V.row_long = nlong > 0 ? row_long.get() : nullptr;
    V.nseg = nseg; V.seg_p0 = seg_p0.get(); V.seg_p1 = seg_p1.get();
    V.lidx = lidx.get(); V.lval = (use_masked && nlong > 0) ? lvalM.get() : lval.get();
    V.nlong = nlong; V.long_row = long_row.get(); V.long_slot = long_slot.get();
    V.long_partials = long_partials.get();
    V.stamps = stamps.size() ? stamps.get() : nullptr;
    V.masked = use_masked ? 1 : 0;
    return V;
}

// ---------------------------------------------------------------------------
// masked values (GatherMatrix::mask_values)
// ---------------------------------------------------------------------------
// row of every stored entry.  Both layouts store a unit (tile / step) row by row with one count per row:
// a workgroup scans the counts of its unit and labels the entries.
__global__ __launch_bounds__(kBlock) void rowof_sliced_kernel(SlicedView M, int* __restrict__ rowof) {
    __shared__ int wsum[kBlock / 64];
    const int tile = blockIdx.x, rb = tile / M.nslices, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rpt = M.R / kBlock;
    const unsigned char* cb = M.cnt + (size_t)tile * M.R + (size_t)tid * rpt;
    int mine = 0;
    for (int q = 0; q < rpt; q++) mine += cb[q];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int p = (int)M.tile_ptr[tile] + incl - mine;
    for (int w = 0; w < wave; w++) p += wsum[w];
    for (int q = 0; q < rpt; q++) {
        const int r = rb * M.R + tid * rpt + q;
        for (int k = 0; k < cb[q]; k++) rowof[p++] = r;
    }
}
__global__ __launch_bounds__(kBlock) void rowof_phased_kernel(GatherView M, int64_t RW, int* __restrict__ rowof) {
    __shared__ int total;
    // one step per workgroup; its rows in order of the count slots (thread-serial scan in chunks of kBlock slots)
    const int64_t st = blockIdx.x;
    const int w = (int)(st % M.G), q = (int)(st / ((int64_t)M.P * M.G));
    const int64_t row0 = (int64_t)q * M.G * M.RWrows + (int64_t)w * M.RWrows;
    __shared__ int wsum[kBlock / 64];
    int base = M.step_ptr[st];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t l0 = 0; l0 < RW; l0 += kBlock) {
        const int64_t lr = l0 + tid;
        const int mine = lr < RW ? M.counts[(size_t)st * RW + lr] : 0;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int p = base + incl - mine;
        for (int ww = 0; ww < wave; ww++) p += wsum[ww];
        for (int k = 0; k < mine; k++) rowof[p + k] = (int)(row0 + lr);
        if (tid == kBlock - 1) total = p + mine;
        __syncthreads();
        base = total;
        __syncthreads();
    }
}
__global__ void mask_values_kernel(int64_t nz, const double* __restrict__ val, const int* __restrict__ key,
                                   const double* __restrict__ weight, double* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nz; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = weight[key[e]] != 0.0 ? val[e] : 0.0;
}
__global__ void mask_long_rows_kernel(GatherView M, const double* __restrict__ weight, int by_row, double* __restrict__ out) {
    const int l = blockIdx.x;
    const int r = M.long_row[l];
    for (int sgm = M.long_slot[l]; sgm < M.long_slot[l + 1]; sgm++)
        for (int p = M.seg_p0[sgm] + threadIdx.x; p < M.seg_p1[sgm]; p += blockDim.x)
            out[p] = weight[by_row ? r : M.lidx[p]] != 0.0 ? M.lval[p] : 0.0;
}

void GatherMatrix::mask_values(const double* weight, bool by_row, hipStream_t s) {
    const int64_t nz = use_sliced ? (int64_t)sliced.idx.size() : (int64_t)idx.size();
    const int* gidx = use_sliced ? sliced.idx.get() : idx.get();
    const double* gval = use_sliced ? sliced.val.get() : val.get();
    if (by_row && rowof.size() == 0 && nz > 0) {
        rowof.resize((size_t)nz);
        IPXK_HIP(hipMemsetAsync(rowof.get(), 0, (size_t)nz * sizeof(int), s));
        if (use_sliced) {
            const SlicedView V = sliced_view();
            hipLaunchKernelGGL(rowof_sliced_kernel, dim3(V.nrb * V.nslices), dim3(kBlock), 0, s, V, rowof.get());
        } else {
            const GatherView V = view();
            const int64_t nsteps = (int64_t)Q * P * G;
            hipLaunchKernelGGL(rowof_phased_kernel, dim3((unsigned)nsteps), dim3(kBlock), 0, s, V, (int64_t)kBlock * RT, rowof.get());
        }
    }
    valM.ensure((size_t)std::max<int64_t>(nz, 1));
    if (nz > 0)
        hipLaunchKernelGGL(mask_values_kernel, dim3((unsigned)std::min<int64_t>(4096, (nz + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, nz, gval,
                           by_row ? rowof.get() : gidx, weight, valM.get());
    if (nlong > 0) {
        lvalM.ensure(lval.size());
        hipLaunchKernelGGL(mask_long_rows_kernel, dim3(nlong), dim3(kBlock), 0, s, view(), weight, by_row ? 1 : 0, lvalM.get());
    }
    IPXK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// compacted tiles (GatherMatrix::compact_tiles)
// ---------------------------------------------------------------------------
// kept entries per row of the tile and per tile; one workgroup per tile.  wkey[e] addresses the weight of
// entry e (its row in the gather matrix, or its gathered index), rowof[e] its row.
__global__ __launch_bounds__(kBlock) void compact_count_kernel(SlicedView M, const int* __restrict__ rowof,
                                                               const int* __restrict__ wkey, const double* __restrict__ weight,
                                                               unsigned char* __restrict__ cnt_out, unsigned* __restrict__ tile_kept) {
    __shared__ int rowcnt[kSlicedRows];
    __shared__ int wsum[kBlock / 64];
    const int tile = blockIdx.x, rb = tile / M.nslices, tid = threadIdx.x;
    for (int r = tid; r < M.R; r += kBlock) rowcnt[r] = 0;
    __syncthreads();
    const unsigned e0 = M.tile_ptr[tile], e1 = M.tile_ptr[tile + 1];
    int mine = 0;
    for (unsigned e = e0 + tid; e < e1; e += kBlock)
        if (weight[wkey[e]] != 0.0) { atomicAdd(&rowcnt[rowof[e] - rb * M.R], 1); mine++; }
    __syncthreads();
    for (int r = tid; r < M.R; r += kBlock) cnt_out[(size_t)tile * M.R + r] = (unsigned char)rowcnt[r];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < kBlock / 64; w++) t += wsum[w]; tile_kept[tile] = (unsigned)t; }
}
// exclusive prefix sum of n counts by one workgroup (n = # tiles, a few thousand)
__global__ __launch_bounds__(1024) void compact_scan_kernel(int n, const unsigned* __restrict__ in, unsigned* __restrict__ out) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const unsigned v = i < n ? in[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (i < n) out[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}
// the kept entries of a tile, in order, to their new place; one workgroup per tile
__global__ __launch_bounds__(kBlock) void compact_fill_kernel(SlicedView M, const int* __restrict__ wkey, const double* __restrict__ weight,
                                                              const unsigned* __restrict__ new_ptr, int* __restrict__ idx_out,
                                                              double* __restrict__ val_out) {
    __shared__ int wsum[kBlock / 64];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned e0 = M.tile_ptr[tile], e1 = M.tile_ptr[tile + 1];
    unsigned base = new_ptr[tile];
    for (unsigned c0 = e0; c0 < e1; c0 += kBlock) {
        const unsigned e = c0 + tid;
        const bool keep = e < e1 && weight[wkey[e]] != 0.0;
        const unsigned long long b = __ballot(keep);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int wbefore = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) { if (w < wave) wbefore += wsum[w]; total += wsum[w]; }
        if (keep) {
            idx_out[base + wbefore + before] = M.idx[e];
            val_out[base + wbefore + before] = M.val[e];
        }
        base += (unsigned)total;
        __syncthreads();
    }
}

void GatherMatrix::compact_tiles(const double* weight, bool by_row, hipStream_t s) {
    compact.valid = false;
    if (!use_sliced || !sliced.built) return;
    const int64_t nz = (int64_t)sliced.idx.size();
    const SlicedView V = sliced_view(0);
    const int ntiles = V.nrb * V.nslices;
    if (nz == 0 || ntiles == 0) return;
    if (rowof.size() == 0) {
        rowof.resize((size_t)nz);
        IPXK_HIP(hipMemsetAsync(rowof.get(), 0, (size_t)nz * sizeof(int), s));
        hipLaunchKernelGGL(rowof_sliced_kernel, dim3(ntiles), dim3(kBlock), 0, s, V, rowof.get());
    }
    compact.tile_ptr.ensure((size_t)ntiles + 1);
    compact.tile_kept.ensure((size_t)ntiles);
    compact.cnt.ensure((size_t)ntiles * V.R);
    compact.idx.ensure((size_t)nz);
    compact.val.ensure((size_t)nz);
    const int* wkey = by_row ? rowof.get() : sliced.idx.get();
    hipLaunchKernelGGL(compact_count_kernel, dim3(ntiles), dim3(kBlock), 0, s, V, rowof.get(), wkey, weight, compact.cnt.get(),
                       compact.tile_kept.get());
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, ntiles, compact.tile_kept.get(), compact.tile_ptr.get());
    hipLaunchKernelGGL(compact_fill_kernel, dim3(ntiles), dim3(kBlock), 0, s, V, wkey, weight, compact.tile_ptr.get(),
                       compact.idx.get(), compact.val.get());
    IPXK_HIP(hipGetLastError());
    if (nlong > 0) {           // long rows stay with their masked values
        lvalM.ensure(lval.size());
        hipLaunchKernelGGL(mask_long_rows_kernel, dim3(nlong), dim3(kBlock), 0, s, view(), weight, by_row ? 1 : 0, lvalM.get());
        IPXK_HIP(hipGetLastError());
    }
    compact.valid = true;
}

// ---------------------------------------------------------------------------
// model upload
// ---------------------------------------------------------------------------
// Dense-column classification of Model::FindDenseColumns (src/model.cc:34-56): with the column counts in ascending
// order, the first count that exceeds max(40, 10 * its predecessor) is the threshold.  Equal neighbours never
// satisfy that, so it is enough to walk the DISTINCT counts in ascending order (a histogram instead of a sort).
static void find_dense_columns(Context* c) {
    const int64_t n = c->n, m = c->m;
    c->num_dense = 0;
    c->nz_dense = m + 1;
    c->dense_cols.clear();
    if (n < 2) return;
    ipxint maxcnt = 0;
    for (int64_t j = 0; j < n; j++) maxcnt = std::max(maxcnt, c->h_Ap[j + 1] - c->h_Ap[j]);
    std::vector<int64_t> hist((size_t)maxcnt + 1, 0);
    for (int64_t j = 0; j < n; j++) hist[(size_t)(c->h_Ap[j + 1] - c->h_Ap[j])]++;
    ipxint prev = -1;
    int64_t below = 0;                  // # columns with a smaller count
    for (ipxint v = 0; v <= maxcnt; v++) {
        if (hist[(size_t)v] == 0) continue;
        if (prev >= 0 && v > std::max<ipxint>(40, 10 * prev)) {
            c->num_dense = n - below;
            c->nz_dense = v;
            break;
        }
        prev = v;
        below += hist[(size_t)v];
    }
    if (c->num_dense > 1000) {
        c->num_dense = 0;
        c->nz_dense = m + 1;
    }
    for (int64_t j = 0; j < n; j++)
        if (c->h_Ap[j + 1] - c->h_Ap[j] >= c->nz_dense) c->dense_cols.push_back(j);
}

static double ms_since(std::chrono::steady_clock::time_point& t0) {
    const auto t1 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
    return ms;
}

// The model goes to the device as it is (one staged copy of the caller's arrays); validation, the narrowing to 32-bit
// indices, Transpose and the layouts of both gather matrices happen there (layout_device.hip).  Matrices the device
// builders do not cover take the host builders on host copies of the entries -- of the caller's arrays for the CSC, a
// download of the device's row-wise copy for the other gather matrix.
// the time of the two products of NormalMatrix::Apply on a pair of gather matrices (microseconds)
float time_normal_pair(Context* c, GatherMatrix& Ac, GatherMatrix& Ar) {
    hipStream_t s = c->stream;
    const size_t m = (size_t)std::max<int64_t>(c->m, 1), n = (size_t)std::max<int64_t>(c->n, 1);
    DevBuf<double> y(m), t(n), out(m);
    IPXK_HIP(hipMemsetAsync(y.get(), 0, m * sizeof(double), s));
    hipEvent_t e0, e1;
    IPXK_HIP(hipEventCreate(&e0));
    IPXK_HIP(hipEventCreate(&e1));
    EpiScale ep1{{}, nullptr, t.get()}, ep2{{}, nullptr, out.get()};
    const int reps = 5;
    for (int r = 0; r < 2 + reps; r++) {
        if (r == 2) IPXK_HIP(hipEventRecord(e0, s));
        launch_spmv(Ac, y.get(), ep1, nullptr, nullptr, s);
        launch_spmv(Ar, t.get(), ep2, nullptr, nullptr, s);
    }
    IPXK_HIP(hipEventRecord(e1, s));
    IPXK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    IPXK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms * 1e3f / reps;
}

void build_model(Context* c, const ipxint* Ap, const ipxint* Ai, const double* Ax) {
    const int64_t m = c->m, n = c->n;
    auto t0 = std::chrono::steady_clock::now();
    upload_plain_model(c, Ap, Ai, Ax);
    const int64_t nz = c->nnz;
    c->create_ms[0] = ms_since(t0);
    std::unique_ptr<LayoutScratch, void (*)(LayoutScratch*)> S(new_layout_scratch(), free_layout_scratch);
    c->Acols.csr_ptr = c->pl_Ap.get(); c->Acols.csr_idx = c->pl_Ai.get(); c->Acols.csr_val = c->pl_Ax.get();
    c->Arows.csr_ptr = c->pl_Tp.get(); c->Arows.csr_idx = c->pl_Ti.get(); c->Arows.csr_val = c->pl_Tx.get();
    if (!c->Acols.build_device(*S, n, m, nz, c->pl_Ap.get(), c->pl_Ai.get(), c->pl_Ax.get(), c->stream))
        c->Acols.build(n, m, Ap, Ai, Ax, c->stream);
    c->create_ms[1] = ms_since(t0);
    if (!c->Arows.build_device(*S, m, n, nz, c->pl_Tp.get(), c->pl_Ti.get(), c->pl_Tx.get(), c->stream)) {
        ensure_host_model(c, true);
        c->Arows.build(m, n, c->h_ATp.data(), c->h_ATi.data(), c->h_ATx.data(), c->stream);
        c->h_ATp = std::vector<ipxint>(); c->h_ATi = std::vector<ipxint>(); c->h_ATx = std::vector<double>();
    }
    c->create_ms[2] = ms_since(t0);
    S.reset();
    find_dense_columns(c);
    c->tcols.resize(n > 0 ? n : 1);
    prepare_dense_columns(c);
    if (c->num_dense == 0) {                       // (the Sherman-Morrison-Woodbury preconditioner keeps the numbering as given)
        try {
            reorder_model(c);
        } catch (const Error& e) {                 // an optional acceleration: a model is created without it rather than not at all
            if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: reordering given up: %s\n", e.what());
            c->reord = Reordered();
        }
    }
    c->create_ms[3] = ms_since(t0);
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: model %lld x %lld nnz %lld on the device: upload + transpose %.1f ms, A' layouts %.1f ms, A layouts %.1f ms, rest %.1f ms\n",
                (long long)m, (long long)n, (long long)nz, c->create_ms[0], c->create_ms[1], c->create_ms[2], c->create_ms[3]);
}

// ---------------------------------------------------------------------------
// NormalMatrix::_Apply on device vectors
// ---------------------------------------------------------------------------
// W: device pointer to n+m weights.  Dot partials go to part(kPartCdot); *ndot
// receives their count (nullptr: no dot product wanted).
// column partition: lhs holds the cross-rank sum of A_g t_g; add the slack term, form the dot
__global__ __launch_bounds__(kBlock) void normal_finish_kernel(int m, const double* __restrict__ wI,
                                                               const double* __restrict__ y,
                                                               double* __restrict__ lhs, double* dot_partials,
                                                               const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double dotpart = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < m; r += gridDim.x * kBlock) {
        const double v = y[r] * wI[r] + lhs[r];
        lhs[r] = v;
        dotpart += y[r] * v;
    }
    if (dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (threadIdx.x == 0) dot_partials[blockIdx.x] = d;
    }
}

void normal_apply_dev(Context* c, const double* W, const double* rhs, double* lhs, int* ndot,
                      const int* done) {
    const int64_t n = c->n;
    if (comm_cols(c)) {
        // this rank's columns: t_g = W_g .* (A_g' y) is local, lhs = sum over ranks of A_g t_g
        EpiScale e1{{}, W, c->tcols.get()};
        launch_spmv(c->Acols, rhs, e1, nullptr, done, c->stream);
        double* stage = comm_stage(c, (size_t)c->m);
        EpiScale e2{{}, nullptr, stage ? stage : lhs};
        launch_spmv(c->Arows, c->tcols.get(), e2, nullptr, done, c->stream);
        if (stage) comm_allreduce_sum_staged(c, lhs, (size_t)c->m);
        else comm_allreduce_sum(c, lhs, (size_t)c->m);
        const int g = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (c->m + kBlock - 1) / kBlock));
        hipLaunchKernelGGL(normal_finish_kernel, dim3(g), dim3(kBlock), 0, c->stream, (int)c->m, W + n, rhs, lhs,
                           ndot ? c->part(kPartCdot) : nullptr, done);
        if (ndot) *ndot = g;
        return;
    }
    if (c->reord.in_use) {
        // the CR loop of the diag path in the renumbered model (layout_device.hip): rhs, lhs and the weights in the new numbering
        Reordered& R = c->reord;
        EpiScale e1{{}, R.W.get(), R.tcols.get()};
        launch_spmv(R.Acols, rhs, e1, nullptr, done, c->stream);
        EpiNormalRows e2{{}, R.W.get() + n, rhs, lhs};
        const int np = launch_spmv(R.Arows, R.tcols.get(), e2, ndot ? c->part(kPartCdot) : nullptr, done, c->stream);
        if (ndot) *ndot = np;
        return;
    }
    double* stage = comm_rows(c) ? comm_stage(c, (size_t)n) : nullptr;
    EpiScale e1{{}, W, stage ? stage : c->tcols.get()};
    launch_spmv(c->Acols, rhs, e1, nullptr, done, c->stream);
    if (stage) comm_allreduce_sum_staged(c, c->tcols.get(), (size_t)n);
    else if (comm_rows(c)) comm_allreduce_sum(c, c->tcols.get(), (size_t)n);
    EpiNormalRows e2{{}, W + n, rhs, lhs};
    const int np = launch_spmv(c->Arows, c->tcols.get(), e2, ndot ? c->part(kPartCdot) : nullptr,
                               done, c->stream);
    if (ndot) *ndot = np;
}

void debug_single_pass(Context* c, int which, const double* x, double* out) {
    if (which == 1) {
        EpiScale e1{{}, c->W, out};
        launch_spmv(c->Acols, x, e1, nullptr, nullptr, c->stream);
    } else {
        EpiNormalRows e2{{}, c->W + c->n, out, out};   // y := out (only timing matters)
        launch_spmv(c->Arows, x, e2, nullptr, nullptr, c->stream);
    }
}

}  // namespace ipxk
