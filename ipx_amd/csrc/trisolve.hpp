// Shared declarations of the basis-preconditioned operator: level-ordered triangular factors on the
// device (trisolve.hip: kernels, solves, host-side analysis; prepare_device.hip: device-side analysis).
#pragma once

#include <vector>

#include "internal.hpp"

namespace ipxk {

constexpr int kTailWidth = 1024;
constexpr int kShortRow = 8;     // rows up to this many entries are solved by one lane

struct SweepView {
    const int* order;      // [dim] unknown index of level-ordered position k
    const int* ptr;        // [dim+1]
    const int* idx;        // dependency unknown index
    const double* val;
    const double* diag;    // [dim] diagonal (1.0 for unit triangular)
};

struct Sweep {
    int dim = 0, nlevels = 0, npos = 0;   // npos: level-ordered positions incl. padding
    bool running = false;          // forward ('n') sweeps subtract one product at a time
    DevBuf<int> order, ptr, idx;
    DevBuf<double> val, diag;      // as given
    DevBuf<double> valS, diagS;    // column-scaled copy (U sweeps only)
    bool has_scaled = false;
    std::vector<int> level_ptr;    // host, [nlevels+1]
    DevBuf<int> level_ptr_dev;
    // tail: a run of narrow levels in one LDS-resident single-workgroup launch (tslot_off: offset of
    // its dependency-slot table in `tslot`; e0/ne: its entries); otherwise one level with gl lanes
    // per unknown (1, or 8 for long rows)
    struct Launch { int l0, l1; bool tail; int gl; int tslot_off, e0, ne; };
    DevBuf<short> tslot;
    DevBuf<unsigned char> chunk_long;   // sync-free sweep: chunk holds rows longer than kShortRow
    std::vector<Launch> plan;
    SweepView view(bool scaled) const {
        SweepView V;
        V.order = order.get(); V.ptr = ptr.get(); V.idx = idx.get();
        V.val = (scaled && has_scaled) ? valS.get() : val.get();
        V.diag = (scaled && has_scaled) ? diagS.get() : diag.get();
        return V;
    }
};

struct SplitOperator {
    int m = 0;
    Sweep Ut, Lt, Lf, Uf;
    DevBuf<double> Wsplit;                 // n+m: colscale^2 on NONBASIC columns, else 0
    DevBuf<int> rowperm, rowperm_inv, colperm, basis, status;
    DevBuf<double> colscale;
    DevBuf<unsigned char> free_mask;       // m, pivot order
    int num_free = 0;
    DevBuf<double> w0, w1, w2, w3;         // m workspaces
    DevBuf<double> wsf;                    // intermediate vector of a sync-free solve pair
    DevBuf<int> ticket, abort_flag;
    bool syncfree = false;                 // IPXK_TRISOLVE=syncfree selects the single-launch sweeps
    DevBuf<double> tI;                     // m
};

constexpr int kTailSlots = 4096;      // unknowns (level-ordered positions) per LDS tail launch
constexpr int kTailEntries = 7168;    // entries per LDS tail launch
constexpr int kTailLevelsLds = 512;   // levels per LDS tail launch
constexpr int kTailMinLevels = 4;     // shorter runs are cheaper as one launch per level
constexpr int kTailLevelWidth = 2048; // widest level (positions) that may join a run
constexpr int kChunkRows = 256;       // sync-free sweep: positions per ticket

// Launch plan of a sweep from its level structure (host arithmetic, O(#levels)).
//   lptr[l]        first level-ordered position of level l (levels padded to 64 positions)
//   level_long[l]  the level has a row with more than kShortRow entries
//   lev_entry[l]   first entry of level l in the level-ordered entry arrays
// Fills S.plan (with the offsets of the tail runs' dependency-slot tables) and returns the total
// number of slot-table entries.
int plan_sweep(Sweep& S, const std::vector<int>& lptr, const std::vector<unsigned char>& level_long,
               const std::vector<int>& lev_entry);
// chunk flags of the sync-free sweep
std::vector<unsigned char> sweep_chunk_flags(const std::vector<int>& lptr, const std::vector<unsigned char>& level_long);

// Device-side analysis of the four sweeps (prepare_device.hip): uploads L and U as given, builds
// the row lists, computes dependency levels, orders the unknowns and gathers the rows on the GPU.
// uscale: column scaling of U in pivot order (1 where none).
void analyse_sweeps_device(Context* c, SplitOperator* S, const ipxint* Lp, const ipxint* Li, const double* Lx,
                           const ipxint* Up, const ipxint* Ui, const double* Ux,
                           const std::vector<double>& uscale);

}  // namespace ipxk
