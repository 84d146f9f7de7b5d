// Shared declarations of the basis-preconditioned operator: level-ordered triangular factors on the
// device (trisolve.hip: sweep kernels, solves; prepare_device.hip: device-side analysis and packing).
#pragma once

#include <vector>

#include "internal.hpp"

namespace ipxk {

constexpr int kShortRow = 8;     // rows up to this many entries are solved by one lane
constexpr int kLongLanes = 8;    // lanes per unknown for longer rows
constexpr int kLenKeyBits = 8;   // sort key = level << kLenKeyBits | (255 - min(len, 255)): inside a level the
                                 // rows come in descending length, the long ones (> kShortRow) first

// One chunk = the work of one wavefront; inside a level the rows are sorted by length, so the rows of
// a chunk have (nearly) the same length and a chunk is a small dense block:
//   width >= 0: 64 consecutive level-ordered positions, one lane each; entry e of lane t lives at
//               ent0 + e*64 + t, e < width = longest row of the chunk (<= kShortRow)
//   width <  0: 8 consecutive positions with 8 lanes each ("long rows"), -width steps of 8 entries per
//               row; in step s lane (row q, g) holds entry 8*s + g of its row at ent0 + s*64 + 8*q + g
// Slots beyond a row's length hold zeros and are never used (lanes stop at their length).  Either way a
// wavefront reads its chunk with fully coalesced loads whose addresses depend on the descriptor only.
// MERGED chunks: consecutive tiny levels that fit one chunk together (all their rows short and <= 64 in total,
// or all long and <= 8 in total) share a chunk; `sub` = number of levels in it (1 for an ordinary chunk).  The
// wavefront then runs the levels one after the other and passes values between them through lane shuffles
// instead of memory -- a chain of tiny levels costs ~0.15 us per level instead of one memory hand-off each.
// The per-position sub-level is kept in the upper half of `len` (kLenBits).
struct ChunkDesc { int pos0, ent0, width, npos, sub, pad0, pad1, pad2; };
constexpr int kLenBits = 24;            // len word of a position: entries | sub-level << kLenBits (a row of a factor with spikes
                                        // can hold a large part of the dimension: 16 bits were too few)
constexpr int kMaxSubLevels = 32;

// A sweep reads its right-hand side through `src` (position -> index into the input vector, -1 for
// padding) and writes its result BY POSITION: y[k] = value of the unknown at level-ordered position k.
// A wavefront's 64 results are then one contiguous 512-byte store, and all values of a cache line
// become final together; the polls of a wavefront's 64 lanes fall on few lines.  Dependencies are positions
// of the same result vector.  (Measured at C3: the backward pair 318 -> 259 us against results stored by
// unknown.  Reading long-finished dependencies with ordinary cacheable loads instead of L1-bypassing ones was
// tried on top of this and changed nothing.)
struct SweepView {
    const ChunkDesc* chunks;
    const int* src;              // [npos] index into the input vector, -1 for padding
    const double* diag;          // [npos] divisor (1.0 for unit triangular and padding)
    const int* len;              // [npos] entries of the row
    const int* idx;              // dependency position
    const double* val;
    // rows of more than 64 entries are summed in rounds of 64; newest_first: the row's FIRST entries are the
    // dependencies that become available last (the L' sweep: descending unknowns, ascending storage order), so
    // its rounds are taken from the end of the row -- all but the last round then run ahead of the chain
    int newest_first;
    // optional second copy of the result in another order (plain stores, read after the launch): out2[dst2[k]] =
    // value at position k for dst2[k] >= 0 -- the L' sweep of the split operator hands inverse(B~') rhs to the
    // N N' product in the row order of A this way, without a permutation pass of its own
    const int* dst2;
    double* out2;
};

// A sweep is a sequence of launches, each a run of consecutive levels (= a range of chunks):
//   all-XCD run: every workgroup takes part, results are stored write-through (visible chip-wide)
//   one-XCD run: for runs of narrow levels; only the workgroups of ONE XCD take part, so that the
//                hand-off of a value from its producer to its consumers goes through that XCD's L2
struct Sweep {
    int dim = 0, nlevels = 0, npos = 0, nchunks = 0;
    int64_t nentries = 0;          // entry slots (padding included)
    bool running = false;          // forward ('n') sweeps subtract one product at a time
    bool newest_first = false;     // see SweepView
    int scale_mode = 0;            // 0: no scaled copy; 1: column scale of the unknown itself (U');
                                   // 2: column scale of the dependency (U)
    DevBuf<ChunkDesc> chunks;
    DevBuf<int> order;             // [npos] unknown at a position (-1: padding)
    DevBuf<int> posof;             // [dim]  position of an unknown
    DevBuf<int> src;               // [npos] right-hand-side index (composed with the producer's layout, trisolve.hip)
    DevBuf<int> idx, len;
    DevBuf<double> y;              // [npos] result of the sweep, by position
    DevBuf<double> val, diag;      // as given
    DevBuf<double> valS, diagS;    // column-scaled copy (U sweeps only)
    std::vector<int> level_chunk;  // host, [nlevels+1] first chunk of each level
    std::vector<int> level_width;  // host, [nlevels] unknowns per level
    enum Kind { kAllXcds = 0, kOneXcd = 1 };
    struct Launch { int c0, c1; int kind; bool merged; };   // merged: the run contains merged chunks
    std::vector<int> merged_prefix;  // host, [nchunks+1] number of merged chunks before chunk c
    std::vector<Launch> plan;
    // INVERTED BLOCKS (large factors; build_sweep_blocks, trisolve.hip).  The first levels of the transposed sweeps
    // and the last levels of every sweep hold few unknowns each -- the planted C3 factors: 36 levels for the first
    // 6384 unknowns of U', 37 levels for the last 1059 of L, whose rows are the longest of the matrix -- and cost one
    // hand-off each however little work they carry.  With the unknowns of such a run of levels as block 2 of
    // T = [T11 0; T21 T22] (block 1: the levels before it, empty for a head),  x2 = inverse(T22) (b2 - T21 x1):  one
    // kernel that subtracts the rows' outside entries (all known: the launches of the earlier levels are over) and
    // one dense lower-triangular matrix-vector product with M = inverse(T22), computed once per Prepare from the
    // UNSCALED values; the column scaling of the U sweeps is applied around it (mode 1, S T: z / s; mode 2, T S: the
    // result / s).  Same solve in exact arithmetic; the block's rows are no longer summed in the reference's order
    // (1e-13 against the level-scheduled form, tests/test_gpu_parity.py).
    struct Block {
        int K = 0;                 // unknowns of the block (0: none, the plan covers these levels)
        int la = 0, lb = 0;        // its levels [la, lb)
        int p0 = 0, p1 = 0;        // its positions [p0, p1)
        int nh = 0;                // outside entries of its rows
        DevBuf<int> pos;           // [K] position of block unknown t (block order = position order)
        DevBuf<int> unk;           // [K] its unknown
        DevBuf<int> hptr, hslot, hidx;   // [K+1], [nh], [nh] outside entries of a row: slots of the packed entry arrays, positions
        DevBuf<int> zsrc;          // [K] where the sweep's input vector holds the right-hand side of unknown t (Sweep::src of its position)
        DevBuf<double> M, z;       // [K*K] row major; [K]
        DevBuf<int> w_base, w_rank, w_hcnt, w_tcnt, w_lev, w_tptr, w_tcol;   // workspaces of build_block, kept (grow-only)
        DevBuf<double> w_tval, w_probe;
    } head, tail;
    SweepView view(bool scaled) const {
        SweepView V;
        V.chunks = chunks.get(); V.src = src.get(); V.len = len.get(); V.idx = idx.get();
        V.val = (scaled && scale_mode) ? valS.get() : val.get();
        V.diag = (scaled && scale_mode) ? diagS.get() : diag.get();
        V.newest_first = newest_first ? 1 : 0;
        V.dst2 = nullptr; V.out2 = nullptr;
        return V;
    }
};

struct SplitOperator {
    int m = 0;
    Sweep Ut, Lt, Lf, Uf;
    DevBuf<double> Wsplit;                 // n+m: colscale^2 on NONBASIC columns, else 0
    DevBuf<int> rowperm, rowperm_inv, colperm, basis, status;
    DevBuf<ipxint> status_raw;             // n+m, as handed over
    DevBuf<int> counters;                  // scratch flags / counts of the scaling kernels
    DevBuf<double> colscale;
    DevBuf<double> uscale;                 // m, pivot order: column scaling of U (1 where none)
    DevBuf<unsigned char> free_mask;       // m, pivot order
    int num_free = 0;
    DevBuf<double> w0, w1, w2, w3;         // m workspaces
    DevBuf<int> perm_after_backward;       // u[i] = Lt.y[perm_after_backward[i]] is inverse(B~') rhs in row order of A
    DevBuf<int> row_after_backward;        // its inverse by position of the L' sweep: row of A of a position, -1 for padding
    DevBuf<unsigned long long> xcc_slots;  // one-XCD runs: placement consensus words
    unsigned epoch = 0;                    // launch counter of the one-XCD runs
    int sweep_grid_all = 0;                // workgroups of an all-XCD run: all resident on this operator's device (0: not asked yet)
    DevBuf<int> abort_flag;
    DevBuf<double> tI;                     // m
    DevBuf<double> eta_t, eta_in;          // m each: vectors by basis position / pivot order around the eta file (Context::etas_live)
    bool level_launches = false;           // IPXK_TRISOLVE=levels: one launch per level (debugging aid)
    bool masked_values = false;            // N N' uses the gather matrices' masked value arrays (spmv.hip)
    bool real_N = false;                   // N N' uses N built as a matrix of its own (nmatrix.hip)
    // Dense bump of the factorization (factors that came from the device LU): with the bump's block D22 =
    // (L22+I) U22 cut out of L and U,  (L+I) U = (L~+I) blockdiag(I, D22) U~,  L~ = L without L22, U~ = U with U22
    // replaced by I.  The level-scheduled sweeps run on L~ and U~ (no chain as long as the bump), and BETWEEN the
    // two sweeps of a pair one workgroup solves with the dense block in place (bump_solve_kernel, trisolve.hip).
    int bump_start = 0, bump_size = 0;     // 0: no dense block
    DevBuf<double> bumpD;                  // bump_size^2, column major: U22 on and above the diagonal, L22 below
    DevBuf<double> bump_invL, bump_invU;   // inverted 64 x 64 diagonal blocks of L22+I and of U22
    // large blocks: inverse(D22) itself, row major, and its transpose (bump_size^2 each; empty for small blocks) --
    // the solve between the sweeps of a pair is then ONE matrix-vector product spread over the chip
    DevBuf<double> bump_inv, bump_invT, bump_x;
    DevBuf<double> bump_gx;                // blocks too large for LDS: the unknowns of the one-workgroup blocked solve (2 x bump_size)
    bool bump_explicit = false;
    DevBuf<double> bump_probe;             // workspace of the guard of the explicit inverse
    DevBuf<int> bump_pos_fwd, bump_pos_bwd;   // position of bump unknown t in the result of the L sweep / of the U' sweep
    // workspaces of Prepare kept from one call to the next (grow-only): the factors as uploaded, the factors with the
    // dense block cut out
    DevBuf<ipxint> in_Lp, in_Li, in_Up, in_Ui, cut_Lp, cut_Up, cut_Ui;
    DevBuf<double> in_Lx, in_Ux, cut_Ux;
    DevBuf<int> cut_cnt, cut_start;
};

// Launch plan of a sweep from its level structure (host arithmetic, O(#levels)).
void plan_sweep(Sweep& S, bool level_launches);
// decides whether the sweep gets an inverted head / tail and builds them from the packed rows (before plan_sweep)
void build_sweep_blocks(Context* c, Sweep& S, bool level_launches);

// Device-side analysis of the four sweeps (prepare_device.hip): uploads L and U as given, builds
// the row lists, computes dependency levels, orders the unknowns and packs the rows on the GPU.
// L and U in the reference's form (64-bit indices, column-wise, U's diagonal last) resident on the device
struct DeviceFactors {
    const ipxint *Lp, *Li, *Up, *Ui;
    const double *Lx, *Ux;
    int64_t nzL, nzU;
};
// hL*/hU*: host copies of the index arrays if the caller has them (else NULL: fetched only if the fall-back needs them)
void analyse_sweeps_resident(Context* c, SplitOperator* S, const DeviceFactors& F, const ipxint* hLp, const ipxint* hLi,
                             const ipxint* hUp, const ipxint* hUi);
// the factors of the last LU factorization of this context (lu.hip); false if there is none
struct LuView {
    int dim = 0, ndep = 0;
    int bump_start = 0, bump_size = 0;     // the last pivots (stages) of the factorization came from a dense bump
    bool from_basis = false;
    DeviceFactors F{};
    const ipxint *rowperm = nullptr, *colperm = nullptr, *basis = nullptr;
};
bool lu_view(const Context* c, LuView* out);
void lu_plain_matrix(const Context* c, const int** Ap, const int** Ai, const double** Ax);
void analyse_sweeps_device(Context* c, SplitOperator* S, const ipxint* Lp, const ipxint* Li, const double* Lx,
                           const ipxint* Up, const ipxint* Ui, const double* Ux);
// (re)computes the column-scaled value sets of the U sweeps from S->uscale
void rescale_sweeps_device(Context* c, SplitOperator* S);

}  // namespace ipxk
