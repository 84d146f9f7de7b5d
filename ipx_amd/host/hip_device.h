// Shared plumbing of the IPX-side classes: hands out the ipxk_context of a Model (the model's matrix on one GPU)
// and maps ABI return codes onto the exceptions LpSolver::Solve already handles (reference
// src/lp_solver.cc:98-105: std::bad_alloc -> IPX_STATUS_out_of_memory, anything else -> IPX_STATUS_internal_error).
//
// One device model per Model, not per solver object.  The reference constructs its KKT solvers three times per
// solve (src/lp_solver.cc:375, 386, 457) and that costs nothing, because NormalMatrix stores a reference to the
// model and copies no data (src/normal_matrix.h:20-27).  The device equivalent of "store a reference" is to look
// the Model up in a small process-wide registry: the first solver object uploads the matrix and builds its layouts
// (ipxk_create), the later ones find that context again -- the same matrix by content, not by address: a cached
// context is handed out only if the dimensions agree, colptr agrees EXACTLY (the registry keeps a copy: 8(n+1) bytes)
// and two independent 64-bit hashes of rowidx / values agree (one pass over the arrays computes both), so a Model
// that was reloaded in place is never mistaken for its predecessor.  A context serves ONE live solver object at a
// time (it holds that object's W, factors and workspaces); a second object constructed while the first is alive gets
// a context of its own.  Every hand-out of a cached context goes through ipxk_reset_solver_state: nothing the
// previous solver object computed (W, preconditioner, iterate, operator, factors, tightened pivot tolerance, interrupt
// callback) is visible to the next one.  Idle contexts stay cached (at most kMaxIdle, least recently used first out) until
// HipModel::Clear(); they are deliberately not destroyed by a static destructor (the HIP runtime may be gone by then).
#ifndef IPX_HIP_DEVICE_H_
#define IPX_HIP_DEVICE_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "device_glue.h"
#include "ipx_kkt_hip.h"
#include "model.h"

namespace ipx {

inline void HipCheck(int rc) { ipx_hip::Check(rc); }

class HipModel {
public:
    // The context of the n structural columns of model.AI() (the slack identity is implicit) on `device`.
    explicit HipModel(const Model& model, int device = 0) {
        const SparseMatrix& AI = model.AI();
        const Int m = model.rows(), n = model.cols();
        const Int nz = AI.colptr()[n];
        Registry& R = registry();
        const bool cache = !(std::getenv("IPXK_MODEL_CACHE") && std::getenv("IPXK_MODEL_CACHE")[0] == '0');
        std::uint64_t fp[2] = {0, 0};
        if (cache) Fingerprint(m, n, AI.colptr(), AI.rowidx(), AI.values(), fp);
        {
            std::lock_guard<std::mutex> lock(R.mutex);
            for (Entry& e : R.entries)
                if (cache && !e.in_use && e.m == m && e.n == n && e.nz == nz && e.device == device &&
                    e.fingerprint[0] == fp[0] && e.fingerprint[1] == fp[1] &&
                    std::equal(e.colptr.begin(), e.colptr.end(), AI.colptr())) {
                    e.in_use = true;
                    e.last_use = ++R.tick;
                    e.rowidx = AI.rowidx();
                    e.values = AI.values();
                    e.owner = std::this_thread::get_id();
                    ctx_ = e.ctx;
                    R.hits++;
                    break;
                }
        }
        if (ctx_) {
            HipCheck(ipxk_reset_solver_state(ctx_, 0.0));
            return;
        }
        ipxk_context* ctx = nullptr;
        HipCheck(ipxk_create(m, n, AI.colptr(), AI.rowidx(), AI.values(), device, &ctx));
        std::lock_guard<std::mutex> lock(R.mutex);
        R.creations++;
        ctx_ = ctx;
        if (cache) R.entries.push_back(Entry{m, n, nz, device, {fp[0], fp[1]}, std::vector<Int>(AI.colptr(), AI.colptr() + n + 1), ctx, true, ++R.tick,
                                            AI.rowidx(), AI.values(), std::this_thread::get_id()});
        else owned_ = true;
    }
    ~HipModel() {
        if (owned_) { ipxk_destroy(ctx_); return; }
        Registry& R = registry();
        std::vector<ipxk_context*> drop;
        {
            std::lock_guard<std::mutex> lock(R.mutex);
            for (Entry& e : R.entries)
                if (e.ctx == ctx_) { e.in_use = false; e.last_use = ++R.tick; }
            // keep at most kMaxIdle idle contexts
            for (;;) {
                int idle = 0, oldest = -1;
                for (int i = 0; i < (int)R.entries.size(); i++)
                    if (!R.entries[i].in_use) {
                        idle++;
                        if (oldest < 0 || R.entries[i].last_use < R.entries[oldest].last_use) oldest = i;
                    }
                if (idle <= kMaxIdle) break;
                drop.push_back(R.entries[oldest].ctx);
                R.entries.erase(R.entries.begin() + oldest);
            }
        }
        for (ipxk_context* c : drop) ipxk_destroy(c);
    }
    // The context a live solver object of THIS thread holds for the model whose AI() arrays these are (nullptr: none).  For
    // an LU kernel inside the reference's Basis (LuKernelHip): a Basis::Factorize / Basis::Load that runs while KKTSolverBasisHip
    // is at work goes through the solver's own context, where ipxk_lu_factorize finds the factors of the basis Maxvolume has
    // just factorized and hands them out instead of computing them again.  The pointers only SELECT the context; that the
    // matrix handed to ipxk_lu_factorize is the resident one is established there, entry by entry.
    static ipxk_context* InUse(const Int* rowidx, const double* values) {
        Registry& R = registry();
        std::lock_guard<std::mutex> lock(R.mutex);
        for (const Entry& e : R.entries)
            if (e.in_use && e.rowidx == rowidx && e.values == values && e.owner == std::this_thread::get_id())
                return e.ctx;
        return nullptr;
    }
    HipModel(const HipModel&) = delete;
    HipModel& operator=(const HipModel&) = delete;
    ipxk_context* get() const { return ctx_; }

    // Destroys every cached context that no solver object is using.
    static void Clear() {
        Registry& R = registry();
        std::vector<ipxk_context*> drop;
        {
            std::lock_guard<std::mutex> lock(R.mutex);
            for (int i = (int)R.entries.size() - 1; i >= 0; i--)
                if (!R.entries[i].in_use) { drop.push_back(R.entries[i].ctx); R.entries.erase(R.entries.begin() + i); }
        }
        for (ipxk_context* c : drop) ipxk_destroy(c);
    }
    // # ipxk_create calls / # constructions served by a cached context so far in this process
    static long creations() { Registry& R = registry(); std::lock_guard<std::mutex> lock(R.mutex); return R.creations; }
    static long hits() { Registry& R = registry(); std::lock_guard<std::mutex> lock(R.mutex); return R.hits; }

    // Two independent 64-bit hashes of the matrix content from ONE pass over colptr / rowidx / values (four host threads:
    // 256 MB at 1M x 2M in ~10 ms): a multiply-xorshift chain and a rotate-add chain with different constants and seeds.
    static void Fingerprint(Int m, Int n, const Int* Ap, const Int* Ai, const double* Ax, std::uint64_t out[2]) {
        const Int nz = Ap[n];
        struct H { std::uint64_t a, b; };
        auto mix = [](H h, std::uint64_t w) -> H {
            h.a ^= w + 0x9E3779B97F4A7C15ull + (h.a << 6) + (h.a >> 2);
            h.a *= 0xBF58476D1CE4E5B9ull;
            h.b = ((h.b << 27) | (h.b >> 37)) + (w ^ 0xC2B2AE3D27D4EB4Full) * 0x94D049BB133111EBull;
            return h;
        };
        auto fold = [&](H x, H y) -> H { return mix(mix(x, y.a), y.b); };
        auto range = [&](const void* base, std::size_t words, int part, int parts) -> H {
            const std::uint64_t* p = static_cast<const std::uint64_t*>(base);
            const std::size_t b = words * part / parts, e = words * (part + 1) / parts;
            H h0{1, 5}, h1{2, 6}, h2{3, 7}, h3{4, 8};
            std::size_t i = b;
            for (; i + 4 <= e; i += 4) { h0 = mix(h0, p[i]); h1 = mix(h1, p[i + 1]); h2 = mix(h2, p[i + 2]); h3 = mix(h3, p[i + 3]); }
            for (; i < e; i++) h0 = mix(h0, p[i]);
            return fold(fold(h0, h1), fold(h2, h3));
        };
        static_assert(sizeof(Int) == 8 && sizeof(double) == 8, "64-bit words");
        const int parts = nz > (Int(1) << 20) ? 4 : 1;
        std::vector<H> part(parts, H{0, 0});
        auto work = [&](int t) {
            H h = range(Ap, (std::size_t)n + 1, t, parts);
            h = fold(h, range(Ai, (std::size_t)nz, t, parts));
            h = fold(h, range(Ax, (std::size_t)nz, t, parts));
            part[t] = h;
        };
        if (parts == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < parts; t++) th.emplace_back(work, t);
            work(0);
            for (std::thread& t : th) t.join();
        }
        H h = mix(mix(H{0xD6E8FEB86659FD93ull, 0x2545F4914F6CDD1Dull}, (std::uint64_t)m), (std::uint64_t)n);
        for (int t = 0; t < parts; t++) h = fold(h, part[t]);
        out[0] = h.a;
        out[1] = h.b;
    }

private:
    static constexpr int kMaxIdle = 2;
    struct Entry {
        Int m, n, nz;
        int device;
        std::uint64_t fingerprint[2];
        std::vector<Int> colptr;
        ipxk_context* ctx;
        bool in_use;
        long last_use;
        const Int* rowidx;          // Model::AI() arrays of the object that holds the context (valid while in_use)
        const double* values;
        std::thread::id owner;
    };
    struct Registry {
        std::mutex mutex;
        std::vector<Entry> entries;
        long tick = 0, creations = 0, hits = 0;
    };
    static Registry& registry() {
        static Registry* R = new Registry;      // never destroyed: see the header comment
        return *R;
    }
    ipxk_context* ctx_{nullptr};
    bool owned_{false};
};

}  // namespace ipx

#endif  // IPX_HIP_DEVICE_H_
