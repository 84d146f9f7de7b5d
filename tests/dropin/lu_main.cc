// TEST INFRASTRUCTURE.  ipx::LuKernelHip (ipx_amd/host/lu_kernel_hip.*, this repo, MI355X) under the reference's
// own LU machinery: the reference's ForrestTomlin (src/forrest_tomlin.cc, built from the reference's sources into
// oracle/_ref) owns it as its LuFactorization and
//   * factorizes (return code, stability() of LuFactorization::Factorize, fill factor),
//   * solves B x = b and B'y = b (SolveDense),
//   * replaces basis columns: FtranForUpdate / BtranForUpdate / Update, then solves with the updated basis.
// Input: a directory with dims.bin (dim, #replacements), Bp/Bi/Bx.bin, and per replacement k: pos_k.bin (position),
// ci_k.bin / cx_k.bin (the entering column).  Built by `make -C oracle lu_dropin`; run by tests/test_gpu_dropin.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "control.h"
#include "forrest_tomlin.h"
#include "indexed_vector.h"
#include "lu_kernel_hip.h"

using ipx::Int;
using ipx::Vector;

// hands back factors computed earlier: stands in for the CPU kernel (BasicLuKernel) that LuKernelHip delegates to
class Replay : public ipx::LuFactorization {
public:
    ipx::SparseMatrix L, U;
    std::vector<Int> rowperm, colperm, dependent;
    int* calls;
private:
    void _Factorize(Int, const Int*, const Int*, const Int*, const double*, double, bool, ipx::SparseMatrix* Lout,
                    ipx::SparseMatrix* Uout, std::vector<Int>* rp, std::vector<Int>* cp, std::vector<Int>* dep) override {
        *Lout = L; *Uout = U; *rp = rowperm; *cp = colperm; *dep = dependent;
        (*calls)++;
    }
};

template <class T>
static std::vector<T> ReadBin(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<T> v(bytes / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), bytes);
    return v;
}

// max |B x - b| / (1 + max|x|) for the CURRENT basis columns (trans: B'x - b)
static double Residual(Int dim, const std::vector<std::vector<Int>>& ci, const std::vector<std::vector<double>>& cx,
                       const Vector& x, const Vector& b, bool trans) {
    Vector r(dim);
    for (Int i = 0; i < dim; i++) r[i] = -b[i];
    for (Int j = 0; j < dim; j++)
        for (size_t p = 0; p < ci[j].size(); p++) {
            if (trans) r[j] += cx[j][p] * x[ci[j][p]];
            else r[ci[j][p]] += cx[j][p] * x[j];
        }
    double num = 0.0, den = 0.0;
    for (Int i = 0; i < dim; i++) { num = std::max(num, std::abs(r[i])); den = std::max(den, std::abs(x[i])); }
    return num / (1.0 + den);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string d = argv[1];
    const std::vector<Int> dims = ReadBin<Int>(d + "/dims.bin");
    const Int dim = dims[0], nrep = dims[1];
    const std::vector<Int> Bp = ReadBin<Int>(d + "/Bp.bin"), Bi = ReadBin<Int>(d + "/Bi.bin");
    const std::vector<double> Bx = ReadBin<double>(d + "/Bx.bin");
    std::vector<std::vector<Int>> ci(dim);
    std::vector<std::vector<double>> cx(dim);
    for (Int j = 0; j < dim; j++) {
        ci[j].assign(Bi.begin() + Bp[j], Bi.begin() + Bp[j + 1]);
        cx[j].assign(Bx.begin() + Bp[j], Bx.begin() + Bp[j + 1]);
    }
    // a context only names the device here
    const ipxint one_p[2] = {0, 1}, one_i[1] = {0};
    const double one_x[1] = {1.0};
    ipxk_context* ctx = nullptr;
    if (ipxk_create(1, 1, one_p, one_i, one_x, 0, &ctx) != 0) { std::fprintf(stderr, "%s\n", ipxk_last_error()); return 1; }

    ipx::Control control;
    ipx::Parameters prm;
    prm.display = 0;
    control.parameters(prm);
    ipx::LuKernelHip* kernel = new ipx::LuKernelHip(ctx);
    std::unique_ptr<ipx::LuFactorization> lu(kernel);
    ipx::ForrestTomlin ft(control, dim, lu);
    ft.pivottol(0.1);
    const Int flag = ft.Factorize(Bp.data(), Bp.data() + 1, Bi.data(), Bx.data(), false);
    std::printf("factorize: flag %ld stability %.3e fill %.3f singletons %ld+%ld bump %ld rounds %ld\n", (long)flag,
                kernel->stability(), ft.fill_factor(), (long)kernel->info().col_singletons,
                (long)kernel->info().row_singletons, (long)kernel->info().bump, (long)kernel->info().rounds);
    int fails = 0;
    if (flag != 0 || !(kernel->stability() < 1e-12)) { std::printf("FAIL factorize\n"); fails++; } else std::printf("PASS factorize\n");

    Vector b(dim), x(dim);
    for (Int i = 0; i < dim; i++) b[i] = std::sin(0.37 * (double)i) + 0.25;
    for (int trans = 0; trans < 2; trans++) {
        ft.SolveDense(b, x, trans ? 'T' : 'N');
        const double r = Residual(dim, ci, cx, x, b, trans != 0);
        std::printf("solve %c: residual %.3e\n", trans ? 'T' : 'N', r);
        if (!(r < 1e-9)) { std::printf("FAIL solve\n"); fails++; } else std::printf("PASS solve\n");
    }
    // column replacements through the reference's Forrest-Tomlin update
    Int done = 0;
    for (Int k = 0; k < nrep; k++) {
        const std::string tag = std::to_string(k);
        const Int pos = ReadBin<Int>(d + "/pos_" + tag + ".bin")[0];
        const std::vector<Int> ni = ReadBin<Int>(d + "/ci_" + tag + ".bin");
        const std::vector<double> nx = ReadBin<double>(d + "/cx_" + tag + ".bin");
        ipx::IndexedVector ftran(dim), btran(dim);
        ft.FtranForUpdate((Int)ni.size(), ni.data(), nx.data(), ftran);
        ft.BtranForUpdate(pos, btran);
        const double pivot = ftran[pos];
        if (std::abs(pivot) < 1e-3) continue;              // would make the basis (nearly) singular: skip
        const Int err = ft.Update(pivot);
        if (err < 0) { std::printf("update %ld: singular (%ld)\n", (long)k, (long)err); continue; }
        ci[pos] = ni;
        cx[pos] = nx;
        done++;
    }
    std::printf("updates: %ld of %ld\n", (long)done, (long)nrep);
    for (int trans = 0; trans < 2; trans++) {
        ft.SolveDense(b, x, trans ? 'T' : 'N');
        const double r = Residual(dim, ci, cx, x, b, trans != 0);
        std::printf("solve %c after updates: residual %.3e\n", trans ? 'T' : 'N', r);
        if (!(r < 1e-8) || done == 0) { std::printf("FAIL updated solve\n"); fails++; } else std::printf("PASS updated solve\n");
    }
    // a basis the device LU declines goes to the fallback kernel given to the constructor
    {
        ipx::LuKernelHip direct(ctx);
        Replay* rep = new Replay;
        int calls = 0;
        rep->calls = &calls;
        direct.Factorize(dim, Bp.data(), Bp.data() + 1, Bi.data(), Bx.data(), 0.1, false, &rep->L, &rep->U, &rep->rowperm,
                         &rep->colperm, &rep->dependent);
        setenv("IPXK_LU_BUMP_MAX", "1", 1);               // now every bump is "too large" for the device ...
        setenv("IPXK_LU_SPARSE", "0", 1);                 // ... and the sparse elimination rounds may not take it either
        ipx::LuKernelHip with_fallback(ctx, std::unique_ptr<ipx::LuFactorization>(rep));
        ipx::SparseMatrix L2, U2;
        std::vector<Int> rp2, cp2, dep2;
        with_fallback.Factorize(dim, Bp.data(), Bp.data() + 1, Bi.data(), Bx.data(), 0.1, false, &L2, &U2, &rp2, &cp2, &dep2);
        bool threw = false;
        try {
            ipx::LuKernelHip none(ctx);
            none.Factorize(dim, Bp.data(), Bp.data() + 1, Bi.data(), Bx.data(), 0.1, false, &L2, &U2, &rp2, &cp2, &dep2);
        } catch (const std::exception& e) { threw = true; }
        unsetenv("IPXK_LU_BUMP_MAX");
        unsetenv("IPXK_LU_SPARSE");
        const bool ok = calls == 1 && with_fallback.fallbacks() == 1 && with_fallback.stability() < 1e-12 && threw &&
                        direct.info().bump > 1;
        std::printf("fallback: calls %d, counted %ld, stability %.2e, without a fallback it throws: %d\n", calls,
                    (long)with_fallback.fallbacks(), with_fallback.stability(), (int)threw);
        if (!ok) { std::printf("FAIL fallback\n"); fails++; } else std::printf("PASS fallback\n");
    }
    ipxk_destroy(ctx);
    std::printf(fails ? "FAILED\n" : "DONE\n");
    return fails ? 1 : 0;
}
