// TEST INFRASTRUCTURE.  Executes the host-side glue of ipx::KKTSolverBasisHip (ipx_amd/host/device_glue.h:
// HandOffBasis, SolveBasisOnDevice, CrDebugMessage) on plain arrays -- the part of the class that runs after
// the reference's Basis has handed over its arrays, which cannot be reached through the class itself here
// because ipx::Basis needs BASICLU.  Reads flat little-endian arrays from a directory written by
// tests/test_gpu_dropin.py (the golden basis fixture), writes results back.  Plain C++, links only the C ABI.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "device_glue.h"

template <class T>
static std::vector<T> Read(const std::string& dir, const char* name) {
    const std::string path = dir + "/" + name + ".bin";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(bytes / sizeof(T));
    if (bytes && fread(v.data(), 1, bytes, f) != (size_t)bytes) exit(2);
    fclose(f);
    return v;
}
template <class T>
static void Write(const std::string& dir, const char* name, const std::vector<T>& v) {
    FILE* f = fopen((dir + "/" + name + ".bin").c_str(), "wb");
    if (!f) exit(2);
    fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const auto dims = Read<ipxint>(dir, "dims");          // m, n
    const ipxint m = dims[0], n = dims[1];
    const auto Ap = Read<ipxint>(dir, "Ap"), Ai = Read<ipxint>(dir, "Ai");
    const auto Ax = Read<double>(dir, "Ax");
    const auto Lp = Read<ipxint>(dir, "Lp"), Li = Read<ipxint>(dir, "Li"), Up = Read<ipxint>(dir, "Up"), Ui = Read<ipxint>(dir, "Ui");
    const auto Lx = Read<double>(dir, "Lx"), Ux = Read<double>(dir, "Ux");
    const auto rowperm = Read<ipxint>(dir, "rowperm"), colperm = Read<ipxint>(dir, "colperm");
    const auto basis = Read<ipxint>(dir, "basis"), status = Read<ipxint>(dir, "status");
    const auto colscale = Read<double>(dir, "colscale"), colscale2 = Read<double>(dir, "colscale2");
    const auto a = Read<double>(dir, "a"), b = Read<double>(dir, "b");
    const auto tolv = Read<double>(dir, "tol");
    try {
        ipxk_context* ctx = nullptr;
        ipx_hip::Check(ipxk_create(m, n, Ap.data(), Ai.data(), Ax.data(), 0, &ctx));
        ipx_hip::BasisHandoff h{m, n, Lp.data(), Li.data(), Lx.data(), Up.data(), Ui.data(), Ux.data(),
                                rowperm.data(), colperm.data(), basis.data(), status.data(), colscale.data()};
        std::vector<double> x(n + m), y(m);
        std::vector<ipxint> iters;
        // 1. first Factorize: full hand-off, then Solve
        ipx_hip::HandOffBasis(ctx, h, false);
        ipx_hip::SolveOutcome r = ipx_hip::SolveBasisOnDevice(ctx, a.data(), b.data(), tolv[0], -1, x.data(), y.data(), nullptr, nullptr, true);
        Write(dir, "x1", x); Write(dir, "y1", y);
        iters.push_back(r.iter); iters.push_back(r.errflag);
        // 2. next Factorize without basis changes: only the scaling is handed over
        h.colscale = colscale2.data();
        ipx_hip::HandOffBasis(ctx, h, true);
        r = ipx_hip::SolveBasisOnDevice(ctx, a.data(), b.data(), tolv[0], -1, x.data(), y.data(), nullptr, nullptr, true);
        Write(dir, "x2", x); Write(dir, "y2", y);
        iters.push_back(r.iter); iters.push_back(r.errflag);
        // 3. the same state through a full hand-off
        ipx_hip::HandOffBasis(ctx, h, false);
        r = ipx_hip::SolveBasisOnDevice(ctx, a.data(), b.data(), tolv[0], -1, x.data(), y.data(), nullptr, nullptr, true);
        Write(dir, "x3", x); Write(dir, "y3", y);
        iters.push_back(r.iter); iters.push_back(r.errflag);
        // 4. iteration cap: errflag 201 and the reference's Debug(3) line
        r = ipx_hip::SolveBasisOnDevice(ctx, a.data(), b.data(), 1e-300, 2, x.data(), y.data(), nullptr, nullptr, true);
        iters.push_back(r.iter); iters.push_back(r.errflag);
        printf("DEBUG3:%s", r.debug3.c_str());
        Write(dir, "iters", iters);
        ipxk_destroy(ctx);
    } catch (const std::exception& e) {
        fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
    printf("DONE\n");
    return 0;
}
