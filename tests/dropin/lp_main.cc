// TEST INFRASTRUCTURE.  The reference's whole LpSolver (src/lp_solver.cc: presolve, starting point, initial IPM,
// starting basis, main IPM with basis preconditioning, crossover) run on an LP read from a directory.
// `make -C oracle lp_dropin` links this file twice:
//   oracle/_ref/test_lp_ref -- with the reference's lp_solver.cc as it is: ipx::KKTSolverDiag / ipx::KKTSolverBasis
//   oracle/_ref/test_lp_hip -- with the three declarations of src/lp_solver.cc:375,386,457 changed as INTEGRATION.md
//                              says (the change is applied by `sed` on the way into the compiler, nothing is stored):
//                              ipx::KKTSolverDiagHip / ipx::KKTSolverBasisHip (this repo, MI355X) under the
//                              reference's IPM::ComputeStartingPoint / IPM::Driver
// Both need ipx::Basis, hence tests/dropin/basiclu_absent.cc (every LU factorization goes to ipx::LuKernelHip; run
// with lu_kernel 1).  tests/test_gpu_lp_dropin.py runs the two programs on the same model and compares.
//
// Input directory: dims.bin (num_var, num_constr), obj/lb/ub/rhs/Ax.bin (double), Ap/Ai.bin (int64),
// constr_type.bin (chars), params.txt ("name value" lines for the ipx_parameters fields used below).
// Output: info.txt ("name value" lines of ipx_info + harness counters), x/y/slack/zl/zu.bin of the interior
// solution when one exists, and the basic solution's x/vbasis/cbasis when crossover produced one.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "lp_solver.h"

extern "C" long dropin_lu_factorizations();
extern "C" long dropin_lu_reused();
extern "C" long dropin_lu_max_bump();
extern "C" double dropin_lu_seconds();
#ifdef IPX_LP_HIP
#include "hip_device.h"
extern "C" double ipx_hip_cpu_prepare_seconds();
extern "C" long ipx_hip_cpu_prepare_calls();
extern "C" long ipx_hip_device_maxvolume_calls();
extern "C" long ipx_hip_cpu_maxvolume_calls();
extern "C" long ipx_hip_kept_eta_calls();
extern "C" long ipx_hip_kept_etas();
extern "C" double ipx_hip_factorize_phase_seconds(int phase);
#endif

template <class T>
static std::vector<T> ReadBin(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<T> v(bytes / sizeof(T));
    if (bytes) f.read(reinterpret_cast<char*>(v.data()), bytes);
    return v;
}
template <class T>
static void WriteBin(const std::string& path, const T* p, size_t count) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), count * sizeof(T));
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <input dir> <output dir>\n", argv[0]); return 2; }
    const std::string in = argv[1], out = argv[2];
    const std::vector<ipxint> dims = ReadBin<ipxint>(in + "/dims.bin");
    const ipxint num_var = dims[0], num_constr = dims[1];
    const std::vector<double> obj = ReadBin<double>(in + "/obj.bin"), lb = ReadBin<double>(in + "/lb.bin"),
                              ub = ReadBin<double>(in + "/ub.bin"), rhs = ReadBin<double>(in + "/rhs.bin"),
                              Ax = ReadBin<double>(in + "/Ax.bin");
    const std::vector<ipxint> Ap = ReadBin<ipxint>(in + "/Ap.bin"), Ai = ReadBin<ipxint>(in + "/Ai.bin");
    const std::vector<char> ct = ReadBin<char>(in + "/constr_type.bin");

    ipx::LpSolver solver;
    ipx::Parameters p = solver.GetParameters();
    p.display = 0;
    p.lu_kernel = 1;                 // ForrestTomlin over an LuFactorization (src/basis.cc:26-28)
    {
        std::ifstream f(in + "/params.txt");
        std::string name;
        double v;
        while (f >> name >> v) {
            if (name == "display") p.display = (ipxint)v;
            else if (name == "debug") p.debug = (ipxint)v;
            else if (name == "dualize") p.dualize = (ipxint)v;
            else if (name == "switchiter") p.switchiter = (ipxint)v;
            else if (name == "crossover") p.crossover = (ipxint)v;
            else if (name == "crash_basis") p.crash_basis = (ipxint)v;
            else if (name == "update_heuristic") p.update_heuristic = (ipxint)v;
            else if (name == "ipm_maxiter") p.ipm_maxiter = (ipxint)v;
            else if (name == "stop_at_switch") p.stop_at_switch = (ipxint)v;
            else if (name == "ipm_feasibility_tol") p.ipm_feasibility_tol = v;
            else if (name == "ipm_optimality_tol") p.ipm_optimality_tol = v;
            else if (name == "ipm_drop_primal") p.ipm_drop_primal = v;
            else if (name == "ipm_drop_dual") p.ipm_drop_dual = v;
            else if (name == "time_limit") p.time_limit = v;
            else { std::fprintf(stderr, "unknown parameter %s\n", name.c_str()); return 2; }
        }
    }
    solver.SetParameters(p);
    const ipxint load_err = solver.LoadModel(num_var, obj.data(), lb.data(), ub.data(), num_constr, Ap.data(),
                                             Ai.data(), Ax.data(), rhs.data(), ct.data());
    if (load_err) { std::printf("LoadModel errflag %ld\n", (long)load_err); return 3; }
    const ipxint status = solver.Solve();
    const ipx::Info info = solver.GetInfo();

    std::ofstream f(out + "/info.txt");
    f.precision(17);
#define PUT(name) f << #name << ' ' << info.name << '\n'
    f << "solve_status " << status << '\n';
    PUT(status); PUT(status_ipm); PUT(status_crossover); PUT(errflag);
    PUT(num_rows_solver); PUT(num_cols_solver); PUT(num_entries_solver); PUT(dualized); PUT(dense_cols);
    PUT(dependent_rows); PUT(dependent_cols); PUT(rows_inconsistent); PUT(cols_inconsistent);
    PUT(primal_dropped); PUT(dual_dropped);
    PUT(abs_presidual); PUT(abs_dresidual); PUT(rel_presidual); PUT(rel_dresidual);
    PUT(pobjval); PUT(dobjval); PUT(rel_objgap); PUT(complementarity); PUT(normx); PUT(normy); PUT(normz);
    PUT(objval); PUT(primal_infeas); PUT(dual_infeas);
    PUT(iter); PUT(kktiter1); PUT(kktiter2); PUT(basis_repairs); PUT(updates_start); PUT(updates_ipm);
    PUT(updates_crossover);
    PUT(time_total); PUT(time_ipm1); PUT(time_ipm2); PUT(time_starting_basis); PUT(time_crossover);
    PUT(time_kkt_factorize); PUT(time_kkt_solve); PUT(time_maxvol); PUT(time_cr1); PUT(time_cr2);
    PUT(time_cr2_NNt); PUT(time_cr2_B); PUT(time_cr2_Bt); PUT(time_lu_invert); PUT(time_lu_update);
    PUT(mean_fill); PUT(max_fill); PUT(volume_increase);
#undef PUT
    f << "lu_factorizations " << dropin_lu_factorizations() << '\n';
    f << "lu_reused " << dropin_lu_reused() << '\n';
    f << "lu_max_bump " << dropin_lu_max_bump() << '\n';
    f << "lu_device_seconds " << dropin_lu_seconds() << '\n';
#ifdef IPX_LP_HIP
    f << "cpu_prepare_seconds " << ipx_hip_cpu_prepare_seconds() << '\n';
    f << "cpu_prepare_calls " << ipx_hip_cpu_prepare_calls() << '\n';
    // one device model per Model: the three solver objects of LpSolver::Solve share one context (hip_device.h)
    f << "device_maxvolume_calls " << ipx_hip_device_maxvolume_calls() << '\n';
    f << "cpu_maxvolume_calls " << ipx_hip_cpu_maxvolume_calls() << '\n';
    f << "kept_eta_calls " << ipx_hip_kept_eta_calls() << '\n';
    f << "kept_etas " << ipx_hip_kept_etas() << '\n';
    {
        const char* names[5] = {"factorize_drop_seconds", "factorize_device_lu_prepare_seconds", "factorize_device_maxvolume_seconds",
                                "factorize_basis_load_seconds", "factorize_cpu_path_seconds"};
        for (int p = 0; p < 5; p++) f << names[p] << ' ' << ipx_hip_factorize_phase_seconds(p) << '\n';
    }
    f << "hip_model_creations " << ipx::HipModel::creations() << '\n';
    f << "hip_model_hits " << ipx::HipModel::hits() << '\n';
#endif

    if (info.status_ipm != IPX_STATUS_not_run) {
        std::vector<double> x(num_var), xl(num_var), xu(num_var), slack(num_constr), y(num_constr), zl(num_var),
            zu(num_var);
        if (solver.GetInteriorSolution(x.data(), xl.data(), xu.data(), slack.data(), y.data(), zl.data(),
                                       zu.data()) == 0) {
            WriteBin(out + "/x.bin", x.data(), x.size());
            WriteBin(out + "/xl.bin", xl.data(), xl.size());
            WriteBin(out + "/xu.bin", xu.data(), xu.size());
            WriteBin(out + "/slack.bin", slack.data(), slack.size());
            WriteBin(out + "/y.bin", y.data(), y.size());
            WriteBin(out + "/zl.bin", zl.data(), zl.size());
            WriteBin(out + "/zu.bin", zu.data(), zu.size());
        }
    }
    if (info.status_crossover == IPX_STATUS_optimal || info.status_crossover == IPX_STATUS_imprecise) {
        std::vector<double> x(num_var), slack(num_constr), y(num_constr), z(num_var);
        std::vector<ipxint> cbasis(num_constr), vbasis(num_var);
        if (solver.GetBasicSolution(x.data(), slack.data(), y.data(), z.data(), cbasis.data(), vbasis.data()) == 0) {
            WriteBin(out + "/bx.bin", x.data(), x.size());
            WriteBin(out + "/bslack.bin", slack.data(), slack.size());
            WriteBin(out + "/by.bin", y.data(), y.size());
            WriteBin(out + "/bz.bin", z.data(), z.size());
            WriteBin(out + "/cbasis.bin", cbasis.data(), cbasis.size());
            WriteBin(out + "/vbasis.bin", vbasis.data(), vbasis.size());
        }
    }
    std::printf("status %ld status_ipm %ld status_crossover %ld errflag %ld iter %ld kktiter1 %ld kktiter2 %ld "
                "updates_ipm %ld pobjval %.12g dobjval %.12g lu_factorizations %ld lu_max_bump %ld\n",
                (long)info.status, (long)info.status_ipm, (long)info.status_crossover, (long)info.errflag,
                (long)info.iter, (long)info.kktiter1, (long)info.kktiter2, (long)info.updates_ipm, info.pobjval,
                info.dobjval, dropin_lu_factorizations(), dropin_lu_max_bump());
    std::printf("DONE\n");
    return 0;
}
