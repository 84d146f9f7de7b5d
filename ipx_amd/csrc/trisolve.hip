// placeholder until the basis path lands (filled in below in this round)
#include "context.hpp"
namespace ipxk {
struct SplitOperator {};
void destroy_split(SplitOperator* s) { delete s; }
void split_prepare_host(Context*, const ipxint*, const ipxint*, const double*, const ipxint*, const ipxint*,
                        const double*, const ipxint*, const ipxint*, const ipxint*, const ipxint*,
                        const double*) { throw Error(IPXK_E_UNSUPPORTED, "basis path not built yet"); }
int split_apply_dev(Context*, const double*, double*, const int*) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
void forward_solve_dev(Context*, double*, bool, const int*) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
void backward_solve_dev(Context*, double*, bool, const int*) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
void solve_dense_dev(Context*, const double*, double*, char) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
CrResult kkt_basis_solve_dev(Context*, const double*, const double*, double, ipxint, double*, double*,
                             ipxk_interrupt_fn, void*, ipxk_times*) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
void split_levels(const Context*, ipxint*) { throw Error(IPXK_E_UNSUPPORTED, "basis path"); }
}
