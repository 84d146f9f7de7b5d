// Shared plumbing of the IPX-side classes: owns one ipxk_context (the model's matrix on one
// GPU) and maps ABI return codes onto the exceptions LpSolver::Solve already handles
// (reference src/lp_solver.cc:98-105: std::bad_alloc -> IPX_STATUS_out_of_memory, anything
// else -> IPX_STATUS_internal_error).
#ifndef IPX_HIP_DEVICE_H_
#define IPX_HIP_DEVICE_H_

#include <new>
#include <stdexcept>
#include <string>

#include "device_glue.h"
#include "ipx_kkt_hip.h"
#include "model.h"

namespace ipx {

inline void HipCheck(int rc) { ipx_hip::Check(rc); }

class HipModel {
public:
    // Uploads the n structural columns of model.AI() (the slack identity is implicit).
    explicit HipModel(const Model& model, int device = 0) {
        const SparseMatrix& AI = model.AI();
        HipCheck(ipxk_create(model.rows(), model.cols(), AI.colptr(), AI.rowidx(), AI.values(),
                             device, &ctx_));
    }
    ~HipModel() { ipxk_destroy(ctx_); }
    HipModel(const HipModel&) = delete;
    HipModel& operator=(const HipModel&) = delete;
    ipxk_context* get() const { return ctx_; }
private:
    ipxk_context* ctx_{nullptr};
};

}  // namespace ipx

#endif  // IPX_HIP_DEVICE_H_
