// Device-side helpers: wave64 / workgroup reductions with a fixed combination
// tree (results are run-to-run reproducible; no atomics anywhere).
#pragma once

#include <hip/hip_runtime.h>

#include "internal.hpp"

namespace ipxk {

struct SumOp {
    static __device__ __forceinline__ double identity() { return 0.0; }
    static __device__ __forceinline__ double apply(double a, double b) { return a + b; }
};
struct MaxOp {
    static __device__ __forceinline__ double identity() { return 0.0; }  // norms are >= 0
    // NaN-propagating max so that a NaN residual cannot pass the tolerance test
    static __device__ __forceinline__ double apply(double a, double b) {
        return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
    }
};
struct MinOp {
    static __device__ __forceinline__ double identity() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double apply(double a, double b) { return a < b ? a : b; }
};

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = Op::apply(v, __shfl_xor(v, off, 64));
    return v;
}

// All kBlock threads call; every thread receives the result.  `scratch` is a
// workgroup-shared array of at least kBlock/64 + 1 doubles.
template <class Op>
__device__ __forceinline__ double block_reduce(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_reduce<Op>(v);
    __syncthreads();  // scratch may still be read by a previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) r = Op::apply(r, scratch[w]);
    return r;
}

// Reduces `count` per-workgroup partials written by an EARLIER kernel on the
// same stream.  Every workgroup of the consumer kernel calls this and obtains
// the bitwise identical value (fixed order: strided per thread, then the block
// tree) -- this replaces a separate "finalize" launch per scalar.
// `stride` > 1 addresses one scalar per rank in the all-gathered array of a
// multi-GPU run (comm.hip); single GPU: the producer's partial array, stride 1.
struct PartRef {
    const double* p;
    int count;
    int stride;
};

template <class Op>
__device__ __forceinline__ double reduce_partials(PartRef part, double* scratch) {
    double v = Op::identity();
    for (int i = threadIdx.x; i < part.count; i += kBlock)
        v = Op::apply(v, part.p[(size_t)i * part.stride]);
    return block_reduce<Op>(v, scratch);
}

}  // namespace ipxk
