// Shared device/host helpers for the gfx950 kernels.  Compiled with -ffp-contract=off:
// every fused multiply-add is written explicitly, so float results do not depend on the
// compiler's contraction choices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dgl_amd.h"

#define DGL_WAVE 64

// host-side error plumbing (api.cpp)
int dgl_set_error(int code, const char *fmt, ...);
#define DGL_CHECK_ARG(cond, ...)                                           \
    do {                                                                   \
        if (!(cond)) return dgl_set_error(DGL_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)
#define DGL_CHECK_LAUNCH(what)                                             \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

#ifdef __HIPCC__
// |p|^2 as the reference's x.pow(2).sum(-1): two rounded products, one rounded add.
__device__ __forceinline__ float dgl_sqnorm(float x, float y) { return __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)); }

// d(i,j) of torch.cdist's matrix-product evaluation (row i = query, column j):
//   [-2x_i, -2y_i, n_i, 1] . [x_j, y_j, 1, n_j] accumulated left to right with FMAs.
__device__ __forceinline__ float dgl_chain_dist(float xi, float yi, float ni, float xj, float yj, float nj) {
    float acc = __fmul_rn(-2.0f * xi, xj);
    acc = __fmaf_rn(-2.0f * yi, yj, acc);
    acc = __fadd_rn(acc, ni);
    acc = __fadd_rn(acc, nj);
    return sqrtf(fmaxf(acc, 0.0f));
}

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
#endif
