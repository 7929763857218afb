// Measurement helpers for bench.py (NOT part of the product's C ABI; built into libdgl_peaks.so): what this
// MI355X itself delivers for a streaming kernel and for back-to-back matrix instructions on random operands, to
// quote beside the nominal peaks.  Random operands matter: the chip holds a lower clock on live data than on zeros.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// a = b + s * c over float4 (STREAM triad: two reads, one write per element) and a = b (one read, one write: the mix
// of the state-update kernels).  A workgroup handles U * 256 consecutive float4 (U independent loads per thread);
// NT = non-temporal loads and stores.  n4 must be a multiple of U * 256.
template <int U, bool NT, bool TRIAD>
__global__ __launch_bounds__(256) void stream_kernel(f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                     const f32x4 *__restrict__ c, float s) {
    const long i0 = (long)blockIdx.x * (256 * U) + threadIdx.x;
    f32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x[u] = NT ? __builtin_nontemporal_load(b + i0 + u * 256) : b[i0 + u * 256];
        if (TRIAD) y[u] = NT ? __builtin_nontemporal_load(c + i0 + u * 256) : c[i0 + u * 256];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f32x4 v = TRIAD ? x[u] + s * y[u] : x[u];
        if (NT) __builtin_nontemporal_store(v, a + i0 + u * 256);
        else a[i0 + u * 256] = v;
    }
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// four independent accumulators per wave, operands in registers, random finite values in (-2, 2)
__global__ __launch_bounds__(256) void mfma_f16_kernel(int iters, float *__restrict__ sink) {
    f16x8 a[4], b[4];
    f32x16 acc[4];
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < 4; ++k) {
        for (int e = 0; e < 8; ++e) {
            a[k][e] = (_Float16)(((int)(hash32(t * 64 + k * 16 + e) & 0xFFFF) - 32768) * (1.0f / 16384.0f));
            b[k][e] = (_Float16)(((int)(hash32(t * 64 + k * 16 + e + 8) & 0xFFFF) - 32768) * (1.0f / 16384.0f));
        }
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    if (s == 123.456f) sink[0] = s;                       // keeps the loop alive
}
// the same flops per wave with the 16x16x32 shape: eight independent 16x16 accumulators
__global__ __launch_bounds__(256) void mfma_f16_16_kernel(int iters, float *__restrict__ sink) {
    f16x8 a[4], b[4];
    f32x4 acc[8];
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < 4; ++k) {
        for (int e = 0; e < 8; ++e) {
            a[k][e] = (_Float16)(((int)(hash32(t * 64 + k * 16 + e) & 0xFFFF) - 32768) * (1.0f / 16384.0f));
            b[k][e] = (_Float16)(((int)(hash32(t * 64 + k * 16 + e + 8) & 0xFFFF) - 32768) * (1.0f / 16384.0f));
        }
    }
    for (int k = 0; k < 8; ++k)
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k & 3], b[k >> 1], acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k)
        for (int e = 0; e < 4; ++e) s += acc[k][e];
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void mfma_f32_kernel(int iters, float *__restrict__ sink) {
    float a[4], b[4];
    f32x16 acc[4];
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < 4; ++k) {
        a[k] = ((int)(hash32(t * 8 + k) & 0xFFFF) - 32768) * (1.0f / 16384.0f);
        b[k] = ((int)(hash32(t * 8 + k + 4) & 0xFFFF) - 32768) * (1.0f / 16384.0f);
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    if (s == 123.456f) sink[0] = s;
}

#define PEAKS_API extern "C" __attribute__((visibility("default")))
template <bool TRIAD>
static int launch_stream(float *a, const float *b, const float *c, float s, int64_t n, int unroll, int nt, void *stream) {
    const long n4 = n / 4;
    if (n4 % (256 * 4) != 0 || (unroll != 1 && unroll != 4)) return -1;
    const hipStream_t st = (hipStream_t)stream;
    f32x4 *A = (f32x4 *)a;
    const f32x4 *B = (const f32x4 *)b, *C = (const f32x4 *)c;
    const dim3 grid((unsigned)(n4 / (256 * unroll)));
    if (unroll == 1 && !nt) hipLaunchKernelGGL((stream_kernel<1, false, TRIAD>), grid, dim3(256), 0, st, A, B, C, s);
    if (unroll == 1 && nt) hipLaunchKernelGGL((stream_kernel<1, true, TRIAD>), grid, dim3(256), 0, st, A, B, C, s);
    if (unroll == 4 && !nt) hipLaunchKernelGGL((stream_kernel<4, false, TRIAD>), grid, dim3(256), 0, st, A, B, C, s);
    if (unroll == 4 && nt) hipLaunchKernelGGL((stream_kernel<4, true, TRIAD>), grid, dim3(256), 0, st, A, B, C, s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// n: floats, a multiple of 4096; unroll 1 or 4; nt != 0: non-temporal loads and stores
PEAKS_API int dglb_triad(float *a, const float *b, const float *c, float s, int64_t n, int unroll, int nt, void *stream) {
    return launch_stream<true>(a, b, c, s, n, unroll, nt, stream);
}
PEAKS_API int dglb_copy(float *a, const float *b, int64_t n, int unroll, int nt, void *stream) {
    return launch_stream<false>(a, b, b, 0.0f, n, unroll, nt, stream);
}
// blocks of 4 waves; flops = blocks * 4 waves * iters * 4 * (2 * 32 * 32 * K)
PEAKS_API int dglb_mfma_f16(int blocks, int iters, float *sink, void *stream) {
    hipLaunchKernelGGL(mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// iters x 8 MFMAs of 2 * 16 * 16 * 32 flops per wave
PEAKS_API int dglb_mfma_f16_16(int blocks, int iters, float *sink, void *stream) {
    hipLaunchKernelGGL(mfma_f16_16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
PEAKS_API int dglb_mfma_f32(int blocks, int iters, float *sink, void *stream) {
    hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
