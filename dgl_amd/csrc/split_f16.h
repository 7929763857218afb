// Splitting binary32 values into two binary16 halves for the split-operand matrix products
// (x ~= hi + lo, 22 significant bits; a product is evaluated as lo*hi + hi*lo + hi*hi on the f16 matrix cores).
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Split of activations: hi = binary16(v) rounded towards zero (one packed conversion for two values),
// lo = binary16(v - hi), the difference formed exactly by one mixed-precision FMA that reads hi
// straight from the packed register: four VALU operations per pair, 22 significant bits kept.
// Magnitudes above 65504 saturate in hi and continue in lo; below 2^-14 the absolute accuracy is 2^-24.
__device__ __forceinline__ unsigned pack_rtz(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// a - (low / high half of hh, read as binary16), exact in binary32
//
// The mixed-precision FMA is only reachable through inline assembly (the compiler turns the expression into a
// conversion and a subtraction: one more instruction per value, the same bits).  Inline assembly is invisible to the
// compiler's hazard recognizer, which is what places the wait states between a matrix instruction and a vector
// instruction that overwrites its accumulator input: with matrix results in ordinary VGPRs (the form the compiler
// picks for kernels limited to 256 registers, or under -mllvm -amdgpu-mfma-vgpr-form) a freed accumulator register can
// become the assembly's output while the matrix instruction still reads it - a policy kernel built that way gave
// different bits from run to run, and the same build without the assembly did not.  With matrix results in AGPRs the
// assembly cannot touch them.  So: the assembly only in kernels whose accumulators live in AGPRs (the decode policy
// kernels, more than 256 registers); every other user defines DGL_SPLIT_NO_ASM before including this header.
#ifdef DGL_SPLIT_NO_ASM
typedef _Float16 dgl_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float minus_half_lo(float a, unsigned hh) { return a - (float)__builtin_bit_cast(dgl_h2, hh)[0]; }
__device__ __forceinline__ float minus_half_hi(float a, unsigned hh) { return a - (float)__builtin_bit_cast(dgl_h2, hh)[1]; }
#else
__device__ __forceinline__ float minus_half_lo(float a, unsigned hh) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(a));
    return r;
}
__device__ __forceinline__ float minus_half_hi(float a, unsigned hh) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(a));
    return r;
}
#endif
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = pack_rtz(a, b);
    lo = pack_rtz(minus_half_lo(a, hi), minus_half_hi(b, hi));
}
