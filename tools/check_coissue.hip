// What a vector instruction costs in the shadow of an MFMA, one wave per SIMD (the policy kernel's situation):
// a stream of v_mfma_f32_32x32x16_f16 (two accumulators in turn) with F filler instructions of one kind written after
// every MFMA; cycles per MFMA from s_memtime.  32 = the matrix pipe's pace; anything above is what the fillers cost.
//   hipcc --offload-arch=gfx950 -O3 tools/check_coissue.hip -o /tmp/check_coissue && /tmp/check_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int KIND, int F>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&u)[4], float (&ag)[8], float *lds) {
#pragma unroll
    for (int i = 0; i < F; ++i) {
        if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]));
        if constexpr (KIND == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(x[i & 7]), "v"(x[(i + 1) & 7]));
        if constexpr (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x[i & 7]) : "v"(u[i & 3]), "v"(x[(i + 2) & 7]));
        if constexpr (KIND == 3) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(x[7]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
        if constexpr (KIND == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[i & 7]) : "a"(ag[i & 7]));
        if constexpr (KIND == 5) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(threadIdx.x * 8 + 2048 * (i & 3))), "v"(*reinterpret_cast<double *>(&x[2 * (i & 3)])) : "memory");
        if constexpr (KIND == 6) asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i & 7]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 5) & 7]));
        if constexpr (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[7]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
        if constexpr (KIND == 9) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i & 7]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
        if constexpr (KIND == 10) asm volatile("v_max_f32 %0, %0, |%1|" : "+v"(x[4 + (i & 3)]) : "v"(x[i & 3]));
        if constexpr (KIND == 11) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) float *>(&x[4 * (i & 1)])) : "v"((unsigned)(threadIdx.x * 16 + 4096 * (i & 3))) : "memory");
        if constexpr (KIND == 12) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(u[(i + 1) & 3]), "v"(u[(i + 2) & 3]));
        if constexpr (KIND == 13) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[3]) : "v"(u[i & 1]), "v"(u[(i + 1) & 1]));
        if constexpr (KIND == 14) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(x[i & 7]), "v"(x[(i + 1) & 7]));
        if constexpr (KIND == 15) asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(u[i & 3]) : "v"(x[i & 7]));
        if constexpr (KIND == 7) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double *>(&x[2 * (i & 3)])) : "v"(*reinterpret_cast<double *>(&x[2 * ((i + 1) & 3)])), "v"(*reinterpret_cast<double *>(&x[2 * ((i + 2) & 3)])));
    }
}

template <int KIND, int F>
__global__ __launch_bounds__(256) void kern(float *out, long long *cyc, int iters) {
    extern __shared__ float lds[];
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
    float x[8], ag[8];
    unsigned u[4] = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    for (int i = 0; i < 8; ++i) { x[i] = 1.0f + 0.001f * (threadIdx.x + i); ag[i] = x[i]; }
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(ag[i]));
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            MFMA(acc0, a, b);
            fillers<KIND, F>(x, u, ag, lds);
            MFMA(acc1, a, b);
            fillers<KIND, F>(x, u, ag, lds);
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int F>
static double run(float *out, long long *cyc, int blocks) {
    const int iters = 200;
    hipLaunchKernelGGL((kern<KIND, F>), dim3(blocks), dim3(256), 32768, 0, out, cyc, iters);
    hipLaunchKernelGGL((kern<KIND, F>), dim3(blocks), dim3(256), 32768, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / blocks / (iters * 48.0);
}

template <int KIND>
static void row(const char *name, float *out, long long *cyc, int blocks) {
    printf("%-28s", name);
    printf(" %6.1f", run<KIND, 0>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 1>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 2>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 3>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 4>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 5>(out, cyc, blocks));
    printf(" %6.1f", run<KIND, 6>(out, cyc, blocks));
    printf(" %6.1f\n", run<KIND, 8>(out, cyc, blocks));
}

int main() {
    const int blocks = 256;
    float *out;
    long long *cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    printf("cycles per MFMA with F fillers after each     F=0      1      2      3      4      5      6      8\n");
    row<0>("v_add_f32", out, cyc, blocks);
    row<6>("v_max_f32", out, cyc, blocks);
    row<1>("v_cvt_pkrtz_f16_f32", out, cyc, blocks);
    row<14>("v_cvt_pk_f16_f32", out, cyc, blocks);
    row<15>("v_cvt_f16_f32", out, cyc, blocks);
    row<2>("v_fma_mix_f32", out, cyc, blocks);
    row<3>("v_max3_f32 |.|", out, cyc, blocks);
    row<4>("v_accvgpr_read_b32", out, cyc, blocks);
    row<8>("v_max3_f32", out, cyc, blocks);
    row<10>("v_max_f32 |.|", out, cyc, blocks);
    row<13>("v_max3_u32", out, cyc, blocks);
    row<9>("v_fma_f32", out, cyc, blocks);
    row<12>("v_and_b32", out, cyc, blocks);
    row<7>("v_pk_add_f32", out, cyc, blocks);
    row<11>("ds_read_b128", out, cyc, blocks);
    row<5>("ds_write_b64", out, cyc, blocks);
    return 0;
}
