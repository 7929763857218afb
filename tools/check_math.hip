// Standalone exhaustive check of the arithmetic primitives in dgl_amd/csrc/common.h against the
// compiler's correctly rounded operations.  Build and run on an MI355X:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/check_math.hip -o /tmp/check_math && /tmp/check_math
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
int dgl_set_error(int, const char *, ...) { return 0; }
#include "../dgl_amd/csrc/common.h"

__global__ void check_sqrt(unsigned long long *bad, unsigned *first) {
    // every bit pattern; bad[0]: +0 and positive normal numbers and +inf, bad[2]: positive denormals,
    // bad[3]: negative numbers and NaNs (both sides must be NaN, or -0 for -0)
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < 0x100000000ull;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float a = dgl_sqrt(x), b = sqrtf(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
        if (!same) {
            const int cls = (i == 0 || (i >= 0x00800000ull && i <= 0x7F800000ull)) ? 0 : (i < 0x00800000ull ? 2 : 3);
            atomicAdd(bad + cls, 1ull);
            if (cls == 0) { atomicMin(first, (unsigned)i); atomicMax(first + 1, (unsigned)i); }
        }
    }
}

__device__ unsigned rng(unsigned &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ void check_div(unsigned long long *bad, int per_thread) {
    const int u = blockIdx.x + 1;                              // divisor 1..gridDim.x
    const DglConstDiv c = dgl_const_div((float)u);
    unsigned s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int i = 0; i < per_thread; ++i) {
        // dividends: random sign, exponents spread over [2^-40, 2^20]
        const unsigned m = rng(s);
        const int e = (int)(rng(s) % 61) - 40;
        float a = ldexpf(1.0f + (m >> 9) * (1.0f / 8388608.0f), e);
        if (m & 1) a = -a;
        if (i == 0) a = 0.0f;
        const float q = dgl_div(a, c), w = a / (float)u;
        if (__float_as_uint(q) != __float_as_uint(w)) atomicAdd(bad, 1ull);
    }
}

int main() {
    unsigned long long *bad; unsigned *first;
    hipMallocManaged(&bad, 32); hipMallocManaged(&first, 8);
    bad[0] = bad[1] = bad[2] = bad[3] = 0; first[0] = 0xFFFFFFFFu; first[1] = 0;
    check_sqrt<<<4096, 256>>>(bad, first);
    hipDeviceSynchronize();
    printf("sqrt: %llu mismatches over +0, all positive normal floats, +inf (bits 0x%08x .. 0x%08x); "
           "%llu over positive denormals; %llu over negative/NaN inputs\n", bad[0], first[0], first[1], bad[2], bad[3]);
    check_div<<<20000, 256>>>(bad + 1, 4096);
    hipDeviceSynchronize();
    printf("div : %llu mismatches over 20000 divisors x 1048576 dividends\n", bad[1]);
    return (bad[0] || bad[1]) ? 1 : 0;
}
