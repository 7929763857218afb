// Pieces shared by the TSP and CVRP state kernels.
#pragma once
#include "common.h"

// Row mean / population std of the pairwise distance matrix of one instance, without
// storing the matrix (TSP/TSPEnv.py:228-235, CVRP/CVRPEnv.py:475-482).
// grid: (ceil(M / 256), B), block 256; thread t handles node i = 256*blockIdx.x + t and sums
// d(i, j) over j = 0..M-1 in binary64 (four interleaved partial sums, rounded once at the end).  The instance's coordinates are
// staged once per block in LDS (dynamic, 8*M bytes) and read back as broadcasts (every lane reads
// the same j).  Also clears the visited bitmask.  z (may be NULL): a third coordinate per node
// ([B][M]; the CVRP training reset measures distances over (x, y, demand), CVRP/CVRPEnv.py:519):
// |p|^2 = (x^2 + y^2) + z^2 and the product chain gets one more FMA.
__device__ __forceinline__ void dgl_reset_rows(const float *__restrict__ xy, const float *__restrict__ z, int M,
                                               int rows_per_instance,
                                               float *__restrict__ avg, float *__restrict__ sd,
                                               uint64_t *__restrict__ visited) {
    extern __shared__ float lds_xy[];   // [M][2]
    const int b = blockIdx.y;
    const float *xyb = xy + (size_t)b * M * 2;
    for (int e = threadIdx.x; e < 2 * M; e += blockDim.x) lds_xy[e] = xyb[e];
    float *lds_z = lds_xy + 2 * M;
    if (z)
        for (int e = threadIdx.x; e < M; e += blockDim.x) lds_z[e] = z[(size_t)b * M + e];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) {
        const float xi = lds_xy[2 * i], yi = lds_xy[2 * i + 1];
        const float2 *pts = reinterpret_cast<const float2 *>(lds_xy);
        double s = 0.0, q = 0.0;
        if (z == nullptr) {
            const float ni = dgl_sqnorm(xi, yi);
            // four independent accumulator pairs: the binary64 adds of one pair would otherwise form one dependent chain
            double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
            int j = 0;
            for (; j + 4 <= M; j += 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 pj = pts[j + e];
                    const double d = (double)dgl_chain_dist(xi, yi, ni, pj.x, pj.y, dgl_sqnorm(pj.x, pj.y));
                    s4[e] += d;
                    q4[e] = fma(d, d, q4[e]);
                }
            }
            for (; j < M; ++j) {
                const float2 pj = pts[j];
                const double d = (double)dgl_chain_dist(xi, yi, ni, pj.x, pj.y, dgl_sqnorm(pj.x, pj.y));
                s4[0] += d;
                q4[0] = fma(d, d, q4[0]);
            }
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
        } else {
            const float zi = lds_z[i], ni = __fadd_rn(dgl_sqnorm(xi, yi), __fmul_rn(zi, zi));
            auto dist3 = [&](int j) {
                const float2 pj = pts[j];
                const float zj = lds_z[j], nj = __fadd_rn(dgl_sqnorm(pj.x, pj.y), __fmul_rn(zj, zj));
                const float cross = __fmaf_rn(-2.0f * zi, zj, dgl_chain_cross(xi, yi, pj.x, pj.y));
                return (double)dgl_sqrt(dgl_chain_sq(cross, ni, nj));
            };
            double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
            int j = 0;
            for (; j + 4 <= M; j += 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = dist3(j + e);
                    s4[e] += d;
                    q4[e] = fma(d, d, q4[e]);
                }
            }
            for (; j < M; ++j) {
                const double d = dist3(j);
                s4[0] += d;
                q4[0] = fma(d, d, q4[0]);
            }
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
        }
        const double mean = s / M;
        const double var = q / M - mean * mean;
        const float favg = (float)mean, fsd = (float)sqrt(var > 0.0 ? var : 0.0);
        for (int p = 0; p < rows_per_instance; ++p) {
            const size_t r = (size_t)b * rows_per_instance + p;
            avg[r * M + i] = favg;
            sd[r * M + i] = fsd;
        }
    }
    if (blockIdx.x == 0) {
        const int W = (M + 63) / 64;
        size_t base = (size_t)b * rows_per_instance * W;
        for (int e = threadIdx.x; e < rows_per_instance * W; e += blockDim.x) visited[base + e] = 0ull;
    }
}
