// Pieces shared by the TSP and CVRP state kernels.
#pragma once
#include "common.h"

// Row mean / population std of the pairwise distance matrix of one instance, without
// storing the matrix (TSP/TSPEnv.py:228-235, CVRP/CVRPEnv.py:475-482).
// grid: (ceil(M / 4), B), block 256 = 4 waves; wave w handles node i = 4*blockIdx.x + w and
// sums d(i, j) over j in binary64 (rounded once at the end).  The instance's coordinates are
// staged once per block in LDS (dynamic, 8*M bytes).  Also clears the visited bitmask.
__device__ __forceinline__ void dgl_reset_rows(const float *__restrict__ xy, int M, int rows_per_instance,
                                               float *__restrict__ avg, float *__restrict__ sd,
                                               uint64_t *__restrict__ visited) {
    extern __shared__ float lds_xy[];   // [M][2]
    const int b = blockIdx.y;
    const float *xyb = xy + (size_t)b * M * 2;
    for (int e = threadIdx.x; e < 2 * M; e += blockDim.x) lds_xy[e] = xyb[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i < M) {
        const float xi = lds_xy[2 * i], yi = lds_xy[2 * i + 1], ni = dgl_sqnorm(xi, yi);
        double s = 0.0, q = 0.0;
        for (int j = lane; j < M; j += 64) {
            float xj = lds_xy[2 * j], yj = lds_xy[2 * j + 1];
            double d = (double)dgl_chain_dist(xi, yi, ni, xj, yj, dgl_sqnorm(xj, yj));
            s += d;
            q = fma(d, d, q);
        }
        s = wave_sum_f64(s);
        q = wave_sum_f64(q);
        double mean = s / M;
        double var = q / M - mean * mean;
        float favg = (float)mean, fsd = (float)sqrt(var > 0.0 ? var : 0.0);
        for (int p = lane; p < rows_per_instance; p += 64) {
            size_t r = (size_t)b * rows_per_instance + p;
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
