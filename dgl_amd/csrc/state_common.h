// Pieces shared by the TSP and CVRP state kernels.
#pragma once
#include <type_traits>
#include "common.h"

// Row mean / population std of the pairwise distance matrix of one instance, without
// storing the matrix (TSP/TSPEnv.py:228-235, CVRP/CVRPEnv.py:475-482).
// grid: (ceil(M / 256), B), block 256; thread t handles node i = 256*blockIdx.x + t and sums
// d(i, j) over j = 0..M-1 in binary64 (four interleaved partial sums, rounded once at the end).  The instance's coordinates are
// staged once per block in LDS (dynamic, 8*M bytes) and read back as broadcasts (every lane reads
// the same j).  Also clears the visited bitmask.  z (may be NULL): a third coordinate per node
// ([B][M]; the CVRP training reset measures distances over (x, y, demand), CVRP/CVRPEnv.py:519):
// |p|^2 = (x^2 + y^2) + z^2 and the product chain gets one more FMA.
// M <= DGL_DIRECT_MAX_POINTS: the direct evaluation (common.h, dgl_direct_sq; a third coordinate adds fma(dz, dz, .)).
__device__ __forceinline__ void dgl_reset_rows(const float *__restrict__ xy, const float *__restrict__ z, int M,
                                               int rows_per_instance,
                                               float *__restrict__ avg, float *__restrict__ sd,
                                               uint64_t *__restrict__ visited) {
    const bool direct = M <= DGL_DIRECT_MAX_POINTS;
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
                    const double d = (double)(direct ? dgl_sqrt(dgl_direct_sq(xi, yi, pj.x, pj.y))
                                                     : dgl_chain_dist(xi, yi, ni, pj.x, pj.y, dgl_sqnorm(pj.x, pj.y)));
                    s4[e] += d;
                    q4[e] = fma(d, d, q4[e]);
                }
            }
            for (; j < M; ++j) {
                const float2 pj = pts[j];
                const double d = (double)(direct ? dgl_sqrt(dgl_direct_sq(xi, yi, pj.x, pj.y))
                                                 : dgl_chain_dist(xi, yi, ni, pj.x, pj.y, dgl_sqnorm(pj.x, pj.y)));
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
                if (direct) {
                    const float dz = __fsub_rn(zi, zj);
                    return (double)dgl_sqrt(__fmaf_rn(dz, dz, dgl_direct_sq(xi, yi, pj.x, pj.y)));
                }
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

// torch's float sum over the last dimension of a contiguous matrix, for one row of M elements, by the 32 lanes of half a
// wave (lane (k, v) = accumulator k, vector lane v): ATen's cascade sum as described at dgl_reset_rows_torch below, which
// interleaves its binary64 sums with the same loops; here for the tour lengths (TSP/TSPEnv.py:560-562,
// CVRP/CVRPEnv.py:991).  value(j, std::true_type) is called by all 64 lanes of the wave together (a model's wave-wide
// root may be taken there), value(j, std::false_type) under a lane mask.  The row's sum is returned on every lane.
template <class F>
__device__ __forceinline__ float dgl_torch_row_sum(int M, F value) {
    const std::true_type together{};
    const std::false_type alone{};
    const int l = threadIdx.x & 31, k = l >> 3, v = l & 7;
    const int vec_size = M / 8, size_ilp = vec_size / 4;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;          // cascade levels
    int t = 0;
    while (t + 16 <= size_ilp) {
        for (int u = 0; u < 16; ++u, ++t) a0 = __fadd_rn(a0, value(32 * t + l, together));
        a1 = __fadd_rn(a1, a0); a0 = 0.0f;
        if ((t & (15 << 4)) == 0) {
            a2 = __fadd_rn(a2, a1); a1 = 0.0f;
            if ((t & (15 << 8)) == 0) { a3 = __fadd_rn(a3, a2); a2 = 0.0f; }
        }
    }
    for (; t < size_ilp; ++t) a0 = __fadd_rn(a0, value(32 * t + l, together));
    a0 = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
    for (int w = 4 * size_ilp; w < vec_size; ++w)              // left-over vectors: accumulator 0 only
        if (k == 0) a0 = __fadd_rn(a0, value(8 * w + v, alone));
    const int base = threadIdx.x & 32;                         // first lane of this half of the wave
    float p0 = a0;                                             // accumulators 1, 2, 3 into accumulator 0, in that order
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) p0 = __fadd_rn(p0, __shfl(a0, base + 8 * kk + v, 64));
    float total = 0.0f;                                        // scalar tail, then the eight lanes in order
    for (int j = 8 * vec_size; j < M; ++j) total = __fadd_rn(total, value(j, together));
#pragma unroll
    for (int vv = 0; vv < 8; ++vv) total = __fadd_rn(total, __shfl(p0, base + vv, 64));
    return total;
}

// The same reset with torch's own arithmetic for the row means, for the square-root models that aim at the reference's
// bits (DglSqrtMkl): torch.mean over the last dimension of a contiguous float matrix is ATen's cascade sum
// (aten/src/ATen/native/cpu/SumKernel.cpp) with 8-lane vectors, whatever the CPU: a row of M floats is read as
// M / 8 vectors; four interleaved vector accumulators (element j of the first 32 (M / 32) goes to accumulator
// (j / 8) % 4, lane j % 8) each sum their sequence through a cascade that folds level 0 into level 1 after every 16
// terms, level 1 into level 2 after every 256, ...; the left-over vectors go to accumulator 0, accumulators 1..3 are
// added to it in order, then a scalar adds the row's last M % 8 elements and the eight lanes, in that order; the mean is
// that sum divided by M in binary32 (checked bit for bit against torch.mean for M = 27 ... 10 000).  torch.std
// accumulates in binary64 and is the correctly rounded population std of the binary32 distances: sums in binary64 here.
// Half a wave owns one node i: lane (k, v) = accumulator k, lane v.  grid: (ceil(M / 8), B), block 256 (8 nodes).
template <class SQ>
__device__ __forceinline__ void dgl_reset_rows_torch(const float *__restrict__ xy, const float *__restrict__ z, int M,
                                                     int rows_per_instance, float *__restrict__ avg, float *__restrict__ sd,
                                                     uint64_t *__restrict__ visited, const SQ &sq) {
    extern __shared__ float lds_xy[];   // [M][2] (+ [M] third coordinates)
    const int b = blockIdx.y;
    const float *xyb = xy + (size_t)b * M * 2;
    for (int e = threadIdx.x; e < 2 * M; e += blockDim.x) lds_xy[e] = xyb[e];
    float *lds_z = lds_xy + 2 * M;
    if (z)
        for (int e = threadIdx.x; e < M; e += blockDim.x) lds_z[e] = z[(size_t)b * M + e];
    __syncthreads();
    const int l = threadIdx.x & 31, k = l >> 3, v = l & 7;
    const int i = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    const bool live = i < M;
    const int ii = live ? i : 0;
    const float2 *pts = reinterpret_cast<const float2 *>(lds_xy);
    const float xi = pts[ii].x, yi = pts[ii].y, zi = z ? lds_z[ii] : 0.0f;
    const float ni = z ? __fadd_rn(dgl_sqnorm(xi, yi), __fmul_rn(zi, zi)) : dgl_sqnorm(xi, yi);
    const bool direct = M <= DGL_DIRECT_MAX_POINTS;            // (common.h, dgl_direct_sq: correctly rounded roots there)
    // ALL: called by all 64 lanes together (the model's wave-wide root, common.h): the interleaved part of the row
    const typename SQ::Lanes ln = sq.lanes();
    auto dist = [&](int j, auto all_lanes) {
        constexpr bool ALL = decltype(all_lanes)::value;
        const float2 pj = pts[j];
        if (direct) {
            float d2 = dgl_direct_sq(xi, yi, pj.x, pj.y);
            if (z) { const float dz = __fsub_rn(zi, lds_z[j]); d2 = __fmaf_rn(dz, dz, d2); }
            return dgl_sqrt(d2);
        }
        float cross = dgl_chain_cross(xi, yi, pj.x, pj.y), nj = dgl_sqnorm(pj.x, pj.y);
        if (z) {
            const float zj = lds_z[j];
            cross = __fmaf_rn(-2.0f * zi, zj, cross);
            nj = __fadd_rn(nj, __fmul_rn(zj, zj));
        }
        const float r2 = dgl_chain_sq(cross, ni, nj);
        return ALL ? sq.all(ln, r2) : sq(r2);
    };
    const std::true_type together{};
    const std::false_type alone{};
    const int vec_size = M / 8, size_ilp = vec_size / 4;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;          // cascade levels (level step 16 up to 2^19 vectors per accumulator)
    double s64 = 0.0, q64 = 0.0;
    int t = 0;
    while (t + 16 <= size_ilp) {
        for (int u = 0; u < 16; ++u, ++t) {
            const float d = dist(32 * t + l, together);
            a0 = __fadd_rn(a0, d);
            s64 += (double)d;
            q64 = fma((double)d, (double)d, q64);
        }
        a1 = __fadd_rn(a1, a0); a0 = 0.0f;
        if ((t & (15 << 4)) == 0) {
            a2 = __fadd_rn(a2, a1); a1 = 0.0f;
            if ((t & (15 << 8)) == 0) { a3 = __fadd_rn(a3, a2); a2 = 0.0f; }
        }
    }
    for (; t < size_ilp; ++t) {
        const float d = dist(32 * t + l, together);
        a0 = __fadd_rn(a0, d);
        s64 += (double)d;
        q64 = fma((double)d, (double)d, q64);
    }
    a0 = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
    // left-over vectors: accumulator 0 only
    for (int w = 4 * size_ilp; w < vec_size; ++w) {
        if (k == 0) {
            const float d = dist(8 * w + v, alone);
            a0 = __fadd_rn(a0, d);
            s64 += (double)d;
            q64 = fma((double)d, (double)d, q64);
        }
    }
    // accumulators 1, 2, 3 into accumulator 0 (lanes k = 0 of this half wave), in that order
    const int base = threadIdx.x & 32;                         // first lane of this half of the wave
    float p0 = a0;
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) p0 = __fadd_rn(p0, __shfl(a0, base + 8 * kk + v, 64));
    // scalar tail, then the eight lanes in order
    float total = 0.0f;
    for (int j = 8 * vec_size; j < M; ++j) {
        const float d = dist(j, together);
        total = __fadd_rn(total, d);
        if (l == 0) { s64 += (double)d; q64 = fma((double)d, (double)d, q64); }
    }
#pragma unroll
    for (int vv = 0; vv < 8; ++vv) total = __fadd_rn(total, __shfl(p0, base + vv, 64));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {                          // binary64 sums over the half wave
        s64 += __shfl_xor(s64, o, 64);
        q64 += __shfl_xor(q64, o, 64);
    }
    if (live && l == 0) {
        const float favg = __fdiv_rn(total, (float)M);
        const double mean = s64 / M, var = q64 / M - mean * mean;
        const float fsd = (float)sqrt(var > 0.0 ? var : 0.0);
        for (int p = 0; p < rows_per_instance; ++p) {
            const size_t r = (size_t)b * rows_per_instance + p;
            avg[r * M + i] = favg;
            sd[r * M + i] = fsd;
        }
    }
    if (blockIdx.x == 0) {
        const int W = (M + 63) / 64;
        size_t base_w = (size_t)b * rows_per_instance * W;
        for (int e = threadIdx.x; e < rows_per_instance * W; e += blockDim.x) visited[base_w + e] = 0ull;
    }
}
