// General (any number of tokens per row) forward of the policy network and its backward pass:
// the teacher-forced training step of CVRPTrainer._train_one_batch (CVRP/CVRPTrainer.py:232-282,
// model train branch CVRP/CVRPModel.py:311-345; TSP/TSPModel.py:206-217), and the same forward for
// inference when a row has more than one 32-token tile (knn + depot_knn + 2 > 32).
//
// Unlike the fused decode kernels this path is a sequence of kernels over the flattened token axis (M = R*n tokens),
// all on the f16 matrix cores with split operands and binary32 accumulation:
//   gemm_aw_kernel    activation x packed weight (Q, K, V; the input gradient of a block)
//   embed_qkv_kernel  the live tokens of the caller's tile, their embedding, the first block's Q, K, V
//   ff_fused_kernel   everything between a block's attention and the next block, forwards, in one launch
//   ff_bwd_kernel     the same chain backwards
//   attn_fwd_kernel, attn_bwd_kernel   attention per (row, head); probabilities recomputed in the backward pass
//   gemm_kernel       activation x activation: the weight gradients, as partial sums per chunk of tokens
//   output_kernel     logits, probabilities, greedy pick; the teacher's probability and d(-mean log p_teacher)/dX
// Sums over tokens (weight and bias gradients) are formed as partial sums per chunk of tokens and
// added in a fixed order: no float atomics, so gradients are bit-reproducible from run to run.
#include "policy_layout.h"
#define DGL_SPLIT_NO_ASM      // these kernels keep their matrix results in VGPRs: see split_f16.h
#include "split_f16.h"

#include <cstdlib>
#include <mutex>
#include <utility>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define HEADS 8
#define HDIM 16
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// ------------------------------------------------------------------------------- GEMM
// C[i][j] (op)= sum over segments s, kk of A_s(i,kk) B_s(kk,j),  A(i,kk) = A[i*sai + kk*sak],  B(kk,j) = B[kk*sbk + j*sbn],
// i < I, j < J, kk in the z-th chunk of [0, KK).  64x64 output tile per workgroup (4 waves, each a 32x32 MFMA tile),
// K staged 32 at a time through LDS.
//
// Arithmetic: binary32 operands are scaled by a power of two, split into binary16 halves (hi + lo, 22 significant
// bits) when they are staged, and multiplied as lo*hi + hi*lo + hi*hi on the f16 matrix cores with binary32
// accumulation - the fused decode kernel's arithmetic, at 16/3 of the f32-input matrix rate.  binary16 has a narrow
// exponent range (gradients of 1e-6 would lose their low half, activations beyond 65504 saturate), so every operand
// that is an activation or a gradient comes with a device scalar holding an upper bound of its magnitude (written by
// the kernel that produced it, atomic max: order-independent) and is scaled so that this bound lands at 2^10; the
// accumulator is scaled back exactly.  Weights go in unscaled.
//   mode 0: C = acc (+ bias[j]) (ReLU) (+ add[i][j]);  mode 1: C += acc;
//   mode 3: C + z * part_stride = acc of K-chunk z (partial products, summed in fixed order by reduce_partials_kernel:
//           no float atomics anywhere, results are bit-reproducible)
#define GEMM_SEGS 1      // (products of several segments run in the activation x weight kernel)
struct GemmArgs {
    const float *A[GEMM_SEGS], *B[GEMM_SEGS];
    const float *amax_a[GEMM_SEGS];   // per segment: bound of |A_s| (device scalar) or NULL = unscaled
    const float *amax_b;              // bound of |B| or NULL
    int nseg;
    float *C;
    const float *bias, *add;
    const float *gate;                // mode 0: C = 0 where gate[i][j] <= 0 (same layout as C), the ReLU's backward; or NULL
    float *colsum_part;               // A with contiguous rows only: colsum_part[z][i] = sum over chunk z of A(i, kk), or NULL
    float *amax_c;                    // raised to max |C written| (mode 0 / 1) or NULL
    int I, J, KK;
    long sai, sak, sbk, sbn;
    int ldc, ldadd;
    int mode, relu, kchunk;
    long part_stride;
    int launch_seq;                   // debug builds: number of the launch
};

#ifndef GBK
#define GBK 64      // K-extent of one LDS stage
#endif
#define LDK (GBK + 8)   // halfs per LDS row: 8 of padding (conflict-free 16-byte operand reads)
#define NG (GBK / 16)   // groups of four elements per thread, operand and stage
static_assert(GBK == 64, "the LDS block permutation of gemm_tile assumes eight 16-byte blocks per row");
#define GEMM_LDS_BYTES (4 * 64 * LDK * 2)

// power of two that brings a magnitude bound to [2^10, 2^11): exact scaling, 6 binades of headroom below the
// binary16 maximum, 24 binades of full 22-bit precision below the bound
__device__ __forceinline__ float pow2_scale_for(float amax) {
    if (!(amax > 0.0f) || !(amax < INFINITY)) return 1.0f;
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;
    e = max(min(10 - e, 60), -60);
    return __uint_as_float((unsigned)(e + 127) << 23);
}
// A magnitude bound is kept as AMAX_SUB partial maxima, 128 bytes apart: thousands of waves raise one bound, and on a
// single address their atomics (and the reads queued behind them) serialise in one L2 channel.  A workgroup raises
// the part chosen by its index; the plain read first keeps all but the few that really raise it away from the atomic
// unit (a stale read only costs an atomic).  v >= 0: unsigned order = numeric order.
#define AMAX_SUB 16
#define AMAX_SUB_STRIDE 32                          // floats between parts
#define AMAX_STRIDE (AMAX_SUB * AMAX_SUB_STRIDE)    // floats per bound
__device__ __forceinline__ void atomic_max_abs(float *bound, float v) {
    float *slot = bound + (blockIdx.x & (AMAX_SUB - 1)) * AMAX_SUB_STRIDE;
    if (v > *reinterpret_cast<volatile float *>(slot)) atomicMax(reinterpret_cast<unsigned *>(slot), __float_as_uint(v));
}
// The same with the read taken earlier (`seen`, by amax_peek): a load placed after a kernel's last stores waits for
// all of them (loads and stores leave one queue in order), and that wait was the last several thousand cycles of
// every workgroup; a value read before the stores is at worst staler, which costs an atomic, never a wrong bound.
__device__ __forceinline__ float amax_peek(const float *bound) {
    return *reinterpret_cast<const volatile float *>(bound + (blockIdx.x & (AMAX_SUB - 1)) * AMAX_SUB_STRIDE);
}
__device__ __forceinline__ void atomic_max_abs_seen(float *bound, float v, float seen) {
    if (v > seen) atomicMax(reinterpret_cast<unsigned *>(bound + (blockIdx.x & (AMAX_SUB - 1)) * AMAX_SUB_STRIDE), __float_as_uint(v));
}
__device__ __forceinline__ float amax_read(const float *bound) {
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < AMAX_SUB; ++i) m = fmaxf(m, bound[i * AMAX_SUB_STRIDE]);
    return m;
}

// four consecutive elements of a strided operand tile, along k (k_fast) or along the tile's row index m:
// one 16-byte load where the four exist and are aligned, guarded scalars otherwise
__device__ __forceinline__ float4 gemm_load4(const float *__restrict__ P, long sm, long sk, bool k_fast, bool vec_ok,
                                             int m, int k, int m_hi, int k_hi) {
    float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (m >= m_hi || k >= k_hi) return v;
    const float *p = P + (long)m * sm + (long)k * sk;
    const int left = k_fast ? k_hi - k : m_hi - m;
    const long step = k_fast ? sk : sm;                       // 1 on the fast index
    if (vec_ok && left >= 4) return *reinterpret_cast<const float4 *>(p);
    v.x = p[0];
    if (left > 1) v.y = p[step];
    if (left > 2) v.z = p[2 * step];
    if (left > 3) v.w = p[3 * step];
    return v;
}

// Debug build (-DGEMM_STAMPS): wave 0 of every workgroup records the shader clock at the phase boundaries of its
// tile; dgl_debug_gemm_stamps copies the records of the last 64 launches out.
#ifdef GEMM_STAMPS
static int gemm_launch_seq = 0;
#define GEMM_STAMP_WGS 4096
static __device__ unsigned long long gemm_stamp_buf[64 * GEMM_STAMP_WGS * 16];
#define GSTAMP(i)                                                                                                           \
    do {                                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < GEMM_STAMP_WGS)                                                                \
            gemm_stamp_buf[((size_t)(g.launch_seq & 63) * GEMM_STAMP_WGS + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GSTAMP(i)
#endif

// One 64x64 tile of one product.  AK / BK: the operand's k index is the contiguous one in memory (else its row index
// is).  FULL: the tile lies inside both operands, every stage is a whole one and every group of four is a 16-byte
// load - the sixteen loads of a stage are then issued back to back with nothing between them (a guard around each
// load would serialise them on the memory latency); the other tiles take guarded loads.
template <bool AK, bool BK, bool FULL>
__device__ __forceinline__ void gemm_tile(const GemmArgs &g, const int i0, const int j0, const int z, _Float16 *lds) {
    _Float16 *AH = lds, *AL = AH + 64 * LDK, *BH = AL + 64 * LDK, *BL = BH + 64 * LDK;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int k_lo = z * g.kchunk, k_hi = min(g.KK, k_lo + g.kchunk);
    const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
    float sa = 1.0f, sb = 1.0f;
    f32x16 acc = zero16();
    // staging map: each thread moves eight groups of four elements per operand and stage.  An operand whose k index
    // is contiguous in memory: four consecutive k of one row per group (32 groups per row, rows tid/32 + 8 e).  An
    // operand whose row index is contiguous: four consecutive rows at k = 32 (e/2) + 2 (tid/16) and at k + 1, so that
    // a (k, k+1) pair of every row splits into one packed word.  The next stage is fetched into registers while the
    // MFMAs of the current one run.
    // LDS offset of (row, k): the 16-byte blocks of a row are permuted by bits 3..5 of the row number.  The 16-byte
    // operand reads stay conflict-free (eight consecutive rows share a permutation and lie 9 blocks apart), and the
    // 4-byte stores of a row-contiguous operand - lanes four rows apart, i.e. 16 banks apart with this row length -
    // spread over all 32 banks instead of 8 (they were the busiest thing in a weight-gradient workgroup).
    auto lds_off = [](int row, int k) { return row * LDK + (k ^ (((row >> 3) & 7) << 3)); };
    auto map_m = [&](bool k_fast, int e) { return k_fast ? tid / (GBK / 4) + (1024 / GBK) * e : (tid & 15) * 4; };
    auto map_k = [&](bool k_fast, int e) { return k_fast ? (tid % (GBK / 4)) * 4 : 32 * (e >> 1) + 2 * (tid >> 4) + (e & 1); };
    // splits and stores one group (k-contiguous operand: v0) or one pair of groups (row-contiguous operand: v0 at k, v1 at k + 1)
    auto stage_group = [&](_Float16 *H, _Float16 *L, bool k_fast, int e, const float4 &v0, const float4 &v1, float s) {
        if (k_fast) {
            unsigned h0, l0, h1, l1;
            split2(v0.x * s, v0.y * s, h0, l0);
            split2(v0.z * s, v0.w * s, h1, l1);
            const int o = lds_off(map_m(true, e), map_k(true, e));
            *reinterpret_cast<u32x2 *>(H + o) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(L + o) = u32x2{l0, l1};
        } else {
            const float k0v[4] = {v0.x, v0.y, v0.z, v0.w}, k1v[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split2(k0v[c] * s, k1v[c] * s, hi, lo);
                const int o = lds_off(map_m(false, e) + c, map_k(false, e));
                *reinterpret_cast<unsigned *>(H + o) = hi;
                *reinterpret_cast<unsigned *>(L + o) = lo;
            }
        }
    };
    float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};                // running sums over k of the A rows this thread stages
    auto add_csum = [&](const float4 &v) { csum[0] += v.x; csum[1] += v.y; csum[2] += v.z; csum[3] += v.w; };
    const bool want_csum = !AK && g.colsum_part != nullptr;
    for (int seg = 0; seg < g.nseg; ++seg) {
        const float *A = g.A[seg], *B = g.B[seg];
        float4 ra[NG], rb[NG];                                 // FULL only: the next stage, in flight during the MFMAs
        auto fetch = [&](int k0) {
            const float *pa = A + (long)(i0 + map_m(AK, 0)) * g.sai + (long)(k0 + map_k(AK, 0)) * g.sak;
            const float *pb = B + (long)(j0 + map_m(BK, 0)) * g.sbn + (long)(k0 + map_k(BK, 0)) * g.sbk;
#pragma unroll
            for (int e = 0; e < NG; ++e)
                ra[e] = *reinterpret_cast<const float4 *>(pa + (long)(map_m(AK, e) - map_m(AK, 0)) * g.sai +
                                                          (long)(map_k(AK, e) - map_k(AK, 0)) * g.sak);
#pragma unroll
            for (int e = 0; e < NG; ++e)
                rb[e] = *reinterpret_cast<const float4 *>(pb + (long)(map_m(BK, e) - map_m(BK, 0)) * g.sbn +
                                                          (long)(map_k(BK, e) - map_k(BK, 0)) * g.sbk);
        };
        if (FULL) fetch(k_lo);
        if (seg == 0) {
            // operand scales (one per product: the segments share the larger bound), read while the first stage's
            // loads are in flight, and the values the epilogue combines with the product
            float bound_a = 0.0f;
            bool scaled_a = false;
            for (int s = 0; s < g.nseg; ++s)
                if (g.amax_a[s]) { bound_a = fmaxf(bound_a, amax_read(g.amax_a[s])); scaled_a = true; }
            if (scaled_a) sa = pow2_scale_for(bound_a);
            if (g.amax_b) sb = pow2_scale_for(amax_read(g.amax_b));
            GSTAMP(1);
        }
        for (int k0 = k_lo; k0 < k_hi; k0 += GBK) {
#ifdef GEMM_STAMPS
            const bool st2 = seg == 0 && k0 == k_lo + GBK;     // the second stage, phase by phase
            if (st2) { GSTAMP(8); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GSTAMP(9); }
#endif
            if (FULL) {
                if (want_csum) {
#pragma unroll
                    for (int e = 0; e < NG; ++e) add_csum(ra[e]);
                }
#pragma unroll
                for (int e = 0; e < NG; e += AK ? 1 : 2) stage_group(AH, AL, AK, e, ra[e], ra[AK ? e : e + 1], sa);
#pragma unroll
                for (int e = 0; e < NG; e += BK ? 1 : 2) stage_group(BH, BL, BK, e, rb[e], rb[BK ? e : e + 1], sb);
            } else {
                // edge tile: group by group, guarded, nothing kept in flight
                const bool a_vec = ((reinterpret_cast<size_t>(A) & 15) == 0) && (((AK ? g.sai : g.sak) & 3) == 0);
                const bool b_vec = ((reinterpret_cast<size_t>(B) & 15) == 0) && (((BK ? g.sbn : g.sbk) & 3) == 0);
#pragma unroll 1
                for (int e = 0; e < NG; e += AK ? 1 : 2) {
                    const float4 v0 = gemm_load4(A, g.sai, g.sak, AK, a_vec, i0 + map_m(AK, e), k0 + map_k(AK, e), g.I, k_hi);
                    const float4 v1 = AK ? v0 : gemm_load4(A, g.sai, g.sak, AK, a_vec, i0 + map_m(AK, e + 1), k0 + map_k(AK, e + 1), g.I, k_hi);
                    if (want_csum) { add_csum(v0); add_csum(v1); }
                    stage_group(AH, AL, AK, e, v0, v1, sa);
                }
#pragma unroll 1
                for (int e = 0; e < NG; e += BK ? 1 : 2) {
                    const float4 v0 = gemm_load4(B, g.sbn, g.sbk, BK, b_vec, j0 + map_m(BK, e), k0 + map_k(BK, e), g.J, k_hi);
                    const float4 v1 = BK ? v0 : gemm_load4(B, g.sbn, g.sbk, BK, b_vec, j0 + map_m(BK, e + 1), k0 + map_k(BK, e + 1), g.J, k_hi);
                    stage_group(BH, BL, BK, e, v0, v1, sb);
                }
            }
            if (k0 == k_lo && seg == 0) GSTAMP(2);
#ifdef GEMM_STAMPS
            if (st2) GSTAMP(10);
#endif
            __syncthreads();
            if (k0 == k_lo && seg == 0) GSTAMP(3);
#ifdef GEMM_STAMPS
            if (st2) GSTAMP(11);
#endif
            if (FULL && k0 + GBK < k_hi) fetch(k0 + GBK);
#ifdef GEMM_STAMPS
            if (st2) GSTAMP(12);
#endif
            const int ksteps = FULL ? GBK / 16 : (min(GBK, k_hi - k0) + 15) / 16;
#pragma unroll 2
            for (int t = 0; t < ksteps; ++t) {
                const int oa = lds_off(wi + q, 16 * t + 8 * h), ob = lds_off(wj + q, 16 * t + 8 * h);
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(AH + oa), al = *reinterpret_cast<const f16x8 *>(AL + oa);
                const f16x8 bh = *reinterpret_cast<const f16x8 *>(BH + ob), bl = *reinterpret_cast<const f16x8 *>(BL + ob);
                acc = MFMA16(al, bh, acc);
                acc = MFMA16(ah, bl, acc);
                acc = MFMA16(ah, bh, acc);
            }
#ifdef GEMM_STAMPS
            if (st2) GSTAMP(13);
#endif
            __syncthreads();
#ifdef GEMM_STAMPS
            if (st2) GSTAMP(14);
#endif
        }
    }
    GSTAMP(4);
    const float unscale = 1.0f / (sa * sb);                   // exact: powers of two within 2^+-120
    const int j = j0 + wj + q;
    float cmax = 0.0f;
    if (j < g.J) {
        float *Cz = g.C + (g.mode == 3 ? (long)z * g.part_stride : 0);
        // everything the epilogue reads is loaded before its first store: C, add and gate may alias as far as the
        // compiler can tell, and a load placed after a store would wait for its own round trip sixteen times over
        float other[16], gate[16];
        const float bias_j = (g.mode == 0 && g.bias) ? g.bias[j] : 0.0f;
        const float *src = g.mode == 1 ? g.C : g.add;
        const int lds_ = g.mode == 1 ? g.ldc : g.ldadd;
        const bool has_other = g.mode == 1 || (g.mode == 0 && g.add != nullptr), has_gate = g.mode == 0 && g.gate != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool in = FULL || i < g.I;
            other[r] = (has_other && in) ? src[(long)i * lds_ + j] : 0.0f;
            gate[r] = (has_gate && in) ? g.gate[(long)i * g.ldc + j] : 1.0f;
        }
        // (fields in registers before the first store: after a store through a float pointer the compiler re-reads
        // every field it needs from the argument segment, one scalar round trip each)
        const int ldc = g.ldc, rows = g.I;
        const bool plain = g.mode == 0, relu = g.relu != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!FULL && i >= rows) continue;
            float v = acc[r] * unscale;
            if (plain) {
                v += bias_j;
                if (relu) v = fmaxf(v, 0.0f);
                if (!(gate[r] > 0.0f)) v = 0.0f;
            }
            v = other[r] + v;
            Cz[(long)i * ldc + j] = v;
            cmax = fmaxf(cmax, fabsf(v));
        }
    }
    GSTAMP(5);
    if (g.amax_c) {
        cmax = wave_max_f32(cmax);
        if (lane == 0) atomic_max_abs(g.amax_c, cmax);
    }
    GSTAMP(6);
    if (!AK && g.colsum_part && j0 == 0) {
        // sum over this chunk's k of every A row: the 16 threads that staged a row add up in a fixed order
        float *red = reinterpret_cast<float *>(lds);             // [16][64]; the last stage's reads ended at the barrier
#pragma unroll
        for (int c = 0; c < 4; ++c) red[(tid >> 4) * 64 + (tid & 15) * 4 + c] = csum[c];
        __syncthreads();
        if (tid < 64 && i0 + tid < g.I) {
            float s = 0.0f;
#pragma unroll
            for (int o = 0; o < 16; ++o) s += red[o * 64 + tid];
            g.colsum_part[(long)z * g.I + i0 + tid] = s;
        }
    }
}

// Several independent products share one launch: workgroup `blockIdx.x` belongs to the product whose block
// range [first[b], first[b+1]) contains it.  Consecutive workgroups go to the chip's eight dies in turn, each with
// its own L2, so the tiles that read the same operand data are numbered eight apart: the 64-column tiles of one
// row tile (they share the A tile) when the reduction is short, all tiles of one K-chunk when it is split.
#define GEMM_BATCH_MAX 6
struct GemmBatch {
    GemmArgs g[GEMM_BATCH_MAX];
    int first[GEMM_BATCH_MAX + 1];
    int count;
};
static inline __host__ __device__ int gemm_round8(int v) { return (v + 7) & ~7; }
static inline __host__ __device__ int gemm_blocks(const GemmArgs &g) {
    const int gx = (g.I + 63) / 64, gy = (g.J + 63) / 64, zs = (g.KK + g.kchunk - 1) / g.kchunk;
    return zs > 1 ? gx * gy * gemm_round8(zs) : gemm_round8(gx) * gy;
}

__device__ __forceinline__ void gemm_batch_block(const GemmBatch &batch, const int block) {
    extern __shared__ __align__(16) _Float16 gemm_lds[];
    int b = 0;
    while (b + 1 < batch.count && block >= batch.first[b + 1]) ++b;
    const GemmArgs &g = batch.g[b];
    GSTAMP(0);
    const int local = block - batch.first[b];
    const int gx = (g.I + 63) / 64, gy = (g.J + 63) / 64, zs = (g.KK + g.kchunk - 1) / g.kchunk;
    int ti, tj, z;
    if (zs > 1) {
        const int z8 = gemm_round8(zs);
        z = local % z8;
        ti = (local / z8) % gx;
        tj = local / (z8 * gx);
    } else {
        z = 0;
        tj = (local % (8 * gy)) / 8;
        ti = (local / (8 * gy)) * 8 + (local & 7);
    }
    if (z >= zs || ti >= gx) return;
#ifdef GEMM_SAME_TILE
    ti = tj = z = 0;                                         // (timing experiment: every workgroup works on one cache-resident tile)
#endif
    const int i0 = ti * 64, j0 = tj * 64;
    const int k_lo = z * g.kchunk, k_len = min(g.KK, k_lo + g.kchunk) - k_lo;
    const bool ak = g.sak == 1, bk = g.sbk == 1;
    bool full = i0 + 64 <= g.I && j0 + 64 <= g.J && (k_len % GBK) == 0 && (((ak ? g.sai : g.sak) | (bk ? g.sbn : g.sbk)) & 3) == 0;
    for (int s = 0; s < g.nseg; ++s) full = full && ((reinterpret_cast<size_t>(g.A[s]) | reinterpret_cast<size_t>(g.B[s])) & 15) == 0;
    if (ak && bk) {
        if (full) gemm_tile<true, true, true>(g, i0, j0, z, gemm_lds);
        else gemm_tile<true, true, false>(g, i0, j0, z, gemm_lds);
    } else if (ak) {
        if (full) gemm_tile<true, false, true>(g, i0, j0, z, gemm_lds);
        else gemm_tile<true, false, false>(g, i0, j0, z, gemm_lds);
    } else {
        if (full) gemm_tile<false, false, true>(g, i0, j0, z, gemm_lds);
        else gemm_tile<false, false, false>(g, i0, j0, z, gemm_lds);
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch batch) { gemm_batch_block(batch, blockIdx.x); }

struct GemmLaunch {
    GemmBatch b;
    int blocks = 0;
    GemmLaunch() { b.count = 0; }
    GemmLaunch &add(const GemmArgs &g) {
        b.g[b.count] = g;
        b.first[b.count++] = blocks;
        blocks += gemm_blocks(g);
        return *this;
    }
    void run(hipStream_t s) {
        b.first[b.count] = blocks;
#ifdef GEMM_STAMPS
        for (int i = 0; i < b.count; ++i) b.g[i].launch_seq = gemm_launch_seq;
        ++gemm_launch_seq;
#endif
        dgl_max_dynamic_lds((const void *)gemm_kernel, GEMM_LDS_BYTES);
        hipLaunchKernelGGL(gemm_kernel, dim3(blocks), dim3(256), GEMM_LDS_BYTES, s, b);
    }
};
#ifdef GEMM_STAMPS
extern "C" __attribute__((visibility("default"))) int dgl_debug_gemm_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm_stamp_buf), sizeof(unsigned long long) * 64 * GEMM_STAMP_WGS * 16);
}
extern "C" __attribute__((visibility("default"))) int dgl_debug_gemm_stamps_clear() {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(gemm_stamp_buf)) != hipSuccess) return 1;
    gemm_launch_seq = 0;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 64 * GEMM_STAMP_WGS * 16);
}
#endif
static GemmArgs gemm_args(const float *A, const float *amax_a, const float *B, const float *amax_b, float *C, float *amax_c) {
    GemmArgs g{};
    g.A[0] = A; g.B[0] = B; g.amax_a[0] = amax_a; g.amax_b = amax_b; g.nseg = 1; g.C = C; g.amax_c = amax_c;
    return g;
}
// Y[M][N] = X[M][K] W[N][K]^T + b (ReLU) (+ add)
static GemmArgs linear_fwd(const float *X, const float *amax_x, int ldx, const float *W, const float *b, float *Y, float *amax_y,
                           int ldy, int M, int N, int K, bool relu, const float *add, int ldadd) {
    GemmArgs g = gemm_args(X, amax_x, W, nullptr, Y, amax_y);
    g.bias = b; g.add = add; g.I = M; g.J = N; g.KK = K; g.sai = ldx; g.sak = 1; g.sbk = 1; g.sbn = K; g.ldc = ldy;
    g.ldadd = ldadd; g.mode = 0; g.relu = relu ? 1 : 0; g.kchunk = K;
    return g;
}
// dX[M][K] (+)= dY[M][N] W[N][K]; mode 0 store, 1 add in place (no other writer of dX in flight)
static GemmArgs linear_bwd_x(const float *dY, const float *amax_dy, int ldy, const float *W, float *dX, float *amax_dx, int ldx,
                             int M, int N, int K, int mode) {
    GemmArgs g = gemm_args(dY, amax_dy, W, nullptr, dX, amax_dx);
    g.I = M; g.J = K; g.KK = N; g.sai = ldy; g.sak = 1; g.sbk = K; g.sbn = 1; g.ldc = ldx; g.mode = mode; g.kchunk = N;
    return g;
}
// partial products of dW[N][K] = dY[M][N]^T X[M][K] over chunks of 512 rows of M: chunk z goes to part + z N K, and
// the chunk's term of db[N] = sum over rows of dY to bias_part + z N (when given)
#define DW_CHUNK 512
static GemmArgs linear_bwd_w(const float *dY, const float *amax_dy, long row_stride_dy, const float *X, const float *amax_x,
                             long row_stride_x, float *part, float *bias_part, int M, int N, int K) {
    GemmArgs g = gemm_args(dY, amax_dy, X, amax_x, part, nullptr);
    g.colsum_part = bias_part;
    g.I = N; g.J = K; g.KK = M; g.sai = 1; g.sak = row_stride_dy; g.sbk = row_stride_x; g.sbn = 1; g.ldc = K; g.mode = 3;
    g.kchunk = DW_CHUNK; g.part_stride = (long)N * K;
    return g;
}

// ------------------------------------------------------------ activation x weight GEMM
// Y[M][N] (op)= sum over segments s of A_s[M][K] B_s, B_s a weight matrix (or its transpose) of the model, K and N
// multiples of 128: every product of the forward pass and every dX product of the backward pass.
//
// The weights are split into binary16 halves once per call, in the order in which the matrix cores read them
// (pack_weights_kernel): the B operand of 32 output columns and one k-step is 1 KB of hi halves and 1 KB of lo
// halves, lane after lane, so a wave reads its operands straight from memory (L2-resident) into registers, 16 KB of
// consecutive addresses per eight k-steps, double-buffered - no LDS, no conversion, no barrier in the loop.  A
// workgroup owns 32 rows of A, converts them ONCE into an LDS image and sweeps all N output columns with it (wave w
// takes the 32-column tiles w, w + 4, ...); several products with the same A (Q, K, V) share one image, and the
// segments of one product (dQ Wq + dK Wk + dV Wv) accumulate in the same tile.
//   packed B: [N / 32][Kd / 16][hi, lo][64 lanes][8 halfs],  lane (q, h) slot e  <->  B(16 t + 8 h + e, 32 jt + q)
#define AW_SEGS 3
struct AwProduct {
    const _Float16 *Bp[AW_SEGS];
    float *C;
    const float *bias, *add, *gate;
    float *amax_c;
    int ldc, ldadd, mode, relu;          // mode 0: C = acc (+ bias) (ReLU) (gate) (+ add);  mode 1: C += acc
};
struct AwArgs {
    const float *A[AW_SEGS];
    const float *amax_a[AW_SEGS];
    int nseg, lda, M, N;                 // every product has N columns
    AwProduct p[3];
    int nprod;
    int launch_seq;                      // debug builds: number of the launch
};
#ifdef GEMM_STAMPS
#define ASTAMP(i)                                                                                                           \
    do {                                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < GEMM_STAMP_WGS)                                                                \
            gemm_stamp_buf[((size_t)(a.launch_seq & 63) * GEMM_STAMP_WGS + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ASTAMP(i)
#endif

// B(k, j) = W[j * ld + k] (transposed == 0: Y = X W^T) or W[k * ld + j] (transposed: dX = dY W), Kd x Jd
struct PackItem {
    const float *W;
    _Float16 *dst;
    int Kd, Jd, ld, transposed;
};
#define PACK_ITEMS_MAX 48
struct PackBatch {
    PackItem it[PACK_ITEMS_MAX];
    int first[PACK_ITEMS_MAX + 1];       // first workgroup of each item
    int count;
};
__global__ __launch_bounds__(256) void pack_weights_kernel(PackBatch batch, float *__restrict__ zero, int zero_count) {
    // (the call's first launch also clears the magnitude bounds)
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid < zero_count) zero[gid] = 0.0f;
    int b = 0;
    while (b + 1 < batch.count && (int)blockIdx.x >= batch.first[b + 1]) ++b;
    const PackItem &it = batch.it[b];
    const int idx = (blockIdx.x - batch.first[b]) * 256 + threadIdx.x;       // (jt, t, lane)
    const int ksteps = it.Kd / 16;
    if (idx >= (it.Jd / 32) * ksteps * 64) return;
    const int lane = idx & 63, t = (idx >> 6) % ksteps, jt = (idx >> 6) / ksteps, q = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * t + 8 * h + e, j = 32 * jt + q;
        v[e] = it.transposed ? it.W[(long)k * it.ld + j] : it.W[(long)j * it.ld + k];
    }
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, c;
        split2(v[2 * e], v[2 * e + 1], a, c);
        hi[e] = a; lo[e] = c;
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(it.dst) + ((long)(jt * ksteps + t) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

template <int K>
__device__ __forceinline__ void gemm_aw_block(const AwArgs &a, const int block) {
    extern __shared__ __align__(16) _Float16 aw_lds[];       // [nseg][hi, lo][32][K + 8]
    constexpr int LDA = K + 8, CH = K / 128, NL = (32 * K / 4) / 256;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int i0 = block * 32;
    ASTAMP(0);
    const int tpp = a.N / 128;                                // tiles per wave and product
    const int G = a.nprod * tpp * a.nseg * CH;                // chunks of eight k-steps this wave walks through
    // the wave's B operands of chunk g: 16 KB of consecutive addresses
    auto chunk_ptr = [&](int g) {
        const int c = g % CH, seg = (g / CH) % a.nseg, tile = g / (CH * a.nseg);
        const int jt = w + 4 * (tile % tpp);
#ifdef AW_SAME_CHUNK
        return reinterpret_cast<const u32x4 *>(a.p[0].Bp[0]) + lane;     // (timing experiment: one L1-resident chunk for every wave)
#endif
        return reinterpret_cast<const u32x4 *>(a.p[tile / tpp].Bp[seg]) + ((long)(jt * (K / 16) + 8 * c) * 2) * 64 + lane;
    };
    u32x4 b0[16], b1[16];
    auto load_chunk = [&](u32x4 (&b)[16], int g) {
        const u32x4 *p = chunk_ptr(g);
#pragma unroll
        for (int e = 0; e < 16; ++e) b[e] = p[64 * e];
    };
    // ---- the A rows: loads, then the first B chunk, then the scale, split and the LDS image
    float sa = 1.0f;
    {
        float4 v[AW_SEGS > 1 && K == 128 ? AW_SEGS * NL : NL];
        const int nsegs = K == 128 ? a.nseg : 1;
#pragma unroll
        for (int seg = 0; seg < (K == 128 ? AW_SEGS : 1); ++seg) {
            if (seg < nsegs) {
#pragma unroll
                for (int e = 0; e < NL; ++e) {
                    const int idx = tid + 256 * e, row = idx / (K / 4), c4 = idx % (K / 4);
                    const int rr = min(i0 + row, a.M - 1);            // (rows beyond M: a valid address, scaled by zero below)
                    v[seg * NL + e] = *reinterpret_cast<const float4 *>(a.A[seg] + (long)rr * a.lda + 4 * c4);
                }
            }
        }
        load_chunk(b0, 0);
        float bound = 0.0f;
        bool scaled = false;
        for (int seg = 0; seg < nsegs; ++seg)
            if (a.amax_a[seg]) { bound = fmaxf(bound, amax_read(a.amax_a[seg])); scaled = true; }
        if (scaled) sa = pow2_scale_for(bound);
        ASTAMP(1);
#ifdef GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ASTAMP(2);
#endif
#pragma unroll
        for (int seg = 0; seg < (K == 128 ? AW_SEGS : 1); ++seg) {
            if (seg < nsegs) {
                _Float16 *H = aw_lds + seg * 2 * 32 * LDA, *L = H + 32 * LDA;
#pragma unroll
                for (int e = 0; e < NL; ++e) {
                    const int idx = tid + 256 * e, row = idx / (K / 4), c4 = idx % (K / 4);
                    const float4 x = v[seg * NL + e];
                    const float sr = i0 + row < a.M ? sa : 0.0f;
                    unsigned h0, l0, h1, l1;
                    split2(x.x * sr, x.y * sr, h0, l0);
                    split2(x.z * sr, x.w * sr, h1, l1);
                    *reinterpret_cast<u32x2 *>(H + row * LDA + 4 * c4) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2 *>(L + row * LDA + 4 * c4) = u32x2{l0, l1};
                }
            }
        }
    }
    ASTAMP(3);
    __syncthreads();
    ASTAMP(4);
    const float unscale = 1.0f / sa;
    f32x16 acc = zero16();
    // The products are evaluated transposed - the weight fragment as the A operand, the activation rows as B - so that
    // in the accumulator layout (lane = column) a lane holds 16 output features of ONE token, four consecutive ones
    // per register group: the epilogue reads and writes 16 bytes per lane and instruction (4 stores per tile instead
    // of 16, and as many fewer loads of the residual and the gate).  Same products, same order of summation.
    const int row = i0 + q;                                     // this lane's token
    const bool row_ok = row < a.M;
    const long row_c = min(row, a.M - 1);                       // (a valid address for the loads of rows beyond M)
    float4 other[4], gate[4], bias4[4];
    float pmax[3] = {0.0f, 0.0f, 0.0f}, seen[3] = {0.0f, 0.0f, 0.0f};
    // one chunk: eight k-steps from registers `b`, the next chunk's loads issued first
    auto chunk = [&](int g, u32x4 (&b)[16], u32x4 (&bn)[16]) {
        if (g + 1 < G) load_chunk(bn, g + 1);
        const int c = g % CH, seg = (g / CH) % a.nseg, tile = g / (CH * a.nseg);
        const AwProduct P = a.p[tile / tpp];          // (a copy in registers: a reference is re-read field by field after every store)
        const int cb = 32 * (w + 4 * (tile % tpp)) + 4 * h;     // this lane's columns: cb + 8 g + {0..3}, g = 0..3
        const bool last = c == CH - 1 && seg == a.nseg - 1;
        if (c == 0 && seg == 0) acc = zero16();
        if (last) {
            // what the epilogue combines with the product, in flight during the last MFMAs (and all of it before the
            // first store: C, add and gate may alias as far as the compiler can tell)
            const float4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f}, one4 = {1.0f, 1.0f, 1.0f, 1.0f};
            if (tile % tpp == tpp - 1 && P.amax_c != nullptr) {
                // the product's last tile of this wave: the bound as it stands, read before the tile's stores
                const float sv = amax_peek(P.amax_c);
#pragma unroll
                for (int pi = 0; pi < 3; ++pi)
                    if (pi == tile / tpp) seen[pi] = sv;
            }
            const bool has_bias = P.mode == 0 && P.bias != nullptr;
            if (has_bias && (reinterpret_cast<size_t>(P.bias) & 15) == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias4[e] = *reinterpret_cast<const float4 *>(P.bias + cb + 8 * e);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *pb = P.bias + cb + 8 * e;
                    bias4[e] = has_bias ? float4{pb[0], pb[1], pb[2], pb[3]} : zero4;
                }
            }
            const float *src = P.mode == 1 ? P.C : P.add;
            const int lds_ = P.mode == 1 ? P.ldc : P.ldadd;
            const bool has_other = P.mode == 1 || P.add != nullptr, has_gate = P.mode == 0 && P.gate != nullptr;
            // (uniform branches: a guard per lane would turn every load into a branch of its own)
            if (has_other) {
#pragma unroll
                for (int e = 0; e < 4; ++e) other[e] = *reinterpret_cast<const float4 *>(src + row_c * lds_ + cb + 8 * e);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) other[e] = zero4;
            }
            if (has_gate) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gate[e] = *reinterpret_cast<const float4 *>(P.gate + row_c * P.ldc + cb + 8 * e);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) gate[e] = one4;
            }
        }
        const _Float16 *Ah = aw_lds + seg * 2 * 32 * LDA + q * LDA + 128 * c + 8 * h, *Al = Ah + 32 * LDA;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(Ah + 16 * t), al = *reinterpret_cast<const f16x8 *>(Al + 16 * t);
            const f16x8 bh = as_f16x8(b[2 * t]), bl = as_f16x8(b[2 * t + 1]);
            acc = MFMA16(bh, al, acc);
            acc = MFMA16(bl, ah, acc);
            acc = MFMA16(bh, ah, acc);
        }
#ifdef GEMM_STAMPS
        if (last && tile < 4) ASTAMP(5 + 2 * tile);
#endif
        if (last) {
            // (the product's fields in registers before the first store: after a store through a float pointer the
            // compiler re-reads every field it needs from the argument segment, one scalar round trip each)
            float *const C = P.C;
            const int ldc = P.ldc;
            const bool plain = P.mode == 0, relu = P.relu != 0;
            float cmax = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ov[4] = {other[e].x, other[e].y, other[e].z, other[e].w};
                const float gv[4] = {gate[e].x, gate[e].y, gate[e].z, gate[e].w};
                const float bv[4] = {bias4[e].x, bias4[e].y, bias4[e].z, bias4[e].w};
                float o[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v = acc[4 * e + m] * unscale;
                    if (plain) {
                        v += bv[m];
                        if (relu) v = fmaxf(v, 0.0f);
                        if (!(gv[m] > 0.0f)) v = 0.0f;
                    }
                    v = ov[m] + v;
                    o[m] = v;
                    cmax = fmaxf(cmax, fabsf(v));
                }
                if (row_ok) *reinterpret_cast<float4 *>(C + (long)row * ldc + cb + 8 * e) = float4{o[0], o[1], o[2], o[3]};
            }
            if (!row_ok) cmax = 0.0f;
#pragma unroll
            for (int pi = 0; pi < 3; ++pi)
                if (pi == tile / tpp) pmax[pi] = fmaxf(pmax[pi], cmax);
#ifdef GEMM_STAMPS
            if (tile < 4) ASTAMP(6 + 2 * tile);
#endif
        }
    };
    for (int g = 0; g < G; g += 2) {
        chunk(g, b0, b1);
        if (g + 1 < G) chunk(g + 1, b1, b0);
    }
    // the products' magnitude bounds, once per wave, against the value read before the last tile's stores
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        if (pi < a.nprod && a.p[pi].amax_c) {
            const float m = wave_max_f32(pmax[pi]);
            if (lane == 0) atomic_max_abs_seen(a.p[pi].amax_c, m, seen[pi]);
        }
    }
    ASTAMP(13);
}
template <int K>
__global__ __launch_bounds__(256) void gemm_aw_kernel(AwArgs a) { gemm_aw_block<K>(a, blockIdx.x); }
// A batch of weight-gradient products and one activation x weight product that do not depend on each other, in one
// launch (a launch costs several microseconds whatever it does): the first aw_first workgroups take the batch.
template <int K>
__global__ __launch_bounds__(256) void gemm_pair_kernel(GemmBatch batch, AwArgs a, int aw_first) {
    if ((int)blockIdx.x < aw_first) gemm_batch_block(batch, blockIdx.x);
    else gemm_aw_block<K>(a, blockIdx.x - aw_first);
}
static size_t aw_lds_bytes(int K, int nseg) { return (size_t)nseg * 2 * 32 * (K + 8) * 2; }
static void launch_aw(hipStream_t s, AwArgs a, int K) {
    const int blocks = (a.M + 31) / 32;
#ifdef GEMM_STAMPS
    a.launch_seq = gemm_launch_seq++;
#endif
    if (K == 128) {
        hipLaunchKernelGGL(gemm_aw_kernel<128>, dim3(blocks), dim3(256), aw_lds_bytes(128, a.nseg), s, a);
    } else {
        dgl_max_dynamic_lds((const void *)gemm_aw_kernel<512>, (int)aw_lds_bytes(512, 1));
        hipLaunchKernelGGL(gemm_aw_kernel<512>, dim3(blocks), dim3(256), aw_lds_bytes(512, 1), s, a);
    }
}

static void launch_pair(hipStream_t s, GemmLaunch &gl, AwArgs a, int K) {
    gl.b.first[gl.b.count] = gl.blocks;
    const int aw_blocks = (a.M + 31) / 32;
    const size_t lds = aw_lds_bytes(K, a.nseg) > (size_t)GEMM_LDS_BYTES ? aw_lds_bytes(K, a.nseg) : (size_t)GEMM_LDS_BYTES;
#ifdef GEMM_STAMPS
    for (int i = 0; i < gl.b.count; ++i) gl.b.g[i].launch_seq = gemm_launch_seq;
    a.launch_seq = gemm_launch_seq++;
#endif
    if (K == 128) {
        dgl_max_dynamic_lds((const void *)gemm_pair_kernel<128>, (int)aw_lds_bytes(128, 3));
        hipLaunchKernelGGL(gemm_pair_kernel<128>, dim3(gl.blocks + aw_blocks), dim3(256), lds, s, gl.b, a, gl.blocks);
    } else {
        dgl_max_dynamic_lds((const void *)gemm_pair_kernel<512>, (int)aw_lds_bytes(512, 1));
        hipLaunchKernelGGL(gemm_pair_kernel<512>, dim3(gl.blocks + aw_blocks), dim3(256), lds, s, gl.b, a, gl.blocks);
    }
}

// ------------------------------------------------------------ a block after its attention, forward pass, one launch
// O1 = Xin + Wc AO + bc;  Y = O1 + W2 relu(W1 O1 + b1) + b2  for 32 tokens per workgroup (TSP/TSPModel.py:380-405,
// CVRP/CVRPModel.py:565-590): the three products of gemm_aw_kernel chained inside the workgroup, O1 and the hidden
// layer passing through LDS instead of memory.  The bound of a whole tensor is not known before every workgroup has
// finished, so an intermediate operand is scaled by the power of two that suits the largest of its values among these
// 32 tokens, and the sum is scaled back exactly.  The hidden layer is produced in two halves of 256 features; each
// half is split into an LDS image and consumed at once as 16 k-steps of the last product.  O1 and the hidden values
// go to memory only when the caller keeps them for the backward pass; the residual O1 stays in registers (the wave
// that owns output tile w of the last product owns tile w of O1).
//   chunks of weight operands a wave walks through (16 KB each, double-buffered in registers):
//   Wc tile w | W1 tiles w, w + 4 | W2 tile w, k-steps 0-15 | W1 tiles w + 8, w + 12 | W2 tile w, k-steps 16-31
struct FfArgs {
    const float *AO, *amax_ao;          // attention output [M][128] and the bound of its magnitude
    const float *Xin;                   // the block's input [M][128]
    const _Float16 *Wcp, *W1p, *W2p;    // packed as for gemm_aw_kernel: Wc (K = 128, 128 columns), W1 (K = 128, 512), W2 (K = 512, 128)
    const float *bc, *b1, *b2;
    float *O1, *amax_o1;                // [M][128] and its bound, or both NULL (not kept)
    float *H, *amax_h;                  // hidden layer [M][512] and its bound, or both NULL
    float *Y, *amax_y;
    // the next block's Q = Y Wq, K = Y Wk, V = Y Wv in the same launch (Y is at hand), or Wqp NULL
    const _Float16 *Wqp, *Wkp, *Wvp;    // packed, K = 128, 128 columns each
    float *Q, *K, *V, *amax_qkv;        // [M][128] each; their three bounds, AMAX_STRIDE apart
    int M;
};
#define FF_LD1 (D_MODEL + 8)
#define FF_LD2 (FF_DIM / 2 + 8)
#define FF_BIAS_FLOATS (FF_DIM + 2 * D_MODEL)
#define FF_LDS_BYTES (2 * 32 * FF_LD1 * 2 + 2 * 32 * FF_LD2 * 2 + FF_BIAS_FLOATS * 4 + 16)
__global__ __launch_bounds__(256, 2) void ff_fused_kernel(FfArgs a) {
    extern __shared__ __align__(16) _Float16 ff_lds[];
    _Float16 *X1h = ff_lds, *X1l = X1h + 32 * FF_LD1, *H2h = X1l + 32 * FF_LD1, *H2l = H2h + 32 * FF_LD2;
    float *bias = reinterpret_cast<float *>(H2l + 32 * FF_LD2);               // b1 [512], b2 [128], bc [128]
    unsigned *wg_max = reinterpret_cast<unsigned *>(bias + FF_BIAS_FLOATS);   // largest value of: hidden half 0, 1, O1, Y
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32;
    u32x4 b0[16], b1r[16];
    auto load16 = [&](u32x4 (&b)[16], const u32x4 *p) {
#pragma unroll
        for (int e = 0; e < 16; ++e) b[e] = p[64 * e];
    };
    auto w1_chunk = [&](int tile) { return reinterpret_cast<const u32x4 *>(a.W1p) + (long)(tile * (D_MODEL / 16)) * 2 * 64 + lane; };
    auto w2_chunk = [&](int kc) { return reinterpret_cast<const u32x4 *>(a.W2p) + (long)(w * (FF_DIM / 16) + 8 * kc) * 2 * 64 + lane; };
    auto qkv_chunk = [&](const _Float16 *Wp) { return reinterpret_cast<const u32x4 *>(Wp) + (long)(w * (D_MODEL / 16)) * 2 * 64 + lane; };
    const bool qkv = a.Wqp != nullptr;
    // eight k-steps: weight fragments from registers (A operand), the tokens' image rows from LDS (B operand)
    auto steps8 = [&](const u32x4 (&b)[16], const _Float16 *Th, const _Float16 *Tl, f32x16 acc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 th = *reinterpret_cast<const f16x8 *>(Th + 16 * t), tl = *reinterpret_cast<const f16x8 *>(Tl + 16 * t);
            const f16x8 bh = as_f16x8(b[2 * t]), bl = as_f16x8(b[2 * t + 1]);
            acc = MFMA16(bh, tl, acc);
            acc = MFMA16(bl, th, acc);
            acc = MFMA16(bh, th, acc);
        }
        return acc;
    };
    float sa;
    {
        float4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, row = idx >> 5, c4 = idx & 31;
            const int rr = min(i0 + row, a.M - 1);                        // (rows beyond M: a valid address, scaled by zero below)
            v[e] = *reinterpret_cast<const float4 *>(a.AO + (long)rr * D_MODEL + 4 * c4);
        }
        load16(b0, reinterpret_cast<const u32x4 *>(a.Wcp) + (long)(w * (D_MODEL / 16)) * 2 * 64 + lane);
        for (int i = tid; i < FF_BIAS_FLOATS; i += 256)
            bias[i] = i < FF_DIM ? a.b1[i] : i < FF_DIM + D_MODEL ? a.b2[i - FF_DIM] : a.bc[i - FF_DIM - D_MODEL];
        if (tid < 4) wg_max[tid] = 0u;
        sa = pow2_scale_for(amax_read(a.amax_ao));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, row = idx >> 5, c4 = idx & 31;
            const float sr = i0 + row < a.M ? sa : 0.0f;
            unsigned h0, l0, h1, l1;
            split2(v[e].x * sr, v[e].y * sr, h0, l0);
            split2(v[e].z * sr, v[e].w * sr, h1, l1);
            *reinterpret_cast<u32x2 *>(X1h + row * FF_LD1 + 4 * c4) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(X1l + row * FF_LD1 + 4 * c4) = u32x2{l0, l1};
        }
    }
    __syncthreads();
    const int row = i0 + q;                                               // this lane's token
    const bool row_ok = row < a.M;
    const long row_c = min(row, a.M - 1);
    float *const O1out = a.O1;
    float *const Hout = a.H;
    float *const Yout = a.Y;
    const _Float16 *X1hq = X1h + q * FF_LD1 + 8 * h, *X1lq = X1l + q * FF_LD1 + 8 * h;
    const _Float16 *H2hq = H2h + q * FF_LD2 + 8 * h, *H2lq = H2l + q * FF_LD2 + 8 * h;
    const int cbw = 32 * w + 4 * h;                                       // this lane's features of a 128-wide tensor: cbw + 8 g + {0..3}
    // ---- O1 = Xin + Wc AO + bc: kept in registers as the last product's residual
    float resid[16];
    float o1max = 0.0f;
    // (each bound as it stands is read before the stores of the tensor it belongs to: see amax_peek)
    const float seen_o1 = a.amax_o1 ? amax_peek(a.amax_o1) : 0.0f;
    float seen_h = 0.0f;
    {
        float4 xin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = *reinterpret_cast<const float4 *>(a.Xin + row_c * D_MODEL + cbw + 8 * g);
        load16(b1r, w1_chunk(w));
        const f32x16 acc = steps8(b0, X1hq, X1lq, zero16());
        load16(b0, w1_chunk(w + 4));
        const float unscale_a = 1.0f / sa;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + FF_DIM + D_MODEL + cbw + 8 * g);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w}, xv[4] = {xin[g].x, xin[g].y, xin[g].z, xin[g].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float v = acc[4 * g + m] * unscale_a;
                v += bv[m];
                v = xv[m] + v;
                v = row_ok ? v : 0.0f;
                resid[4 * g + m] = v;
                o1max = fmaxf(o1max, fabsf(v));
            }
            if (O1out != nullptr && row_ok)
                *reinterpret_cast<float4 *>(O1out + (long)row * D_MODEL + cbw + 8 * g) =
                    float4{resid[4 * g], resid[4 * g + 1], resid[4 * g + 2], resid[4 * g + 3]};
        }
        o1max = wave_max_f32(o1max);
        if (lane == 0) atomicMax(&wg_max[2], __float_as_uint(o1max));
    }
    __syncthreads();                           // (every wave has read the attention output's image: O1's image takes its place)
    const float s1 = pow2_scale_for(__uint_as_float(wg_max[2]));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned h0, l0, h1, l1;
        split2(resid[4 * g] * s1, resid[4 * g + 1] * s1, h0, l0);
        split2(resid[4 * g + 2] * s1, resid[4 * g + 3] * s1, h1, l1);
        *reinterpret_cast<u32x2 *>(X1h + q * FF_LD1 + cbw + 8 * g) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(X1l + q * FF_LD1 + cbw + 8 * g) = u32x2{l0, l1};
    }
    __syncthreads();
    const float unscale_1 = 1.0f / s1;
    f32x16 y = zero16();
    float hmax = 0.0f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        // hidden features 256 hf + 32 (w + 4 j) + ..., j = 0, 1
        f32x16 t[2];
        t[0] = steps8(b1r, X1hq, X1lq, zero16());
        load16(b1r, w2_chunk(2 * hf));
        t[1] = steps8(b0, X1hq, X1lq, zero16());
        load16(b0, w2_chunk(2 * hf + 1));
        if (hf == 1 && a.amax_h) seen_h = amax_peek(a.amax_h);
        float lmax = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cb = 256 * hf + 32 * (w + 4 * j) + 4 * h;           // this lane's features: cb + 8 g + {0..3}
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bias + cb + 8 * g);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
                float o[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v = t[j][4 * g + m] * unscale_1;
                    v += bv[m];
                    v = fmaxf(v, 0.0f);
                    o[m] = row_ok ? v : 0.0f;
                    lmax = fmaxf(lmax, o[m]);
                    t[j][4 * g + m] = o[m];
                }
                if (Hout != nullptr && row_ok) *reinterpret_cast<float4 *>(Hout + (long)row * FF_DIM + cb + 8 * g) = float4{o[0], o[1], o[2], o[3]};
            }
        }
        lmax = wave_max_f32(lmax);
        hmax = fmaxf(hmax, lmax);
        if (lane == 0) atomicMax(&wg_max[hf], __float_as_uint(lmax));
        __syncthreads();                       // (every wave is past the previous half's last product: the image is free)
        const float sh = pow2_scale_for(__uint_as_float(wg_max[hf]));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h0, l0, h1, l1;
                split2(t[j][4 * g] * sh, t[j][4 * g + 1] * sh, h0, l0);
                split2(t[j][4 * g + 2] * sh, t[j][4 * g + 3] * sh, h1, l1);
                const int lc = 32 * (w + 4 * j) + 8 * g + 4 * h;
                *reinterpret_cast<u32x2 *>(H2h + q * FF_LD2 + lc) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(H2l + q * FF_LD2 + lc) = u32x2{l0, l1};
            }
        }
        __syncthreads();
        f32x16 part = steps8(b1r, H2hq, H2lq, zero16());
        if (hf == 0) load16(b1r, w1_chunk(w + 8));
        else if (qkv) load16(b1r, qkv_chunk(a.Wqp));
        part = steps8(b0, H2hq + 128, H2lq + 128, part);
        if (hf == 0) load16(b0, w1_chunk(w + 12));
        else if (qkv) load16(b0, qkv_chunk(a.Wkp));
        const float unscale_h = 1.0f / sh;
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] += part[e] * unscale_h;
    }
    float ymax = 0.0f;
    const float seen_y = a.amax_y ? amax_peek(a.amax_y) : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + FF_DIM + cbw + 8 * g);
        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
        float o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float v = y[4 * g + m] + bv[m];
            v = resid[4 * g + m] + v;
            o[m] = v;
            y[4 * g + m] = row_ok ? v : 0.0f;
            ymax = fmaxf(ymax, fabsf(v));
        }
        if (row_ok) *reinterpret_cast<float4 *>(Yout + (long)row * D_MODEL + cbw + 8 * g) = float4{o[0], o[1], o[2], o[3]};
    }
    if (!row_ok) ymax = 0.0f;
    ymax = wave_max_f32(ymax);
    if (lane == 0) {
        if (a.amax_y) atomic_max_abs_seen(a.amax_y, ymax, seen_y);
        if (a.amax_h) atomic_max_abs_seen(a.amax_h, hmax, seen_h);
        if (a.amax_o1) atomic_max_abs_seen(a.amax_o1, o1max, seen_o1);
    }
    if (!qkv) return;
    // ---- the next block's Q, K, V from Y's image (wave w: columns 32 w .. 32 w + 31 of each)
    if (lane == 0) atomicMax(&wg_max[3], __float_as_uint(ymax));
    __syncthreads();                           // (O1's image was last read before the hidden layer's second half was written)
    const float sy = pow2_scale_for(__uint_as_float(wg_max[3]));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned h0, l0, h1, l1;
        split2(y[4 * g] * sy, y[4 * g + 1] * sy, h0, l0);
        split2(y[4 * g + 2] * sy, y[4 * g + 3] * sy, h1, l1);
        *reinterpret_cast<u32x2 *>(X1h + q * FF_LD1 + cbw + 8 * g) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(X1l + q * FF_LD1 + cbw + 8 * g) = u32x2{l0, l1};
    }
    __syncthreads();
    const float unscale_y = 1.0f / sy;
    float *const out[3] = {a.Q, a.K, a.V};
    float seen_p[3], pmax[3];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) seen_p[pi] = amax_peek(a.amax_qkv + pi * AMAX_STRIDE);
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        // chunks: Wq in b1r, Wk in b0 (requested during the last product above), Wv into b1r once Wq's steps are done
        const f32x16 acc = pi == 1 ? steps8(b0, X1hq, X1lq, zero16()) : steps8(b1r, X1hq, X1lq, zero16());
        if (pi == 0) load16(b1r, qkv_chunk(a.Wvp));
        float m_ = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                o[m] = acc[4 * g + m] * unscale_y;
                m_ = fmaxf(m_, fabsf(o[m]));
            }
            if (row_ok) *reinterpret_cast<float4 *>(out[pi] + (long)row * D_MODEL + cbw + 8 * g) = float4{o[0], o[1], o[2], o[3]};
        }
        pmax[pi] = wave_max_f32(row_ok ? m_ : 0.0f);
    }
    if (lane == 0) {
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) atomic_max_abs_seen(a.amax_qkv + pi * AMAX_STRIDE, pmax[pi], seen_p[pi]);
    }
}
// The network's entry in one launch: the live tokens of the caller's tile, their embedding X0 = T We^T + be
// (TSP/TSPModel.py:262-266, CVRP/CVRPModel.py:402-408: F <= 12 features per token, binary32 FMAs) and the first
// block's Q, K, V from X0's image, as at the end of ff_fused_kernel.  32 tokens per workgroup; lane (q, h) of wave w
// owns token q and features 32 w + 4 h + 8 g + {0..3}.
struct EmbedArgs {
    const float *tile;                  // [R][row_stride]: n tokens of DGL_TILE_STRIDE floats per row
    int row_stride, n, F, M;
    const float *We, *be;               // [128][F], [128]
    float *T;                           // compact copy of the tokens [M][DGL_TILE_STRIDE] (the backward pass reads it), or NULL
    float *X0, *amax_x0;
    const _Float16 *Wqp, *Wkp, *Wvp;    // packed, K = 128, 128 columns each
    float *Q, *K, *V, *amax_qkv;
};
#define EMBED_LDS_BYTES (2 * 32 * FF_LD1 * 2 + (32 * DGL_TILE_STRIDE + D_MODEL * DGL_TILE_STRIDE + D_MODEL) * 4 + 16)
__global__ __launch_bounds__(256, 2) void embed_qkv_kernel(EmbedArgs a) {
    extern __shared__ __align__(16) _Float16 ff_lds[];
    _Float16 *X1h = ff_lds, *X1l = X1h + 32 * FF_LD1;
    float *Tl = reinterpret_cast<float *>(X1l + 32 * FF_LD1);                 // [32][DGL_TILE_STRIDE]
    float *Wl = Tl + 32 * DGL_TILE_STRIDE, *bl = Wl + D_MODEL * DGL_TILE_STRIDE;  // [128][F], [128]
    unsigned *wg_max = reinterpret_cast<unsigned *>(bl + D_MODEL);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32;
    u32x4 b0[16], b1r[16];
    auto load16 = [&](u32x4 (&b)[16], const u32x4 *p) {
#pragma unroll
        for (int e = 0; e < 16; ++e) b[e] = p[64 * e];
    };
    auto qkv_chunk = [&](const _Float16 *Wp) { return reinterpret_cast<const u32x4 *>(Wp) + (long)(w * (D_MODEL / 16)) * 2 * 64 + lane; };
    auto steps8 = [&](const u32x4 (&b)[16], const _Float16 *Th, const _Float16 *Tlo, f32x16 acc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 th = *reinterpret_cast<const f16x8 *>(Th + 16 * t), tl = *reinterpret_cast<const f16x8 *>(Tlo + 16 * t);
            const f16x8 bh = as_f16x8(b[2 * t]), blo = as_f16x8(b[2 * t + 1]);
            acc = MFMA16(bh, tl, acc);
            acc = MFMA16(blo, th, acc);
            acc = MFMA16(bh, th, acc);
        }
        return acc;
    };
    // ---- the tokens (rows beyond M: the last token, masked below), the embedding's weights
    for (int o = tid; o < 32 * DGL_TILE_STRIDE; o += 256) {
        const int row = o / DGL_TILE_STRIDE, c = o % DGL_TILE_STRIDE;
        const int m = min(i0 + row, a.M - 1), r = m / a.n, t = m % a.n;
        const float v = a.tile[(long)r * a.row_stride + t * DGL_TILE_STRIDE + c];
        Tl[o] = v;
        if (a.T != nullptr && i0 + row < a.M) a.T[(long)(i0 + row) * DGL_TILE_STRIDE + c] = v;
    }
    load16(b1r, qkv_chunk(a.Wqp));
    load16(b0, qkv_chunk(a.Wkp));
    for (int o = tid; o < D_MODEL * a.F; o += 256) Wl[o] = a.We[o];
    if (tid < D_MODEL) bl[tid] = a.be[tid];
    if (tid == 0) wg_max[0] = 0u;
    __syncthreads();
    const int row = i0 + q;
    const bool row_ok = row < a.M;
    const int cbw = 32 * w + 4 * h;
    float *const X0 = a.X0;
    float y[16], tq[DGL_TILE_STRIDE];
#pragma unroll
    for (int f = 0; f < DGL_TILE_STRIDE; ++f) tq[f] = Tl[q * DGL_TILE_STRIDE + f];
    float ymax = 0.0f;
    const float seen_x = a.amax_x0 ? amax_peek(a.amax_x0) : 0.0f;
    const int F = a.F;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c = cbw + 8 * g + m;
            float acc = 0.0f;
#pragma unroll
            for (int f = 0; f < DGL_TILE_STRIDE; ++f)
                if (f < F) acc = fmaf(tq[f], Wl[c * F + f], acc);
            const float v = row_ok ? acc + bl[c] : 0.0f;
            y[4 * g + m] = v;
            ymax = fmaxf(ymax, fabsf(v));
        }
        if (row_ok) *reinterpret_cast<float4 *>(X0 + (long)row * D_MODEL + cbw + 8 * g) = float4{y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
    }
    ymax = wave_max_f32(ymax);
    if (lane == 0) {
        if (a.amax_x0) atomic_max_abs_seen(a.amax_x0, ymax, seen_x);
        atomicMax(&wg_max[0], __float_as_uint(ymax));
    }
    __syncthreads();
    const float sy = pow2_scale_for(__uint_as_float(wg_max[0]));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned h0, l0, h1, l1;
        split2(y[4 * g] * sy, y[4 * g + 1] * sy, h0, l0);
        split2(y[4 * g + 2] * sy, y[4 * g + 3] * sy, h1, l1);
        *reinterpret_cast<u32x2 *>(X1h + q * FF_LD1 + cbw + 8 * g) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(X1l + q * FF_LD1 + cbw + 8 * g) = u32x2{l0, l1};
    }
    __syncthreads();
    const _Float16 *X1hq = X1h + q * FF_LD1 + 8 * h, *X1lq = X1l + q * FF_LD1 + 8 * h;
    const float unscale_y = 1.0f / sy;
    float *const out[3] = {a.Q, a.K, a.V};
    float seen_p[3], pmax[3];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) seen_p[pi] = amax_peek(a.amax_qkv + pi * AMAX_STRIDE);
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const f32x16 acc = pi == 1 ? steps8(b0, X1hq, X1lq, zero16()) : steps8(b1r, X1hq, X1lq, zero16());
        if (pi == 0) load16(b1r, qkv_chunk(a.Wvp));
        float m_ = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                o[m] = acc[4 * g + m] * unscale_y;
                m_ = fmaxf(m_, fabsf(o[m]));
            }
            if (row_ok) *reinterpret_cast<float4 *>(out[pi] + (long)row * D_MODEL + cbw + 8 * g) = float4{o[0], o[1], o[2], o[3]};
        }
        pmax[pi] = wave_max_f32(row_ok ? m_ : 0.0f);
    }
    if (lane == 0) {
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) atomic_max_abs_seen(a.amax_qkv + pi * AMAX_STRIDE, pmax[pi], seen_p[pi]);
    }
}
// The same chain backwards, one launch: dH = (dY W2) where H > 0;  dO1 = dY + dH W1;  dAO = dO1 Wc  (dY the gradient of
// the block's output).  All three go to memory: the weight gradients and the attention backward read them.
//   chunks: W2^T tiles w, w + 4 | W1^T tile w, k-steps 0-15 | W2^T tiles w + 8, w + 12 | W1^T tile w, k-steps 16-31 | Wc^T tile w
struct FfBwdArgs {
    const float *dY, *amax_dy;          // [M][128] and the bound of its magnitude
    const float *H;                     // the forward pass's hidden layer [M][512]: the ReLU's gate
    const _Float16 *W2p, *W1p, *Wcp;    // packed transposed weights: W2 (K = 128, 512 columns), W1 (K = 512, 128), Wc (K = 128, 128)
    float *dH, *amax_dh;                // [M][512]
    float *dO1, *amax_do1;              // [M][128]
    float *dAO, *amax_dao;              // [M][128]
    int M;
};
__global__ __launch_bounds__(256, 2) void ff_bwd_kernel(FfBwdArgs a) {
    extern __shared__ __align__(16) _Float16 ff_lds[];
    _Float16 *X1h = ff_lds, *X1l = X1h + 32 * FF_LD1, *H2h = X1l + 32 * FF_LD1, *H2l = H2h + 32 * FF_LD2;
    unsigned *wg_max = reinterpret_cast<unsigned *>(H2l + 32 * FF_LD2);       // largest magnitude of: dH half 0, 1, dO1
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32;
    u32x4 b0[16], b1r[16];
    auto load16 = [&](u32x4 (&b)[16], const u32x4 *p) {
#pragma unroll
        for (int e = 0; e < 16; ++e) b[e] = p[64 * e];
    };
    auto w2_chunk = [&](int tile) { return reinterpret_cast<const u32x4 *>(a.W2p) + (long)(tile * (D_MODEL / 16)) * 2 * 64 + lane; };
    auto w1_chunk = [&](int kc) { return reinterpret_cast<const u32x4 *>(a.W1p) + (long)(w * (FF_DIM / 16) + 8 * kc) * 2 * 64 + lane; };
    auto steps8 = [&](const u32x4 (&b)[16], const _Float16 *Th, const _Float16 *Tl, f32x16 acc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 th = *reinterpret_cast<const f16x8 *>(Th + 16 * t), tl = *reinterpret_cast<const f16x8 *>(Tl + 16 * t);
            const f16x8 bh = as_f16x8(b[2 * t]), bl = as_f16x8(b[2 * t + 1]);
            acc = MFMA16(bh, tl, acc);
            acc = MFMA16(bl, th, acc);
            acc = MFMA16(bh, th, acc);
        }
        return acc;
    };
    float sa;
    {
        float4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, row = idx >> 5, c4 = idx & 31;
            const int rr = min(i0 + row, a.M - 1);                        // (rows beyond M: a valid address, scaled by zero below)
            v[e] = *reinterpret_cast<const float4 *>(a.dY + (long)rr * D_MODEL + 4 * c4);
        }
        load16(b1r, w2_chunk(w));
        load16(b0, w2_chunk(w + 4));
        if (tid < 3) wg_max[tid] = 0u;
        sa = pow2_scale_for(amax_read(a.amax_dy));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, row = idx >> 5, c4 = idx & 31;
            const float sr = i0 + row < a.M ? sa : 0.0f;
            unsigned h0, l0, h1, l1;
            split2(v[e].x * sr, v[e].y * sr, h0, l0);
            split2(v[e].z * sr, v[e].w * sr, h1, l1);
            *reinterpret_cast<u32x2 *>(X1h + row * FF_LD1 + 4 * c4) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(X1l + row * FF_LD1 + 4 * c4) = u32x2{l0, l1};
        }
    }
    __syncthreads();
    const int row = i0 + q;                                               // this lane's token
    const bool row_ok = row < a.M;
    const long row_c = min(row, a.M - 1);
    float *const dHout = a.dH;
    float *const dO1out = a.dO1;
    float *const dAOout = a.dAO;
    const _Float16 *X1hq = X1h + q * FF_LD1 + 8 * h, *X1lq = X1l + q * FF_LD1 + 8 * h;
    const _Float16 *H2hq = H2h + q * FF_LD2 + 8 * h, *H2lq = H2l + q * FF_LD2 + 8 * h;
    const int cbw = 32 * w + 4 * h;                                       // this lane's features of a 128-wide tensor: cbw + 8 g + {0..3}
    const float unscale_a = 1.0f / sa;
    f32x16 y = zero16();
    float hmax = 0.0f, seen_h = 0.0f;
    float4 resid[4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        // hidden features 256 hf + 32 (w + 4 j) + ..., j = 0, 1
        f32x16 t[2];
        float4 gate[2][4];
        t[0] = steps8(b1r, X1hq, X1lq, zero16());
        load16(b1r, w1_chunk(2 * hf));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gate[j][g] = *reinterpret_cast<const float4 *>(a.H + row_c * FF_DIM + 256 * hf + 32 * (w + 4 * j) + 4 * h + 8 * g);
        t[1] = steps8(b0, X1hq, X1lq, zero16());
        load16(b0, w1_chunk(2 * hf + 1));
        if (hf == 1 && a.amax_dh) seen_h = amax_peek(a.amax_dh);
        float lmax = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cb = 256 * hf + 32 * (w + 4 * j) + 4 * h;           // this lane's features: cb + 8 g + {0..3}
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float gv[4] = {gate[j][g].x, gate[j][g].y, gate[j][g].z, gate[j][g].w};
                float o[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float v = t[j][4 * g + m] * unscale_a;
                    if (!(gv[m] > 0.0f)) v = 0.0f;
                    o[m] = row_ok ? v : 0.0f;
                    lmax = fmaxf(lmax, fabsf(o[m]));
                    t[j][4 * g + m] = o[m];
                }
                if (row_ok) *reinterpret_cast<float4 *>(dHout + (long)row * FF_DIM + cb + 8 * g) = float4{o[0], o[1], o[2], o[3]};
            }
        }
        lmax = wave_max_f32(lmax);
        hmax = fmaxf(hmax, lmax);
        if (lane == 0) atomicMax(&wg_max[hf], __float_as_uint(lmax));
        __syncthreads();                       // (every wave is past the previous half's last product: the image is free)
        const float sh = pow2_scale_for(__uint_as_float(wg_max[hf]));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h0, l0, h1, l1;
                split2(t[j][4 * g] * sh, t[j][4 * g + 1] * sh, h0, l0);
                split2(t[j][4 * g + 2] * sh, t[j][4 * g + 3] * sh, h1, l1);
                const int lc = 32 * (w + 4 * j) + 8 * g + 4 * h;
                *reinterpret_cast<u32x2 *>(H2h + q * FF_LD2 + lc) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(H2l + q * FF_LD2 + lc) = u32x2{l0, l1};
            }
        }
        __syncthreads();
        if (hf == 1) {
            // the residual, in flight during the last MFMAs
#pragma unroll
            for (int g = 0; g < 4; ++g) resid[g] = *reinterpret_cast<const float4 *>(a.dY + row_c * D_MODEL + cbw + 8 * g);
        }
        f32x16 part = steps8(b1r, H2hq, H2lq, zero16());
        if (hf == 0) load16(b1r, w2_chunk(w + 8));
        else load16(b1r, reinterpret_cast<const u32x4 *>(a.Wcp) + (long)(w * (D_MODEL / 16)) * 2 * 64 + lane);
        part = steps8(b0, H2hq + 128, H2lq + 128, part);
        if (hf == 0) load16(b0, w2_chunk(w + 12));
        const float unscale_h = 1.0f / sh;
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] += part[e] * unscale_h;
    }
    // ---- dO1 = dY + dH W1, stored and (scaled for these 32 tokens) turned into the image of the last product
    float o1max = 0.0f;
    const float seen_o1 = a.amax_do1 ? amax_peek(a.amax_do1) : 0.0f, seen_ao = a.amax_dao ? amax_peek(a.amax_dao) : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float rv[4] = {resid[g].x, resid[g].y, resid[g].z, resid[g].w};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float v = rv[m] + y[4 * g + m];
            v = row_ok ? v : 0.0f;
            y[4 * g + m] = v;
            o1max = fmaxf(o1max, fabsf(v));
        }
        if (row_ok) *reinterpret_cast<float4 *>(dO1out + (long)row * D_MODEL + cbw + 8 * g) = float4{y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
    }
    o1max = wave_max_f32(o1max);
    if (lane == 0) atomicMax(&wg_max[2], __float_as_uint(o1max));
    __syncthreads();                           // (the first product's image has long been read by every wave)
    const float s1 = pow2_scale_for(__uint_as_float(wg_max[2]));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned h0, l0, h1, l1;
        split2(y[4 * g] * s1, y[4 * g + 1] * s1, h0, l0);
        split2(y[4 * g + 2] * s1, y[4 * g + 3] * s1, h1, l1);
        *reinterpret_cast<u32x2 *>(X1h + q * FF_LD1 + cbw + 8 * g) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(X1l + q * FF_LD1 + cbw + 8 * g) = u32x2{l0, l1};
    }
    __syncthreads();
    const f32x16 acc = steps8(b1r, X1hq, X1lq, zero16());
    const float unscale_1 = 1.0f / s1;
    float aomax = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            o[m] = acc[4 * g + m] * unscale_1;
            aomax = fmaxf(aomax, fabsf(o[m]));
        }
        if (row_ok) *reinterpret_cast<float4 *>(dAOout + (long)row * D_MODEL + cbw + 8 * g) = float4{o[0], o[1], o[2], o[3]};
    }
    if (!row_ok) aomax = 0.0f;
    aomax = wave_max_f32(aomax);
    if (lane == 0) {
        if (a.amax_dh) atomic_max_abs_seen(a.amax_dh, hmax, seen_h);
        if (a.amax_do1) atomic_max_abs_seen(a.amax_do1, o1max, seen_o1);
        if (a.amax_dao) atomic_max_abs_seen(a.amax_dao, aomax, seen_ao);
    }
}
// Experiment switches, read from the environment once per call of dgl_policy_general (the test that compares the
// variants changes them between calls):
//   DGL_FF_FUSED=0     keeps the three products after a block's attention in separate launches
//   DGL_GRAD_PAIRED=1  sends a block's weight-gradient batch out beside the product that writes its input gradient
static bool g_ff_fused = true, g_grad_paired = false;
static void read_switches() {
    const char *e = getenv("DGL_FF_FUSED");
    g_ff_fused = !(e && atoi(e) == 0);
    e = getenv("DGL_GRAD_PAIRED");
    g_grad_paired = e && atoi(e) != 0;
}
static bool ff_fused_enabled() { return g_ff_fused; }

// dst[e] += part[0][e] + part[1][e] + ... (in that order), e < count, rows `stride` floats apart: the ordered sum of the
// partial products / partial column sums.  Several sums share one launch (block ranges as in gemm_kernel).
#define REDUCE_BATCH_MAX 16
struct ReduceItem {
    const float *part;
    float *dst;
    int Z;
    long count, stride;
};
struct ReduceBatch {
    ReduceItem it[REDUCE_BATCH_MAX];
    int first[REDUCE_BATCH_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceBatch batch) {
    // 32 elements per workgroup, 8 threads per element: thread (element, slice) adds the parts of its slice of Z in
    // order, four loads in flight, and the eight slice sums are added as a fixed tree
    __shared__ float red[8][32];
    int b = 0;
    while (b + 1 < batch.count && (int)blockIdx.x >= batch.first[b + 1]) ++b;
    const ReduceItem &it = batch.it[b];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long e = (long)(blockIdx.x - batch.first[b]) * 32 + el;
    const int per = (it.Z + 7) / 8, z_lo = sl * per, z_hi = min(it.Z, z_lo + per);
    float s = 0.0f;
    if (e < it.count) {
        const float *p = it.part + e;
        int z = z_lo;
        for (; z + 4 <= z_hi; z += 4) {
            const float a = p[(long)z * it.stride], b2 = p[(long)(z + 1) * it.stride], c = p[(long)(z + 2) * it.stride],
                        d = p[(long)(z + 3) * it.stride];
            s += a; s += b2; s += c; s += d;
        }
        for (; z < z_hi; ++z) s += p[(long)z * it.stride];
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < it.count)
        it.dst[e] += ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}
struct ReduceLaunch {
    ReduceBatch b;
    int blocks = 0;
    ReduceLaunch() { b.count = 0; }
    ReduceLaunch &add(const float *part, float *dst, int Z, long count, long stride) {
        b.it[b.count] = ReduceItem{part, dst, Z, count, stride};
        b.first[b.count++] = blocks;
        blocks += (int)((count + 31) / 32);
        return *this;
    }
    void run(hipStream_t s) {
        if (b.count == 0) return;
        b.first[b.count] = blocks;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, s, b);
    }
};

// compact copy of the live tokens: T[r*n + t][0..11] = tile[r][t][0..11]
__global__ void gather_tile_kernel(const float *__restrict__ tile, int row_stride, int R, int n, float *__restrict__ T) {
    const long total = (long)R * n * DGL_TILE_STRIDE;
    for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = o % DGL_TILE_STRIDE;
        const long tk = o / DGL_TILE_STRIDE;
        const int r = tk / n, t = tk % n;
        T[o] = tile[(long)r * row_stride + t * DGL_TILE_STRIDE + c];
    }
}

// ---------------------------------------------------------------------------- attention
// One workgroup per (row, head), n <= 256 tokens in NT = ceil(n / 32) tiles, head h = columns 16h..16h+15 of
// Q, K, V [M][128] (TSP/TSPModel.py:420-441): P[i][j] = softmax_j(q_i.k_j / 4), O_i = sum_j P[i][j] V_j.
// The n x n probabilities are never stored: the backward pass recomputes q.k and gets the row maximum and the row
// sum of exponentials from the forward pass (stats [R][8][n][2]); it uses  sum_j P[i][j] dP[i][j] = dO_i . O_i.
//
// All products run on the f16 matrix cores with split operands, as in gemm_kernel.  One wave owns a tile of 32
// queries (or, for dK and dV, 32 keys) and walks over the tiles of the other index.  A score tile is computed
// transposed - keys along the rows, the wave's queries along the columns - so that in the accumulator layout of
// v_mfma_f32_32x32x16 (lane = column, 16 rows per lane) each lane holds scores of ONE query: the softmax
// statistics are per lane, and the lane's 16 values are, as they stand, two B operands (k = key) of the next
// product, whose A operand (V^T, K^T, ...) is read from a transposed LDS image at the matching key positions:
//     accumulator register r of lane (q, h)  <->  row (r & 3) + 8 (r >> 2) + 4 h;
//     registers 8 s .. 8 s + 7  =  k-slots 0..7 of k-step s  <->  keys 16 s + 4 h + {0..3, 8..11} of the tile.
// Only 16 of the 32 output rows of those products (the head dimension) are used.
//
// LDS images of a [n][16] head slice, as binary16 hi and lo halves of (value * scale), rows >= n zero:
//   row-major, 16 halfs per row, the two 8-half blocks of a row swapped on every other group of four rows
//   (conflict-free 16-byte operand reads without padding);
//   transposed, [16][32 NT + 4] (row stride = 2 banks mod 32: conflict-free 8-byte reads).
__device__ __forceinline__ int rm_off(int row, int blk) { return row * 16 + ((blk ^ ((row >> 2) & 1)) << 3); }
#define ATT_TS(NT) (32 * (NT) + 4)

// Scores are kept in the base-2 domain: s2 = (q.k / 4) log2(e), the row statistics are max_j s2 and
// sum_j 2^(s2 - max), and a probability is one FMA (the raw accumulator times a constant, minus the maximum) and one
// v_exp_f32.  The rounding of that FMA's result, 2^-24 of |s2 - max|, is the error the subtraction s - max of a
// natural-base softmax carries too.
#define LOG2E 1.44269502f
__device__ __forceinline__ float exp2_neg(float t) { return __builtin_amdgcn_exp2f(t); }      // 0 for t = -inf

// Staging of head hd of src rows tok0 .. tok0 + n - 1 (times scale) into the images that are given (else nullptr), in
// two parts, so that a kernel can request everything it stages before it waits for any of it (a load placed after
// the LDS stores of the previous image would start a round trip of its own).  Thread `item` < n_pad takes the pair of
// tokens 2 (item / 2), + 1 and the 8-column block item % 2: the transposed image takes the pair as one packed word.
struct HeadRows { float4 a0, a1, b0, b1; };
__device__ __forceinline__ HeadRows stage_head_load(const float *__restrict__ src, long tok0, int n, int hd) {
    const int item = threadIdx.x, t0 = (item >> 1) * 2, blk = item & 1;
    // (rows beyond n: loaded from the last row and scaled by zero - a guard per lane would turn every 16-byte load
    // into four guarded 4-byte loads)
    const float *pa = src + (tok0 + min(t0, n - 1)) * D_MODEL + hd * HDIM + 8 * blk;
    const float *pb = src + (tok0 + min(t0 + 1, n - 1)) * D_MODEL + hd * HDIM + 8 * blk;
    return HeadRows{*reinterpret_cast<const float4 *>(pa), *reinterpret_cast<const float4 *>(pa + 4),
                    *reinterpret_cast<const float4 *>(pb), *reinterpret_cast<const float4 *>(pb + 4)};
}
__device__ __forceinline__ void stage_head_store(const HeadRows &r, int n, int n_pad, float scale,
                                                 _Float16 *rm_hi, _Float16 *rm_lo, _Float16 *t_hi, _Float16 *t_lo, int TS) {
    const int item = threadIdx.x, t0 = (item >> 1) * 2, blk = item & 1;
    if (item >= n_pad) return;
    float a[8], b[8];
    const float sa = t0 < n ? scale : 0.0f, sb = t0 + 1 < n ? scale : 0.0f;
    a[0] = r.a0.x * sa; a[1] = r.a0.y * sa; a[2] = r.a0.z * sa; a[3] = r.a0.w * sa;
    a[4] = r.a1.x * sa; a[5] = r.a1.y * sa; a[6] = r.a1.z * sa; a[7] = r.a1.w * sa;
    b[0] = r.b0.x * sb; b[1] = r.b0.y * sb; b[2] = r.b0.z * sb; b[3] = r.b0.w * sb;
    b[4] = r.b1.x * sb; b[5] = r.b1.y * sb; b[6] = r.b1.z * sb; b[7] = r.b1.w * sb;
    if (rm_hi) {
        u32x4 ah, al, bh, bl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hi, lo;
            split2(a[2 * e], a[2 * e + 1], hi, lo); ah[e] = hi; al[e] = lo;
            split2(b[2 * e], b[2 * e + 1], hi, lo); bh[e] = hi; bl[e] = lo;
        }
        *reinterpret_cast<u32x4 *>(rm_hi + rm_off(t0, blk)) = ah;
        *reinterpret_cast<u32x4 *>(rm_lo + rm_off(t0, blk)) = al;
        *reinterpret_cast<u32x4 *>(rm_hi + rm_off(t0 + 1, blk)) = bh;
        *reinterpret_cast<u32x4 *>(rm_lo + rm_off(t0 + 1, blk)) = bl;
    }
    if (t_hi) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned hi, lo;
            split2(a[e], b[e], hi, lo);
            *reinterpret_cast<unsigned *>(t_hi + (8 * blk + e) * TS + t0) = hi;
            *reinterpret_cast<unsigned *>(t_lo + (8 * blk + e) * TS + t0) = lo;
        }
    }
}
// the wave's own 32 rows as a B operand (lane (q, h): columns 8h..8h+7 of row tok0 + 32 tile + q), rows >= n zero
__device__ __forceinline__ void load_own_rows(const float *__restrict__ src, long tok0, int n, int row, int hd, int h, float (&v)[8]) {
    const float *p = src + (tok0 + min(row, n - 1)) * D_MODEL + hd * HDIM + 8 * h;     // (unguarded loads, masked afterwards)
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    const bool in = row < n;
    v[0] = in ? a.x : 0.0f; v[1] = in ? a.y : 0.0f; v[2] = in ? a.z : 0.0f; v[3] = in ? a.w : 0.0f;
    v[4] = in ? b.x : 0.0f; v[5] = in ? b.y : 0.0f; v[6] = in ? b.z : 0.0f; v[7] = in ? b.w : 0.0f;
}
__device__ __forceinline__ void split8(const float (&v)[8], float scale, f16x8 &hi, f16x8 &lo) {
    u32x4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b;
        split2(v[2 * e] * scale, v[2 * e + 1] * scale, a, b);
        h4[e] = a; l4[e] = b;
    }
    hi = as_f16x8(h4); lo = as_f16x8(l4);
}
// registers 8 s .. 8 s + 7 of an accumulator tile (times scale) as the B operand of k-step s
__device__ __forceinline__ void acc_operand(const f32x16 &v, int s, float scale, f16x8 &hi, f16x8 &lo) {
    u32x4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b;
        split2(v[8 * s + 2 * e] * scale, v[8 * s + 2 * e + 1] * scale, a, b);
        h4[e] = a; l4[e] = b;
    }
    hi = as_f16x8(h4); lo = as_f16x8(l4);
}
// A operand of k-step s from a transposed image: row d = lane & 15, positions tile * 32 + 16 s + 4 h + {0..3, 8..11}
__device__ __forceinline__ f16x8 t_operand(const _Float16 *img, int TS, int d, int tile, int s, int h) {
    const _Float16 *p = img + d * TS + tile * 32 + 16 * s + 4 * h;
    const u32x2 a = *reinterpret_cast<const u32x2 *>(p), b = *reinterpret_cast<const u32x2 *>(p + 8);
    return as_f16x8(u32x4{a[0], a[1], b[0], b[1]});
}
// acc += A B with split operands
__device__ __forceinline__ f32x16 mfma_split(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x16 acc) {
    acc = MFMA16(al, bh, acc);
    acc = MFMA16(ah, bl, acc);
    return MFMA16(ah, bh, acc);
}
// rows d = 4h..4h+3 and 8+4h..8+4h+3 of an accumulator (registers 0..7) times scale -> dst[16 hd + d] of the lane's token
__device__ __forceinline__ float store_head_rows(float *__restrict__ dst, const f32x16 &acc, float scale, int h) {
    const float4 a = {acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale};
    const float4 b = {acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale};
    *reinterpret_cast<float4 *>(dst + 4 * h) = a;
    *reinterpret_cast<float4 *>(dst + 8 + 4 * h) = b;
    return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                 fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}

// amax_qkv: the bounds of |Q|, |K|, |V| (AMAX_STRIDE apart).  blockDim = 64 NT.
__global__ __launch_bounds__(512) void attn_fwd_kernel(const float *__restrict__ Q, const float *__restrict__ K,
                                                       const float *__restrict__ V, int n, const float *__restrict__ amax_qkv,
                                                       float *__restrict__ stats, float *__restrict__ O, float *__restrict__ amax_o) {
    extern __shared__ __align__(16) _Float16 att_lds[];
    const int NT = blockDim.x >> 6, n_pad = 32 * NT, TS = ATT_TS(NT);
    _Float16 *Kh = att_lds, *Kl = Kh + n_pad * 16, *VTh = Kl + n_pad * 16, *VTl = VTh + 16 * TS;
    const int r = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x & 63, it = threadIdx.x >> 6, q = lane & 31, h = lane >> 5;
    const long tok0 = (long)r * n;
    const int i = 32 * it + q;
    const HeadRows rk = stage_head_load(K, tok0, n, hd), rv = stage_head_load(V, tok0, n, hd);
    float qv[8];
    load_own_rows(Q, tok0, n, i, hd, h, qv);
    const float sq = pow2_scale_for(amax_read(amax_qkv)), sk = pow2_scale_for(amax_read(amax_qkv + AMAX_STRIDE)),
                sv = pow2_scale_for(amax_read(amax_qkv + 2 * AMAX_STRIDE));
    stage_head_store(rk, n, n_pad, sk, Kh, Kl, nullptr, nullptr, TS);
    stage_head_store(rv, n, n_pad, sv, nullptr, nullptr, VTh, VTl, TS);
    f16x8 qh, ql;
    split8(qv, sq, qh, ql);
    __syncthreads();
    const float cs = 0.25f * LOG2E / (sq * sk);
    // one pass over the key tiles with a running maximum: whenever it grows, the sum and the output accumulated so
    // far are rescaled (2^(old - new) <= 1).  Equal to softmax(s) V up to rounding.
    float mx = -INFINITY, sum = 0.0f;
    f32x16 o = zero16();
    for (int jt = 0; jt < NT; ++jt) {
        const int ko = rm_off(32 * jt + q, h);
        f32x16 sc = mfma_split(*reinterpret_cast<const f16x8 *>(Kh + ko), *reinterpret_cast<const f16x8 *>(Kl + ko), qh, ql, zero16());
        if (32 * jt + 32 > n) {                          // the last tile: keys beyond n drop out
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (32 * jt + (e & 3) + 8 * (e >> 2) + 4 * h >= n) sc[e] = -INFINITY;
        }
        float tmax = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, sc[e]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * cs;           // cs > 0: the maximum commutes with the scaling
        if (tmax > mx) {
            const float f = exp2_neg(mx - tmax);         // 0 on the first tile (mx = -inf)
            sum *= f;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] *= f;
            mx = tmax;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sc[e] = exp2_neg(fmaf(sc[e], cs, -mx));
            sum += sc[e];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ph, pl;
            acc_operand(sc, s, 1024.0f, ph, pl);
            o = mfma_split(t_operand(VTh, TS, q & 15, jt, s, h), t_operand(VTl, TS, q & 15, jt, s, h), ph, pl, o);
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    float omax = 0.0f;
    const float seen_o = amax_peek(amax_o);
    if (i < n) {
        omax = store_head_rows(O + (tok0 + i) * D_MODEL + hd * HDIM, o, 1.0f / (1024.0f * sv * sum), h);
        if (h == 0) {
            float *st = stats + (((long)r * HEADS + hd) * n + i) * 2;
            st[0] = mx;
            st[1] = sum;
        }
    }
    omax = wave_max_f32(omax);
    if (lane == 0) atomic_max_abs_seen(amax_o, omax, seen_o);
}

// Backward of the above.  dO, O [M][128] -> dQ, dK, dV [M][128].
//   dS[i][j] = P[i][j] (dO_i . V_j - dO_i . O_i) / 4 (gradient with respect to q.k, the 1/4 included)
//   dQ_i = sum_j dS[i][j] K_j;  dK_j = sum_i dS[i][j] Q_i;  dV_j = sum_i P[i][j] dO_i
// dS is scaled for the binary16 operands by a bound of its magnitude, |dS[i][j]| <= (|dO_i|_1 max|V| + |dO_i . O_i|) / 4.
//
// attn_bwd_q: a wave owns 32 queries and walks over the key tiles (score tiles transposed, as in the forward).
__device__ __forceinline__ void attn_bwd_q(const float *__restrict__ Q, const float *__restrict__ K,
                                                         const float *__restrict__ V, const float *__restrict__ O,
                                                         const float *__restrict__ dO, int n, const float *__restrict__ amax_qkv,
                                                         const float *__restrict__ amax_do, const float *__restrict__ stats,
                                                         float *__restrict__ dQ, float *__restrict__ amax_dq) {
    extern __shared__ __align__(16) _Float16 att_lds[];
    const int NT = blockDim.x >> 6, n_pad = 32 * NT, TS = ATT_TS(NT);
    _Float16 *Kh = att_lds, *Kl = Kh + n_pad * 16, *Vh = Kl + n_pad * 16, *Vl = Vh + n_pad * 16, *KTh = Vl + n_pad * 16, *KTl = KTh + 16 * TS;
    const int r = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x & 63, it = threadIdx.x >> 6, q = lane & 31, h = lane >> 5;
    const long tok0 = (long)r * n;
    const int i = 32 * it + q;
    const HeadRows rk = stage_head_load(K, tok0, n, hd), rv = stage_head_load(V, tok0, n, hd);
    float qv[8], gv[8], ov[8];
    load_own_rows(Q, tok0, n, i, hd, h, qv);
    load_own_rows(dO, tok0, n, i, hd, h, gv);
    load_own_rows(O, tok0, n, i, hd, h, ov);
    const float *st = stats + (((long)r * HEADS + hd) * n + min(i, n - 1)) * 2;
    const float st_max = st[0], st_sum = st[1];
    const float bound_v = amax_read(amax_qkv + 2 * AMAX_STRIDE);
    const float sq = pow2_scale_for(amax_read(amax_qkv)), sk = pow2_scale_for(amax_read(amax_qkv + AMAX_STRIDE)),
                sv = pow2_scale_for(bound_v), sg = pow2_scale_for(amax_read(amax_do));
    stage_head_store(rk, n, n_pad, sk, Kh, Kl, KTh, KTl, TS);
    stage_head_store(rv, n, n_pad, sv, Vh, Vl, nullptr, nullptr, TS);
    float dt = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dt = fmaf(gv[e], ov[e], dt); l1 += fabsf(gv[e]); }
    dt += __shfl_xor(dt, 32, 64);
    l1 += __shfl_xor(l1, 32, 64);
    const float c = pow2_scale_for((l1 * bound_v + fabsf(dt)) * 0.25f);       // per query: a column of the dS^T operand
    f16x8 qh, ql, gh, gl;
    split8(qv, sq, qh, ql);
    split8(gv, sg, gh, gl);
    const float mx = i < n ? st_max : 0.0f, inv = i < n ? 1.0f / st_sum : 1.0f;
    __syncthreads();
    const float cs = 0.25f * LOG2E / (sq * sk), cg = 1.0f / (sv * sg);
    f32x16 dq = zero16();
    for (int jt = 0; jt < NT; ++jt) {
        const int ko = rm_off(32 * jt + q, h);
        f32x16 sc = mfma_split(*reinterpret_cast<const f16x8 *>(Kh + ko), *reinterpret_cast<const f16x8 *>(Kl + ko), qh, ql, zero16());
        const f32x16 dp = mfma_split(*reinterpret_cast<const f16x8 *>(Vh + ko), *reinterpret_cast<const f16x8 *>(Vl + ko), gh, gl, zero16());
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = exp2_neg(fmaf(sc[e], cs, -mx)) * inv;
            sc[e] = p * (dp[e] * cg - dt) * 0.25f;                              // dS[i][j]
        }
        if (32 * jt + 32 > n) {                          // the last tile: keys beyond n drop out
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (32 * jt + (e & 3) + 8 * (e >> 2) + 4 * h >= n) sc[e] = 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 dh, dl;
            acc_operand(sc, s, c, dh, dl);
            dq = mfma_split(t_operand(KTh, TS, q & 15, jt, s, h), t_operand(KTl, TS, q & 15, jt, s, h), dh, dl, dq);
        }
    }
    float m = 0.0f;
    const float seen_dq = amax_peek(amax_dq);
    if (i < n) m = store_head_rows(dQ + (tok0 + i) * D_MODEL + hd * HDIM, dq, 1.0f / (sk * c), h);
    m = wave_max_f32(m);
    if (lane == 0) atomic_max_abs_seen(amax_dq, m, seen_dq);
}

// attn_bwd_kv: a wave owns 32 keys and walks over the query tiles (score tiles with queries along the rows:
// the per-query statistics come from LDS, four consecutive queries per accumulator register group).
__device__ __forceinline__ void attn_bwd_kv(const float *__restrict__ Q, const float *__restrict__ K,
                                                          const float *__restrict__ V, const float *__restrict__ O,
                                                          const float *__restrict__ dO, int n, const float *__restrict__ amax_qkv,
                                                          const float *__restrict__ amax_do, const float *__restrict__ stats,
                                                          float *__restrict__ dK, float *__restrict__ dV,
                                                          float *__restrict__ amax_dkv) {        // bounds of |dK|, |dV|
    extern __shared__ __align__(16) _Float16 att_lds[];
    const int NT = blockDim.x >> 6, n_pad = 32 * NT, TS = ATT_TS(NT);
    _Float16 *Qh = att_lds, *Ql = Qh + n_pad * 16, *Gh = Ql + n_pad * 16, *Gl = Gh + n_pad * 16;
    _Float16 *QTh = Gl + n_pad * 16, *QTl = QTh + 16 * TS, *GTh = QTl + 16 * TS, *GTl = GTh + 16 * TS;
    float *Mx = reinterpret_cast<float *>(GTl + 16 * TS), *Inv = Mx + n_pad, *Dt = Inv + n_pad;
    unsigned *bound_bits = reinterpret_cast<unsigned *>(Dt + n_pad);
    const int r = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x & 63, jt = threadIdx.x >> 6, q = lane & 31, h = lane >> 5;
    const long tok0 = (long)r * n;
    const float bound_v = amax_read(amax_qkv + 2 * AMAX_STRIDE);
    const float sq = pow2_scale_for(amax_read(amax_qkv)), sk = pow2_scale_for(amax_read(amax_qkv + AMAX_STRIDE)),
                sv = pow2_scale_for(bound_v), sg = pow2_scale_for(amax_read(amax_do));
    if (threadIdx.x == 0) *bound_bits = 0u;
    // everything this workgroup reads from memory is requested here, before the first wait
    const HeadRows rq = stage_head_load(Q, tok0, n, hd), rg = stage_head_load(dO, tok0, n, hd);
    const int j = 32 * jt + q;
    float kv[8], vv[8];
    load_own_rows(K, tok0, n, j, hd, h, kv);
    load_own_rows(V, tok0, n, j, hd, h, vv);
    float bound_t;
    {   // per query (thread t): row maximum, 1 / row sum, dO.O, and the bound of its dS entries
        const int t = threadIdx.x, tc = min(t, n - 1);
        const float *st = stats + (((long)r * HEADS + hd) * n + tc) * 2;
        const float st_max = st[0], st_sum = st[1];
        const float *pg = dO + (tok0 + tc) * D_MODEL + hd * HDIM, *po = O + (tok0 + tc) * D_MODEL + hd * HDIM;
        float4 g4[HDIM / 4], o4[HDIM / 4];
#pragma unroll
        for (int e = 0; e < HDIM / 4; ++e) { g4[e] = *reinterpret_cast<const float4 *>(pg + 4 * e); o4[e] = *reinterpret_cast<const float4 *>(po + 4 * e); }
        stage_head_store(rq, n, n_pad, sq, Qh, Ql, QTh, QTl, TS);
        stage_head_store(rg, n, n_pad, sg, Gh, Gl, GTh, GTl, TS);
        float d = 0.0f, l1 = 0.0f;
#pragma unroll
        for (int e = 0; e < HDIM / 4; ++e) {
            d = fmaf(g4[e].x, o4[e].x, d); d = fmaf(g4[e].y, o4[e].y, d); d = fmaf(g4[e].z, o4[e].z, d); d = fmaf(g4[e].w, o4[e].w, d);
            l1 += fabsf(g4[e].x) + fabsf(g4[e].y) + fabsf(g4[e].z) + fabsf(g4[e].w);
        }
        const bool in = t < n;
        if (!in) { d = 0.0f; l1 = 0.0f; }
        if (t < n_pad) { Mx[t] = in ? st_max : 0.0f; Inv[t] = in ? 1.0f / st_sum : 0.0f; Dt[t] = d; }   // Inv = 0 beyond n: those rows' P vanish
        bound_t = wave_max_f32((l1 * bound_v + fabsf(d)) * 0.25f);
    }
    f16x8 kh, kl, vh, vl;
    split8(kv, sk, kh, kl);
    split8(vv, sv, vh, vl);
    __syncthreads();                                         // (the bound's zero is visible)
    if (lane == 0) atomicMax(bound_bits, __float_as_uint(bound_t));
    __syncthreads();
    const float c = pow2_scale_for(__uint_as_float(*bound_bits));
    const float cs = 0.25f * LOG2E / (sq * sk), cg = 1.0f / (sv * sg);
    f32x16 dk = zero16(), dv = zero16();
    for (int it = 0; it < NT; ++it) {
        const int qo = rm_off(32 * it + q, h);
        f32x16 sc = mfma_split(*reinterpret_cast<const f16x8 *>(Qh + qo), *reinterpret_cast<const f16x8 *>(Ql + qo), kh, kl, zero16());
        f32x16 dp = mfma_split(*reinterpret_cast<const f16x8 *>(Gh + qo), *reinterpret_cast<const f16x8 *>(Gl + qo), vh, vl, zero16());
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int i0 = 32 * it + 8 * g4 + 4 * h;
            const float4 m4 = *reinterpret_cast<const float4 *>(Mx + i0), v4 = *reinterpret_cast<const float4 *>(Inv + i0),
                         d4 = *reinterpret_cast<const float4 *>(Dt + i0);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {v4.x, v4.y, v4.z, v4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = exp2_neg(fmaf(sc[4 * g4 + e], cs, -mm[e])) * ii[e];
                sc[4 * g4 + e] = p;                                               // P[i][j]
                dp[4 * g4 + e] = p * (dp[4 * g4 + e] * cg - dd[e]) * 0.25f;       // dS[i][j]
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ph, pl, dh, dl;
            acc_operand(sc, s, 1024.0f, ph, pl);
            acc_operand(dp, s, c, dh, dl);
            dv = mfma_split(t_operand(GTh, TS, q & 15, it, s, h), t_operand(GTl, TS, q & 15, it, s, h), ph, pl, dv);
            dk = mfma_split(t_operand(QTh, TS, q & 15, it, s, h), t_operand(QTl, TS, q & 15, it, s, h), dh, dl, dk);
        }
    }
    float mk = 0.0f, mv = 0.0f;
    const float seen_dk = amax_peek(amax_dkv), seen_dv = amax_peek(amax_dkv + AMAX_STRIDE);
    if (j < n) {
        mk = store_head_rows(dK + (tok0 + j) * D_MODEL + hd * HDIM, dk, 1.0f / (sq * c), h);
        mv = store_head_rows(dV + (tok0 + j) * D_MODEL + hd * HDIM, dv, 1.0f / (1024.0f * sg), h);
    }
    mk = wave_max_f32(mk);
    mv = wave_max_f32(mv);
    if (lane == 0) {
        atomic_max_abs_seen(amax_dkv, mk, seen_dk);
        atomic_max_abs_seen(amax_dkv + AMAX_STRIDE, mv, seen_dv);
    }
}
// both halves of the attention backward in one launch: blockIdx.z = 0 computes dQ, 1 computes dK and dV
__global__ __launch_bounds__(512, 4) void attn_bwd_kernel(const float *__restrict__ Q, const float *__restrict__ K,
                                                       const float *__restrict__ V, const float *__restrict__ O,
                                                       const float *__restrict__ dO, int n, const float *__restrict__ amax_qkv,
                                                       const float *__restrict__ amax_do, const float *__restrict__ stats,
                                                       float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV,
                                                       float *__restrict__ amax_dqkv) {       // bounds of |dQ|, |dK|, |dV|
    if (blockIdx.z == 0) attn_bwd_q(Q, K, V, O, dO, n, amax_qkv, amax_do, stats, dQ, amax_dqkv);
    else attn_bwd_kv(Q, K, V, O, dO, n, amax_qkv, amax_do, stats, dK, dV, amax_dqkv + AMAX_STRIDE);
}
static size_t attn_lds_bytes(int NT, int rm_images, int t_images, int stat_floats) {
    return (size_t)2 * 2 * (rm_images * 32 * NT * 16 + t_images * 16 * ATT_TS(NT)) + (size_t)stat_floats * 4;
}

// ------------------------------------------------------------------------------ anchors
// Decoder entry (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-449): the anchor token(s) of every row are
// replaced by  W_a x_anchor + b_a  in place; the replaced rows are kept in Ea[anchor][row][128].
// One workgroup per (row, anchor), 128 threads.
__global__ __launch_bounds__(128) void anchor_fwd_kernel(float *__restrict__ X, float *__restrict__ Ea, int n, int tok0_, int tok1_,
                                                         const float *__restrict__ W0, const float *__restrict__ b0,
                                                         const float *__restrict__ W1, const float *__restrict__ b1,
                                                         float *__restrict__ amax_y) {
    __shared__ float xs[D_MODEL];
    const int r = blockIdx.x, a = blockIdx.y, c = threadIdx.x;
    const long tk = (long)r * n + (a == 0 ? tok0_ : tok1_);
    const float *W = a == 0 ? W0 : W1, *b = a == 0 ? b0 : b1;
    xs[c] = X[tk * D_MODEL + c];
    Ea[((long)a * gridDim.x + r) * D_MODEL + c] = xs[c];          // the encoder's output at the anchor, for the backward pass
    __syncthreads();
    float s = 0.0f;
    for (int k = 0; k < D_MODEL; ++k) s = fmaf(W[c * D_MODEL + k], xs[k], s);
    s += b[c];
    X[tk * D_MODEL + c] = s;                                       // in place: the row was read into LDS first
    s = wave_max_f32(fabsf(s));
    if ((c & 63) == 0) atomic_max_abs(amax_y, s);
}
// dX = dY except at the anchors, where dX_anchor = W_a^T dY_anchor; row r's term of db_a (= dY_anchor) goes to
// db_part[a][r][128] (dW_a += dY_anchor x_anchor^T runs before this kernel as a strided GEMM over the rows' anchor tokens)
__global__ __launch_bounds__(128) void anchor_bwd_kernel(float *__restrict__ dY, int n, int tok0_, int tok1_,
                                                         const float *__restrict__ W0, const float *__restrict__ W1,
                                                         float *__restrict__ db_part, float *__restrict__ amax_dx) {
    __shared__ float gs[D_MODEL];
    const int r = blockIdx.x, a = blockIdx.y, c = threadIdx.x;
    const long tk = (long)r * n + (a == 0 ? tok0_ : tok1_);
    const float *W = a == 0 ? W0 : W1;
    gs[c] = dY[tk * D_MODEL + c];
    __syncthreads();
    float s = 0.0f;
    for (int o = 0; o < D_MODEL; ++o) s = fmaf(W[o * D_MODEL + c], gs[o], s);
    dY[tk * D_MODEL + c] = s;                                     // becomes dX in place
    db_part[((long)a * gridDim.x + r) * D_MODEL + c] = gs[c];
    s = wave_max_f32(fabsf(s));
    if ((c & 63) == 0) atomic_max_abs(amax_dx, s);
}

// -------------------------------------------------------------------------- output stage
// Per row: logits = X wf + bf, masking, softmax, floor (TSP/TSPModel.py:312-326,
// CVRP/CVRPModel.py:462-493), the outputs of dgl_policy_forward, and for training the teacher's
// probability (1e-20 when the teacher's action is not a candidate, as in the dense tensor the
// reference indexes) and dX = d(-mean log p_teacher)/dX, dwf, dbf.  One workgroup of four waves per row.
__global__ __launch_bounds__(256) void output_kernel(const float *__restrict__ X, const float *__restrict__ wf,
                                                     const float *__restrict__ bf, dgl_policy_io_t io,
                                                     const int32_t *__restrict__ teacher, float grad_scale,
                                                     float *__restrict__ teacher_prob, float *__restrict__ dX,
                                                     float *__restrict__ dwf_part,      // [R][128]: row r's term of dwf
                                                     float *__restrict__ dbf_part,      // [R]
                                                     float *__restrict__ amax_dx) {
    extern __shared__ __align__(16) float xs[]; // [n][129]: the row's final hidden states (odd stride: a thread per token
                                                // reads down its row without bank conflicts); then lg and the partial dwf
    constexpr int XS = D_MODEL + 1;
    float *lg = xs + (size_t)io.n * XS;         // [n] logits, then probabilities, then logit gradients; [256] partial dwf
    __shared__ int s_teacher;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = io.n;
    const int stride = io.out_stride > 0 ? io.out_stride : DGL_MAX_TOKENS;
    const int anchor0 = io.k_last, anchor1 = n - 1;
    // (the descriptor's fields in registers: after a store the compiler re-reads from the argument segment whatever
    // it needs of a by-value struct, one scalar round trip per use)
    float *const logits_out = io.logits_out, *const prob_out = io.prob_out;
    int32_t *const selected_out = io.selected_out;
    const uint8_t *const infeasible = io.infeasible;
    const int32_t *const idx_last = io.idx_last, *const idx_depot = io.idx_depot;
    const int idx_stride = io.idx_stride, idx_dep_stride = io.idx_dep_stride, k_last = io.k_last, n_nodes = io.N;
    const bool cvrp = io.problem == DGL_PROBLEM_CVRP;
    const float *xr = X + (long)r * n * D_MODEL;
    float *dw_half = lg + n;
    auto is_cand = [&](int t) { return t != anchor0 && t != anchor1; };
    auto is_open = [&](int t) {
        if (!is_cand(t)) return false;
        if (cvrp && infeasible != nullptr && t < k_last) return infeasible[(size_t)r * idx_stride + t] == 0;
        return true;
    };
    auto action = [&](int t) {
        if (!cvrp || t < k_last) return idx_last[(size_t)r * idx_stride + t];
        return idx_depot[(size_t)r * idx_dep_stride + (t - k_last - 1)] + n_nodes + 1;
    };
    // the row's hidden states into LDS, sixteen 16-byte loads in flight per thread (64 workgroups cannot hide
    // memory latency behind each other: every dependent round trip to memory counts in full)
    {
        const float4 *src = reinterpret_cast<const float4 *>(xr);
        const int total = n * (D_MODEL / 4);
        for (int e0 = tid; e0 < total; e0 += 16 * 256) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[min(e0 + 256 * u, total - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + 256 * u;
                if (e < total) {
                    float *d = xs + (e >> 5) * XS + (e & 31) * 4;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
        if (tid < D_MODEL) lg[n + tid] = wf[tid];                   // (the partial-dwf area holds wf until the logits are done)
    }
    __syncthreads();
    // logits: a thread per token, columns in order (one FMA chain, as the fused decode kernels' final layer)
    if (tid < n) {
        const float *x = xs + tid * XS, *wv = lg + n;
        float sum = 0.0f;
#pragma unroll 16
        for (int c = 0; c < D_MODEL; ++c) sum = fmaf(x[c], wv[c], sum);
        sum += bf[0];
        lg[tid] = sum;
        if (logits_out) logits_out[(size_t)r * stride + tid] = sum;
    }
    __syncthreads();
    const bool train = teacher != nullptr;
    float pt = 1e-20f;
    if (wave == 0) {
        // masking, softmax, floor, greedy pick and the teacher's probability: one wave, four tokens per lane
        // (n <= 256), each token's mask bit and action id fetched once, all of them in flight together
        bool cand[4], open[4];
        int act[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = lane + 64 * u;
            cand[u] = t < n && is_cand(t);
            open[u] = t < n && is_open(t);
            act[u] = cand[u] ? action(t) : 0x7FFFFFFF;
        }
        float v[4], mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = lane + 64 * u < n ? lg[lane + 64 * u] : 0.0f;
            if (open[u]) mx = fmaxf(mx, v[u]);
        }
        mx = wave_max_f32(mx);
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = open[u] ? expf(v[u] - mx) : 0.0f;
            sum += v[u];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        float best = -1.0f;
        int best_act = 0x7FFFFFFF, t_teacher = -1;
        const int want = train ? teacher[r] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = lane + 64 * u;
            v[u] = v[u] / sum;                           // pure softmax (kept for the gradient)
            float pf = v[u];
            if (pf <= 1e-5f) pf = pf + 1e-7f;
            if (!cand[u]) pf = 0.0f;
            if (t < n) {
                lg[t] = v[u];
                if (prob_out) prob_out[(size_t)r * stride + t] = pf;
            }
            if (cand[u]) {
                if (pf > best || (pf == best && act[u] < best_act)) { best = pf; best_act = act[u]; }
                if (act[u] == want) t_teacher = t;
            }
        }
        // greedy pick: highest probability, lowest action id on ties
        const float wbest = wave_max_f32(best);
        int cand_act = (best == wbest) ? best_act : 0x7FFFFFFF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand_act = min(cand_act, __shfl_xor(cand_act, o, 64));
        if (lane == 0 && selected_out) selected_out[r] = cand_act;
        if (train) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t_teacher = max(t_teacher, __shfl_xor(t_teacher, o, 64));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            float g_logit_scale = 0.0f, p_soft = 0.0f;
            if (t_teacher >= 0) {
                p_soft = lg[t_teacher];
                pt = p_soft <= 1e-5f ? p_soft + 1e-7f : p_soft;
                // L = -mean log p_teacher: dL/dp_t = -grad_scale / p_t (grad_scale = 1 / rows of the global batch)
                g_logit_scale = -grad_scale / pt;
            }
            if (lane == 0 && teacher_prob) teacher_prob[r] = pt;
            if (dX) {
                float dbsum = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = lane + 64 * u;
                    // d p_t / d logit_j = p_t (delta_tj - p_j) over the open tokens
                    float dl = 0.0f;
                    if (t_teacher >= 0 && open[u]) dl = g_logit_scale * p_soft * ((t == t_teacher ? 1.0f : 0.0f) - v[u]);
                    if (t < n) lg[t] = dl;
                    dbsum += dl;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dbsum += __shfl_xor(dbsum, o, 64);
                if (lane == 0) dbf_part[r] = dbsum;
            }
        }
    }
    if (!train || !dX) return;
    __syncthreads();
    // dX = dlogit wf^T and row r's term of dwf = sum_t dlogit_t x_t: thread = (column, half of the tokens)
    const int c = tid & (D_MODEL - 1), half = tid >> 7;
    const float w = wf[c];
    float dw = 0.0f, gmax = 0.0f;
    // (loads and stores in separate loops: interleaved, every load would wait behind the store before it)
#pragma unroll 8
    for (int t = half; t < n; t += 2) dw = fmaf(lg[t], xs[t * XS + c], dw);
#pragma unroll 8
    for (int t = half; t < n; t += 2) {
        const float g = lg[t] * w;
        dX[((long)r * n + t) * D_MODEL + c] = g;
        gmax = fmaxf(gmax, fabsf(g));
    }
    dw_half[tid] = dw;
    __syncthreads();
    if (half == 0) dwf_part[(long)r * D_MODEL + c] = dw_half[c] + dw_half[D_MODEL + c];
    gmax = wave_max_f32(gmax);
    if (lane == 0) atomic_max_abs(amax_dx, gmax);
}

// --------------------------------------------------------------------------------- Adam
// torch.optim.Adam with L2 weight decay (CVRP/train.py:49-54, CVRP/CVRPTrainer.py:272):
//   g += wd p;  m = m + (1-b1)(g - m);  v = b2 v + (1-b2) g g;
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                            float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        if (wd != 0.0f) gi = fmaf(wd, p[i], gi);
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);
        const float vi = fmaf(1.0f - b2, gi * gi, b2 * v[i]);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

extern "C" int dgl_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t step,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                             dgl_stream_t stream) {
    DGL_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                       params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    DGL_CHECK_LAUNCH("dgl_adam_step");
    return DGL_OK;
}

// ----------------------------------------------------------------------- orchestration
// Workspace (floats), M = R*n tokens:
//   magnitude bounds [64][16 parts, 128 bytes apart] | T [M][12] | X[0..4] [M][128] (input of block l; X[4] = final) | Ea [2][R][128] (encoder
//   output at the anchors, before their re-embedding) | per block: Q K V AO O1 [M][128] each, H [M][512], attention row
//   statistics [R][8][n][2] | backward scratch: dA dA2 dA3 dB dQ dK dV [M][128] each, dH [M][512] | partial sums of one
//   block's weight and bias gradients | packed weights
#define AMAX_SLOTS 64
#define AMAX_FLOATS (AMAX_SLOTS * AMAX_STRIDE)
enum {
    AX_X = 0,        // +l, l = 0..4 (X[1]'s slot also bounds E)
    AX_AO = 5,       // +l
    AX_O1 = 9,       // +l
    AX_H = 13,       // +l
    AX_DX = 17,      // +l, l = 0..4: gradient with respect to X[l]
    AX_DMID = 22,    // +l: gradient with respect to O1[l]
    AX_DB = 26,      // +l: gradient with respect to AO[l]
    AX_DQKV = 30,    // +3l: dQ, dK, dV of block l
    AX_DH = 42,      // +l
    AX_QKV = 46,     // +3l: Q, K, V of block l
};
// packed weights of the four blocks, for the forward products and (transposed) for the dX products: hi + lo halves = 4 bytes a weight
#define BLOCK_WEIGHTS (4 * D_MODEL * D_MODEL + 2 * D_MODEL * FF_DIM)
#define PACKED_FLOATS (2 * 4 * BLOCK_WEIGHTS)
static size_t part_floats(int R, int n) {
    const size_t M = (size_t)R * n;
    const size_t zm = (M + DW_CHUNK - 1) / DW_CHUNK, zr = ((size_t)R + DW_CHUNK - 1) / DW_CHUNK;
    // one block: dW2, dW1 (128x512 each), dWc, dWq, dWk, dWv (128x128 each), db2, dbc (128), db1 (512); with it at most:
    // the two anchor layers (dW over rows, db per row), the output stage's per-row terms, the embedding layer
    return zm * (2 * (size_t)D_MODEL * FF_DIM + 4 * (size_t)D_MODEL * D_MODEL) + zm * (2 * D_MODEL + FF_DIM) +
           2 * zr * D_MODEL * D_MODEL + 2 * (size_t)R * D_MODEL + (size_t)R * (D_MODEL + 4) + zm * D_MODEL * DGL_TILE_STRIDE +
           zm * D_MODEL;
}
static size_t ws_floats(int R, int n) {
    const size_t M = (size_t)R * n;
    const size_t per_block = M * (5 * D_MODEL + FF_DIM) + (size_t)R * HEADS * n * 2;
    return AMAX_FLOATS + M * DGL_TILE_STRIDE + 5 * M * D_MODEL + 2 * (size_t)R * D_MODEL + 4 * per_block + 7 * M * D_MODEL + M * FF_DIM +
           part_floats(R, n) + PACKED_FLOATS;
}
extern "C" int64_t dgl_policy_general_workspace_floats(int32_t R, int32_t n) { return (int64_t)ws_floats(R, n); }

static int check_general(const dgl_policy_weights_t *w, const dgl_policy_io_t *io, const float *ws) {
    DGL_CHECK_ARG(w != nullptr && io != nullptr && ws != nullptr, "NULL argument");
    DGL_CHECK_ARG(io->problem == w->problem, "io and weights disagree on the problem");
    DGL_CHECK_ARG(io->R > 0 && io->n >= 2 && io->n <= 256, "R=%d, n=%d (2 <= n <= 256)", io->R, io->n);
    DGL_CHECK_ARG(io->tile != nullptr && io->idx_last != nullptr, "tile / idx_last missing");
    DGL_CHECK_ARG(io->tile_row_stride >= io->n * DGL_TILE_STRIDE, "tile_row_stride too small");
    const int stride = io->out_stride > 0 ? io->out_stride : DGL_MAX_TOKENS;
    DGL_CHECK_ARG((!io->logits_out && !io->prob_out) || stride >= io->n, "out_stride %d < n %d", stride, io->n);
    if (io->problem == DGL_PROBLEM_TSP) {
        DGL_CHECK_ARG(io->k_last == io->n - 1, "TSP needs k_last == n-1");
    } else {
        DGL_CHECK_ARG(io->k_last >= 1 && io->k_last <= io->n - 3 && io->idx_depot != nullptr, "CVRP token layout invalid");
    }
    return DGL_OK;
}

// Partial sums of one stage of the backward pass: space handed out from the workspace's last region, the ordered
// sums queued, everything added to the gradient buffers by one launch at the end of the stage.
struct PartialSums {
    float *base, *next, *end;
    ReduceLaunch red;
    PartialSums(float *b, size_t floats) : base(b), next(b), end(b + floats) {}
    // space for Z parts of `count` floats, to be summed into dst
    float *take(float *dst, int Z, long count) {
        float *p = next;
        next += (size_t)Z * count;
        red.add(p, dst, Z, count, count);
        return p;
    }
    bool fits() const { return next <= end && red.b.count <= REDUCE_BATCH_MAX; }
    void flush(hipStream_t s) {
        red.run(s);
        red = ReduceLaunch();
        next = base;
    }
};
extern "C" int dgl_policy_general(const dgl_policy_weights_t *w, const dgl_policy_weights_t *grad,
                                  const dgl_policy_io_t *io, const int32_t *teacher, float grad_scale, float *ws,
                                  float *teacher_prob_out, dgl_stream_t stream) {
    if (int e = check_general(w, io, ws)) return e;
    DGL_CHECK_ARG(grad == nullptr || teacher != nullptr, "gradients need the teacher's actions");
    DGL_CHECK_ARG((reinterpret_cast<size_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
    read_switches();
    hipStream_t s = (hipStream_t)stream;
    const int R = io->R, n = io->n, F = io->problem == DGL_PROBLEM_TSP ? 7 : 9;
    const int M = R * n;
    const size_t MD = (size_t)M * D_MODEL, MF = (size_t)M * FF_DIM, PP = (size_t)R * HEADS * n * 2;
    float *ax = ws;
    auto bound = [ax](int t) { return ax + (size_t)t * AMAX_STRIDE; };      // the t-th magnitude bound
    float *T = ws + AMAX_FLOATS;
    float *X[5];
    float *p = T + (size_t)M * DGL_TILE_STRIDE;
    for (int l = 0; l < 5; ++l) { X[l] = p; p += MD; }
    float *Ea = p; p += (size_t)2 * R * D_MODEL;
    float *Qb[4], *Kb[4], *Vb[4], *AO[4], *O1[4], *H[4], *P[4];
    for (int l = 0; l < 4; ++l) {
        Qb[l] = p; p += MD; Kb[l] = p; p += MD; Vb[l] = p; p += MD; AO[l] = p; p += MD; O1[l] = p; p += MD;
        H[l] = p; p += MF; P[l] = p; p += PP;
    }
    float *dA = p; p += MD;
    float *dA2 = p; p += MD;
    float *dA3 = p; p += MD;
    float *dB = p; p += MD;
    float *dQ = p; p += MD;
    float *dK = p; p += MD;
    float *dV = p; p += MD;
    float *dH = p; p += MF;
    PartialSums ps(p, part_floats(R, n));
    p += part_floats(R, n);
    const bool train = grad != nullptr;
    // packed weights: [block][wq wk wv wc w1 w2] for Y = X W^T, then the same for dX = dY W
    _Float16 *packed = reinterpret_cast<_Float16 *>(p);
    const _Float16 *pk[2][4][6];
    {
        PackBatch pb;
        pb.count = 0;
        int blocks = 0;
        _Float16 *dst = packed;
        for (int tr = 0; tr < (train ? 2 : 1); ++tr)
            for (int l = 0; l < 4; ++l) {
                const dgl_block_weights_t &k = w->block[l];
                const float *W[6] = {k.wq, k.wk, k.wv, k.wc, k.w1, k.w2};
                const int rows[6] = {D_MODEL, D_MODEL, D_MODEL, D_MODEL, FF_DIM, D_MODEL}, cols[6] = {D_MODEL, D_MODEL, D_MODEL, D_MODEL, D_MODEL, FF_DIM};
                for (int m = 0; m < 6; ++m) {
                    // W [rows][cols]: Y = X W^T reduces over cols, dX = dY W over rows
                    PackItem it{W[m], dst, tr ? rows[m] : cols[m], tr ? cols[m] : rows[m], cols[m], tr};
                    pb.it[pb.count] = it;
                    pb.first[pb.count++] = blocks;
                    blocks += (it.Jd / 32) * (it.Kd / 16) * 64 / 256;
                    pk[tr][l][m] = dst;
                    dst += 2 * (size_t)rows[m] * cols[m];
                }
            }
        pb.first[pb.count] = blocks;
        static_assert(AMAX_FLOATS <= 4 * 96 * 256, "the packing launch has fewer threads than there are bounds to clear");
        hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, pb, ax, (int)AMAX_FLOATS);
    }
    // one product of the activation x weight kernel
    auto aw_product = [](const _Float16 *Bp, float *C, float *amax_c, int ldc) {
        AwProduct P{};
        P.Bp[0] = Bp; P.C = C; P.amax_c = amax_c; P.ldc = ldc;
        return P;
    };
    auto aw_args = [M](const float *A, const float *amax_a, int lda, int N) {
        AwArgs a{};
        a.A[0] = A; a.amax_a[0] = amax_a; a.nseg = 1; a.lda = lda; a.M = M; a.N = N; a.nprod = 1;
        return a;
    };
    const int anchor0 = io->k_last, anchor1 = n - 1;
    const bool cvrp = io->problem == DGL_PROBLEM_CVRP;
    const int n_anchor = cvrp ? 2 : 1;
    const int NT = (n + 31) / 32;                      // attention: 32-token tiles, one wave each
    const size_t attn_fwd_lds = attn_lds_bytes(NT, 1, 1, 0), attn_q_lds = attn_lds_bytes(NT, 2, 1, 0),
                 attn_kv_lds = attn_lds_bytes(NT, 2, 2, 3 * 32 * NT + 4);
    dgl_max_dynamic_lds((const void *)attn_bwd_kernel, (int)attn_lds_bytes(8, 2, 2, 3 * 256 + 4));

    // ---- forward
    bool qkv_done = false;                             // this block's Q, K, V came out of the previous launch
    if (ff_fused_enabled()) {
        // the live tokens, their embedding and the first block's Q, K, V in one launch
        EmbedArgs e{io->tile, io->tile_row_stride, n, F, M, w->emb_w, w->emb_b, train ? T : nullptr, X[0], bound(AX_X),
                    pk[0][0][0], pk[0][0][1], pk[0][0][2], Qb[0], Kb[0], Vb[0], bound(AX_QKV)};
        hipLaunchKernelGGL(embed_qkv_kernel, dim3((M + 31) / 32), dim3(256), EMBED_LDS_BYTES, s, e);
        qkv_done = true;
    } else {
        hipLaunchKernelGGL(gather_tile_kernel, dim3(min((M * DGL_TILE_STRIDE + 255) / 256, 2048)), dim3(256), 0, s, io->tile,
                           io->tile_row_stride, R, n, T);
        GemmLaunch()
            .add(linear_fwd(T, nullptr, DGL_TILE_STRIDE, w->emb_w, w->emb_b, X[0], bound(AX_X), D_MODEL, M, D_MODEL, F, false, nullptr, 0))
            .run(s);
    }
    for (int l = 0; l < 4; ++l) {
        const dgl_block_weights_t &k = w->block[l];
        const float *Xin = X[l], *ax_in = bound(AX_X + l);
        if (l == 1) {
            // X[1] = the encoder output with re-embedded anchors; Ea keeps the anchors' encoder output
            hipLaunchKernelGGL(anchor_fwd_kernel, dim3(R, n_anchor), dim3(128), 0, s, X[1], Ea, n, anchor0, anchor1,
                               w->anchor_w[0], w->anchor_b[0], w->anchor_w[cvrp ? 1 : 0], w->anchor_b[cvrp ? 1 : 0], bound(AX_X + 1));
        }
        if (!qkv_done) {   // Q, K, V: three products with the same A
            AwArgs a = aw_args(Xin, ax_in, D_MODEL, D_MODEL);
            a.p[0] = aw_product(pk[0][l][0], Qb[l], bound(AX_QKV + 3 * l), D_MODEL);
            a.p[1] = aw_product(pk[0][l][1], Kb[l], bound(AX_QKV + 3 * l + 1), D_MODEL);
            a.p[2] = aw_product(pk[0][l][2], Vb[l], bound(AX_QKV + 3 * l + 2), D_MODEL);
            a.nprod = 3;
            launch_aw(s, a, D_MODEL);
        }
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(R, HEADS), dim3(64 * NT), attn_fwd_lds, s, Qb[l], Kb[l], Vb[l], n, bound(AX_QKV + 3 * l), P[l],
                           AO[l], bound(AX_AO + l));
        if (ff_fused_enabled()) {
            // O1 = Xin + Wc AO + bc and X[l+1] = O1 + W2 relu(W1 O1 + b1) + b2 in one launch; O1 and the hidden layer are
            // kept only for the backward pass
            FfArgs f{AO[l], bound(AX_AO + l), Xin, pk[0][l][3], pk[0][l][4], pk[0][l][5], k.bc, k.b1, k.b2,
                     train ? O1[l] : nullptr, train ? bound(AX_O1 + l) : nullptr, train ? H[l] : nullptr,
                     train ? bound(AX_H + l) : nullptr, X[l + 1], bound(AX_X + l + 1),
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M};
            // the next block's Q, K, V ride along - except into block 1, whose input has its anchors re-embedded first
            qkv_done = l + 1 < 4 && l + 1 != 1;
            if (qkv_done) {
                f.Wqp = pk[0][l + 1][0]; f.Wkp = pk[0][l + 1][1]; f.Wvp = pk[0][l + 1][2];
                f.Q = Qb[l + 1]; f.K = Kb[l + 1]; f.V = Vb[l + 1]; f.amax_qkv = bound(AX_QKV + 3 * (l + 1));
            }
            hipLaunchKernelGGL(ff_fused_kernel, dim3((M + 31) / 32), dim3(256), FF_LDS_BYTES, s, f);
            continue;
        }
        {   // O1 = Xin + Wc AO + bc
            AwArgs a = aw_args(AO[l], bound(AX_AO + l), D_MODEL, D_MODEL);
            a.p[0] = aw_product(pk[0][l][3], O1[l], bound(AX_O1 + l), D_MODEL);
            a.p[0].bias = k.bc; a.p[0].add = Xin; a.p[0].ldadd = D_MODEL;
            launch_aw(s, a, D_MODEL);
        }
        {   // H = relu(W1 O1 + b1)
            AwArgs a = aw_args(O1[l], bound(AX_O1 + l), D_MODEL, FF_DIM);
            a.p[0] = aw_product(pk[0][l][4], H[l], bound(AX_H + l), FF_DIM);
            a.p[0].bias = k.b1; a.p[0].relu = 1;
            launch_aw(s, a, D_MODEL);
        }
        {   // X[l+1] = O1 + W2 H + b2
            AwArgs a = aw_args(H[l], bound(AX_H + l), FF_DIM, D_MODEL);
            a.p[0] = aw_product(pk[0][l][5], X[l + 1], bound(AX_X + l + 1), D_MODEL);
            a.p[0].bias = k.b2; a.p[0].add = O1[l]; a.p[0].ldadd = D_MODEL;
            launch_aw(s, a, FF_DIM);
        }
    }
    float *dwf_part = nullptr, *dbf_part = nullptr;
    if (train) {
        dwf_part = ps.take((float *)grad->final_w, R, D_MODEL);
        dbf_part = ps.take((float *)grad->final_b, R, 1);
    }
    dgl_max_dynamic_lds((const void *)output_kernel, (256 * (D_MODEL + 1) + 512) * (int)sizeof(float));
    hipLaunchKernelGGL(output_kernel, dim3(R), dim3(256), ((size_t)n * (D_MODEL + 1) + n + 256) * sizeof(float), s, X[4], w->final_w, w->final_b, *io,
                       teacher, grad_scale, teacher_prob_out, train ? dA : nullptr, dwf_part, dbf_part, bound(AX_DX + 4));
    DGL_CHECK_LAUNCH("dgl_policy_general (forward)");
    if (!train) return DGL_OK;

    // ---- backward: dA holds the gradient of the current block's output
    const int zm = (M + DW_CHUNK - 1) / DW_CHUNK, zr = (R + DW_CHUNK - 1) / DW_CHUNK;
    for (int l = 3; l >= 0; --l) {
        const dgl_block_weights_t &k = w->block[l];
        const dgl_block_weights_t &g = grad->block[l];
        const float *Xin = X[l];
        const float *ax_out = bound(AX_DX + l + 1), *ax_h = bound(AX_DH + l), *ax_q = bound(AX_DQKV + 3 * l);
        float *ax_mid = bound(AX_DMID + l);
        // The dX products run on the packed transposed weights.  The weight-gradient products (which also sum their dY
        // operand over the tokens: the bias gradient) depend on nothing the chain writes after them, so each batch of
        // them shares a launch with the chain's next product.  dA holds the gradient of the block's output, dA2 that
        // of O1 (the batch reading dA runs beside the product writing dA2), and the block's input gradient goes back
        // into dA.
        if (ff_fused_enabled()) {
            // dH, dO1 (dA2) and dAO (dB) in one launch; after the attention backward all six weight gradients of the
            // block beside the product that writes the block's input gradient - into the other of two buffers, since
            // the gradient of W2 still reads the block's output gradient
            FfBwdArgs fb{dA, ax_out, H[l], pk[1][l][5], pk[1][l][4], pk[1][l][3], dH, bound(AX_DH + l), dA2, ax_mid, dB, bound(AX_DB + l), M};
            hipLaunchKernelGGL(ff_bwd_kernel, dim3((M + 31) / 32), dim3(256), FF_LDS_BYTES, s, fb);
            hipLaunchKernelGGL(attn_bwd_kernel, dim3(R, HEADS, 2), dim3(64 * NT), attn_kv_lds > attn_q_lds ? attn_kv_lds : attn_q_lds, s, Qb[l],
                               Kb[l], Vb[l], AO[l], dB, n, bound(AX_QKV + 3 * l), bound(AX_DB + l), P[l], dQ, dK, dV, bound(AX_DQKV + 3 * l));
            float *b2_part = ps.take((float *)g.b2, zm, D_MODEL), *b1_part = ps.take((float *)g.b1, zm, FF_DIM);
            float *bc_part = ps.take((float *)g.bc, zm, D_MODEL);
            GemmLaunch gl;
            gl.add(linear_bwd_w(dA, ax_out, D_MODEL, H[l], bound(AX_H + l), FF_DIM, ps.take((float *)g.w2, zm, D_MODEL * FF_DIM), b2_part, M,
                                D_MODEL, FF_DIM))
                .add(linear_bwd_w(dH, ax_h, FF_DIM, O1[l], bound(AX_O1 + l), D_MODEL, ps.take((float *)g.w1, zm, FF_DIM * D_MODEL), b1_part, M,
                                  FF_DIM, D_MODEL))
                .add(linear_bwd_w(dA2, ax_mid, D_MODEL, AO[l], bound(AX_AO + l), D_MODEL, ps.take((float *)g.wc, zm, D_MODEL * D_MODEL), bc_part, M,
                                  D_MODEL, D_MODEL))
                .add(linear_bwd_w(dQ, ax_q, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wq, zm, D_MODEL * D_MODEL), nullptr, M,
                                  D_MODEL, D_MODEL))
                .add(linear_bwd_w(dK, ax_q + AMAX_STRIDE, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wk, zm, D_MODEL * D_MODEL),
                                  nullptr, M, D_MODEL, D_MODEL))
                .add(linear_bwd_w(dV, ax_q + 2 * AMAX_STRIDE, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wv, zm, D_MODEL * D_MODEL),
                                  nullptr, M, D_MODEL, D_MODEL));
            AwArgs a = aw_args(dQ, ax_q, D_MODEL, D_MODEL);
            a.A[1] = dK; a.amax_a[1] = ax_q + AMAX_STRIDE;
            a.A[2] = dV; a.amax_a[2] = ax_q + 2 * AMAX_STRIDE;
            a.nseg = 3;
            a.p[0] = aw_product(pk[1][l][0], dA3, bound(AX_DX + l), D_MODEL);
            a.p[0].Bp[1] = pk[1][l][1]; a.p[0].Bp[2] = pk[1][l][2];
            a.p[0].add = dA2; a.p[0].ldadd = D_MODEL;
            if (g_grad_paired) {
                launch_pair(s, gl, a, D_MODEL);
            } else {
                gl.run(s);
                launch_aw(s, a, D_MODEL);
            }
            std::swap(dA, dA3);
        } else {
        // X[l+1] = O1 + W2 H + b2;  H = relu(..): dH = (dA W2) where H > 0
        {
            AwArgs a = aw_args(dA, ax_out, D_MODEL, FF_DIM);
            a.p[0] = aw_product(pk[1][l][5], dH, bound(AX_DH + l), FF_DIM);
            a.p[0].gate = H[l];
            launch_aw(s, a, D_MODEL);
        }
        {   // H = relu(W1 O1 + b1);  dO1 = dA + dHpre W1, beside dW2 and dW1
            float *b2_part = ps.take((float *)g.b2, zm, D_MODEL), *b1_part = ps.take((float *)g.b1, zm, FF_DIM);
            GemmLaunch gl;
            gl.add(linear_bwd_w(dA, ax_out, D_MODEL, H[l], bound(AX_H + l), FF_DIM, ps.take((float *)g.w2, zm, D_MODEL * FF_DIM), b2_part, M,
                                D_MODEL, FF_DIM))
                .add(linear_bwd_w(dH, ax_h, FF_DIM, O1[l], bound(AX_O1 + l), D_MODEL, ps.take((float *)g.w1, zm, FF_DIM * D_MODEL), b1_part, M,
                                  FF_DIM, D_MODEL));
            AwArgs a = aw_args(dH, ax_h, FF_DIM, D_MODEL);
            a.p[0] = aw_product(pk[1][l][4], dA2, ax_mid, D_MODEL);
            a.p[0].add = dA; a.p[0].ldadd = D_MODEL;
            launch_pair(s, gl, a, FF_DIM);
        }
        {   // O1 = Xin + Wc AO + bc:  dB = dAO = dO1 Wc
            AwArgs a = aw_args(dA2, ax_mid, D_MODEL, D_MODEL);
            a.p[0] = aw_product(pk[1][l][3], dB, bound(AX_DB + l), D_MODEL);
            launch_aw(s, a, D_MODEL);
        }
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(R, HEADS, 2), dim3(64 * NT), attn_kv_lds > attn_q_lds ? attn_kv_lds : attn_q_lds, s, Qb[l],
                           Kb[l], Vb[l], AO[l], dB, n, bound(AX_QKV + 3 * l), bound(AX_DB + l), P[l], dQ, dK, dV, bound(AX_DQKV + 3 * l));
        {   // dXin = dO1 + dQ Wq + dK Wk + dV Wv (the three products accumulate in one tile), beside dWc, dWq, dWk, dWv
            float *bc_part = ps.take((float *)g.bc, zm, D_MODEL);
            GemmLaunch gl;
            gl.add(linear_bwd_w(dA2, ax_mid, D_MODEL, AO[l], bound(AX_AO + l), D_MODEL, ps.take((float *)g.wc, zm, D_MODEL * D_MODEL), bc_part, M,
                                D_MODEL, D_MODEL))
                .add(linear_bwd_w(dQ, ax_q, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wq, zm, D_MODEL * D_MODEL), nullptr, M,
                                  D_MODEL, D_MODEL))
                .add(linear_bwd_w(dK, ax_q + AMAX_STRIDE, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wk, zm, D_MODEL * D_MODEL),
                                  nullptr, M, D_MODEL, D_MODEL))
                .add(linear_bwd_w(dV, ax_q + 2 * AMAX_STRIDE, D_MODEL, Xin, bound(AX_X + l), D_MODEL, ps.take((float *)g.wv, zm, D_MODEL * D_MODEL),
                                  nullptr, M, D_MODEL, D_MODEL));
            AwArgs a = aw_args(dQ, ax_q, D_MODEL, D_MODEL);
            a.A[1] = dK; a.amax_a[1] = ax_q + AMAX_STRIDE;
            a.A[2] = dV; a.amax_a[2] = ax_q + 2 * AMAX_STRIDE;
            a.nseg = 3;
            a.p[0] = aw_product(pk[1][l][0], dA, bound(AX_DX + l), D_MODEL);
            a.p[0].Bp[1] = pk[1][l][1]; a.p[0].Bp[2] = pk[1][l][2];
            a.p[0].add = dA2; a.p[0].ldadd = D_MODEL;
            launch_pair(s, gl, a, D_MODEL);
        }
        }
        if (l == 1) {
            // dW_a[c][k] += sum_r dA[anchor of row r][c] E[anchor of row r][k]: the rows' anchor tokens are n tokens apart
            GemmLaunch gl;
            for (int a = 0; a < n_anchor; ++a) {
                const long off = (long)(a == 0 ? anchor0 : anchor1) * D_MODEL;
                gl.add(linear_bwd_w(dA + off, bound(AX_DX + 1), (long)n * D_MODEL, Ea + (size_t)a * R * D_MODEL, bound(AX_X + 1), D_MODEL,
                                    ps.take((float *)grad->anchor_w[a], zr, D_MODEL * D_MODEL), nullptr, R, D_MODEL, D_MODEL));
            }
            gl.run(s);
            float *db_part = ps.next;
            for (int a = 0; a < n_anchor; ++a) ps.take((float *)grad->anchor_b[a], R, D_MODEL);
            hipLaunchKernelGGL(anchor_bwd_kernel, dim3(R, n_anchor), dim3(128), 0, s, dA, n, anchor0, anchor1, w->anchor_w[0],
                               w->anchor_w[cvrp ? 1 : 0], db_part, bound(AX_DX + 1));
        }
        if (l == 0) {
            // X[0] = T We^T + be
            float *be_part = ps.take((float *)grad->emb_b, zm, D_MODEL);
            GemmLaunch()
                .add(linear_bwd_w(dA, bound(AX_DX), D_MODEL, T, nullptr, DGL_TILE_STRIDE, ps.take((float *)grad->emb_w, zm, D_MODEL * F),
                                  be_part, M, D_MODEL, F))
                .run(s);
        }
        if (!ps.fits()) return dgl_set_error(DGL_ERR_INVALID_ARGUMENT, "partial-sum region of the workspace too small (internal)");
        ps.flush(s);
    }
    DGL_CHECK_LAUNCH("dgl_policy_general (backward)");
    return DGL_OK;
}

// ------------------------------------------------------------ allocation-free variant
// dgl_policy_general for callers that cannot allocate device memory (HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): the workspace is a file-scope device array instead of an
// argument.  Every device the library runs on holds its own copy of the array at its own address
// until the process ends; the host learns that address once per device (hipGetSymbolAddress with the
// device current) and hands it to the kernels exactly as dgl_policy_general hands them `ws`.  The
// array's name itself is never passed to a kernel or a copy from host code.
// All calls on one device share the array: calls that overlap in time on one device race.  The only
// part of the workspace that has to start at zero, the magnitude bounds at its head, is cleared by
// every call; every other buffer is written in full before it is read, so the previous call's
// contents are simply overwritten.
static __device__ __align__(256) float scratch[DGL_STATIC_WORKSPACE_FLOATS];

#define DGL_MAX_DEVICES 64
static float *scratch_ptr[DGL_MAX_DEVICES];          // per device, filled on the first call on that device
static std::mutex scratch_mutex;

static int static_workspace(hipStream_t stream, float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    // the array handed to the kernels is the current device's copy: a stream of another device would run them on a
    // device whose own copy lives elsewhere
    if (stream != nullptr) {
        int sdev = dev;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "hipStreamGetDevice: %s", hipGetErrorString(e));
        DGL_CHECK_ARG(sdev == dev, "the stream belongs to device %d, the current device is %d", sdev, dev);
    }
    if (dev < 0 || dev >= DGL_MAX_DEVICES) return dgl_set_error(DGL_ERR_UNSUPPORTED, "device ordinal %d beyond %d", dev, DGL_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr)
            return dgl_set_error(DGL_ERR_LAUNCH, "hipGetSymbolAddress(static workspace) on device %d: %s", dev,
                                 hipGetErrorString(e));
        scratch_ptr[dev] = static_cast<float *>(p);
    }
    *out = scratch_ptr[dev];
    return DGL_OK;
}

extern "C" int dgl_policy_general_static(const dgl_policy_weights_t *w, const dgl_policy_weights_t *grad,
                                         const dgl_policy_io_t *io, const int32_t *teacher, float grad_scale,
                                         float *teacher_prob_out, dgl_stream_t stream) {
    DGL_CHECK_ARG(io != nullptr, "NULL argument");
    DGL_CHECK_ARG(io->R > 0 && io->n >= 2 && io->n <= 256, "R=%d, n=%d (2 <= n <= 256)", io->R, io->n);
    const size_t need = ws_floats(io->R, io->n);
    DGL_CHECK_ARG(need <= (size_t)DGL_STATIC_WORKSPACE_FLOATS,
                  "R=%d rows of n=%d tokens need %zu floats of workspace, the static workspace holds %zu "
                  "(use dgl_policy_general with a caller-owned workspace)",
                  io->R, io->n, need, (size_t)DGL_STATIC_WORKSPACE_FLOATS);
    float *ws = nullptr;
    if (int e = static_workspace((hipStream_t)stream, &ws)) return e;
    return dgl_policy_general(w, grad, io, teacher, grad_scale, ws, teacher_prob_out, stream);
}
