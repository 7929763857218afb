// Body of the split-f16 policy kernel (see policy16.hip for the description): a device function, so that the same
// code runs as a kernel of its own (policy16.hip) and inside the fused decode kernel (rollout.hip).
#pragma once
#include "policy_layout.h"
#include "split_f16.h"

#define LDH 136                               // LDS row stride in halfs
#define ARR (32 * LDH)                        // halfs per [32 tokens][128 features] image
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f16x8 as_h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

#define NHEAD PK_STREAM0                         // embedding + bias vectors + final weights: staged in LDS (multiple of 4)

// Range guard: the largest operand magnitude this lane has split, as binary32 bits of a non-negative float (one
// v_max3_f32 with |.| source modifiers per pair of values).  65504 is where the round-towards-zero conversion to
// binary16 saturates.
// Attention products (K Q^T and P V) as three split-operand passes like every other product (3), or as a single
// binary16 pass with operands rounded to nearest (1): 11 significant bits per operand in a sum of 16 (scores) or 32
// (weighted values) terms.
#ifndef DGL_ATTN_PASSES
#define DGL_ATTN_PASSES 3
#endif
#ifndef DGL_W2_PASSES
#define DGL_W2_PASSES 3                      // 2: the second feed-forward matrix enters rounded to binary16 (experiment)
#endif
#ifndef DGL_RANGE_GUARD
#define DGL_RANGE_GUARD 1
#endif
#ifndef DGL_PIN_PRODUCTS
#define DGL_PIN_PRODUCTS 0                   // 1: the 32-token kernel's products are pinned as well (see gemm16_t, FENCE)
#endif
__device__ __forceinline__ void track_range(float a, float b, float &mx) {
#if DGL_RANGE_GUARD
    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(a)), __builtin_fabsf(b));
#endif
}
// registers r0..r0+7 of an accumulator tile as one MFMA operand fragment (hi and lo)
__device__ __forceinline__ void split8(const f32x16 &x, int r0, f16x8 &hi, f16x8 &lo, float &mx) {
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned a, b;
        split2(x[r0 + 2 * p], x[r0 + 2 * p + 1], a, b);
        track_range(x[r0 + 2 * p], x[r0 + 2 * p + 1], mx);
        h[p] = a;
        l[p] = b;
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// registers r0..r0+7 as ONE binary16 fragment, rounded to nearest (v_cvt_pk_f16_f32): the single-pass products
__device__ __forceinline__ f16x8 round8(const f32x16 &x, int r0, float &mx) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    u32x4 h;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        h[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x[r0 + 2 * p], x[r0 + 2 * p + 1]}, h2));
        track_range(x[r0 + 2 * p], x[r0 + 2 * p + 1], mx);
    }
    return __builtin_bit_cast(f16x8, h);
}

// hi/lo images of a transposed-form tile (lane = token, register r = feature base + (r&3) + 8(r>>2)
// of this lane's half): dst points at [token][feature base + 4h]
__device__ __forceinline__ void store_split(const f32x16 &x, _Float16 *H, _Float16 *L, float &mx) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 hi, lo;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned a, b;
            split2(x[4 * g + 2 * p], x[4 * g + 2 * p + 1], a, b);
            track_range(x[4 * g + 2 * p], x[4 * g + 2 * p + 1], mx);
            hi[p] = a;
            lo[p] = b;
        }
        *reinterpret_cast<u32x2 *>(H + 8 * g) = hi;
        *reinterpret_cast<u32x2 *>(L + 8 * g) = lo;
    }
}

// acc[rr] += W(out tile) * X[rr]^T over K = 128 (8 steps of 16): transposed product, lane = token.
// H, L: this lane's operand pointer into the hi / lo image of row 0 ([token q][8h]); rows ARRS apart.
// BASE: position of the GEMM's first fragment in the wave's stream, modulo the ring depth.
// PASSES = 2 leaves the weights' low halves out (W rounded to binary16, the activation still split): an experiment
// switch for single GEMMs (DGL_W2_PASSES), the stream is consumed the same way.
// FENCE: every K-step's matrix instructions stay where they are written.  Matrix instructions have no side effects, so
// in a kernel whose accumulators are read much later the compiler may sink a whole product's instructions past the next
// product's loads - keeping every operand fragment of both alive, far more than the register file holds.  An empty
// assembly statement that "updates" the accumulators after each step (no instruction is emitted) ties them to their
// place; a compiler-level memory fence does the same for the loads.
__device__ __forceinline__ void pin_acc(f32x16 &acc) { asm volatile("" : "+a"(acc)); }
template <int ROWS, int ARRS, int BASE, int PASSES = 3, bool FENCE = (DGL_PIN_PRODUCTS != 0)>
__device__ __forceinline__ void gemm16_t(WeightRing<PF16> &ring, const _Float16 *H, const _Float16 *L,
                                         f32x16 (&acc)[ROWS]) {
    // activation operands are read one K-step ahead of the MFMAs that consume them
    f16x8 nh[ROWS], nl[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        nh[rr] = *reinterpret_cast<const f16x8 *>(H + rr * ARRS);
        nl[rr] = *reinterpret_cast<const f16x8 *>(L + rr * ARRS);
    }
    Unroll<0, 8>::run([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if constexpr (FENCE) asm volatile("" ::: "memory");
        const f16x8 wh = as_h8(ring.template take<(BASE + 2 * s) % PF16>());
        const f16x8 wl = as_h8(ring.template take<(BASE + 2 * s + 1) % PF16>());
        f16x8 xh[ROWS], xl[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) { xh[rr] = nh[rr]; xl[rr] = nl[rr]; }
        if constexpr (s + 1 < 8) {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                nh[rr] = *reinterpret_cast<const f16x8 *>(H + rr * ARRS + 16 * (s + 1));
                nl[rr] = *reinterpret_cast<const f16x8 *>(L + rr * ARRS + 16 * (s + 1));
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the operand reads ahead of this step's MFMAs
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            if constexpr (PASSES == 3) acc[rr] = MFMA16(wl, xh[rr], acc[rr]);
            acc[rr] = MFMA16(wh, xl[rr], acc[rr]);
            acc[rr] = MFMA16(wh, xh[rr], acc[rr]);
            if constexpr (FENCE) pin_acc(acc[rr]);
        }
    });
}

// The same GEMM with independent VALU / LDS-store work placed in the shadow of its MFMAs: one wave
// per SIMD issues in order, an MFMA occupies the issue port for 8 of its 32 cycles, so up to five
// 4-cycle VALU instructions written between two MFMAs cost nothing.  side(step, row, part) is called
// after MFMA `part` (0..2) of `row` in K-step `step` and must not depend on acc; scheduling barriers
// pin the written order.
template <int ROWS, int ARRS, int BASE, int PASSES = 3, bool FENCE = (DGL_PIN_PRODUCTS != 0), class Side>
__device__ __forceinline__ void gemm16_t_with(WeightRing<PF16> &ring, const _Float16 *H, const _Float16 *L,
                                              f32x16 (&acc)[ROWS], Side side) {
    f16x8 nh[ROWS], nl[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        nh[rr] = *reinterpret_cast<const f16x8 *>(H + rr * ARRS);
        nl[rr] = *reinterpret_cast<const f16x8 *>(L + rr * ARRS);
    }
    Unroll<0, 8>::run([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if constexpr (FENCE) asm volatile("" ::: "memory");
        const f16x8 wh = as_h8(ring.template take<(BASE + 2 * s) % PF16>());
        const f16x8 wl = as_h8(ring.template take<(BASE + 2 * s + 1) % PF16>());
        f16x8 xh[ROWS], xl[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) { xh[rr] = nh[rr]; xl[rr] = nl[rr]; }
        if constexpr (s + 1 < 8) {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                nh[rr] = *reinterpret_cast<const f16x8 *>(H + rr * ARRS + 16 * (s + 1));
                nl[rr] = *reinterpret_cast<const f16x8 *>(L + rr * ARRS + 16 * (s + 1));
            }
        }
        Unroll<0, 3>::run([&](auto pp) {
            constexpr int part = decltype(pp)::value;
            Unroll<0, ROWS>::run([&](auto rc) {
                constexpr int rr = decltype(rc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (part == 0 && PASSES == 3) acc[rr] = MFMA16(wl, xh[rr], acc[rr]);
                if constexpr (part == 1) acc[rr] = MFMA16(wh, xl[rr], acc[rr]);
                if constexpr (part == 2) acc[rr] = MFMA16(wh, xh[rr], acc[rr]);
                if constexpr (FENCE) pin_acc(acc[rr]);
                __builtin_amdgcn_sched_barrier(0);
                side(ss, rc, pp);
            });
        });
    });
    __builtin_amdgcn_sched_barrier(0);
}

// PROFILE_STAMPS=1: one stamp per phase of every block; =2: fine stamps inside block 2 only (FF per chunk)
#if defined(PROFILE_STAMPS) && PROFILE_STAMPS == 2
#define STAMP() do { } while (0)
#define STAMP2() do { if (tid == 0 && layer == 2 && nst < 32) st_[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
#elif defined(PROFILE_STAMPS)
#define STAMP() do { if (tid == 0 && nst < 32) st_[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP2() do { } while (0)
#else
#define STAMP() do { } while (0)
#define STAMP2() do { } while (0)
#endif

// The network for the rows of one workgroup: rows (wg * GROUPS + grp) * ROWS ...  `lds`: the workgroup's dynamic LDS
// (policy16_lds_bytes<ROWS, GROUPS>()); called by all 256 * GROUPS threads (it contains workgroup barriers).
// GROUPS = 2: a workgroup of 8 waves, two per SIMD.  Waves w and w + 4 own the same output features (the same
// weight fragments) and work on different rows; they run the same program between the same barriers.  (Measured:
// no faster than GROUPS = 1 with two rows per wave - the two waves of a SIMD do not speed each other's VALU phases
// up, and the weight stream through the vector L1 doubles.  Kept for experiments, DGL_POLICY_SHAPE=21.)
template <int ROWS, int GROUPS>
__device__ __forceinline__ void policy16_rows(const float *__restrict__ P, const dgl_policy_io_t &io, float *lds, int wg) {
    constexpr int ROWH = 6 * ARR;                                   // halfs per row: XH XL BH BL CH CL
    const int grp = GROUPS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
    _Float16 *base = reinterpret_cast<_Float16 *>(lds) + grp * ROWS * ROWH;
    float *part = lds + GROUPS * ROWS * ROWH / 2 + grp * 4 * ROWS * 32;   // [GROUPS][4][ROWS][32]
    float *headv = lds + GROUPS * ROWS * ROWH / 2 + GROUPS * 4 * ROWS * 32;  // [NHEAD] copy of P[0 .. PK_STREAM0)
    float *vecs = headv + PK_VEC0;                                           // the bias vectors and final weights in it
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int row0 = (wg * GROUPS + grp) * ROWS;
    const int n = io.n;
    // image pointers of row 0 (rows ROWH apart)
    _Float16 *XH = base, *XL = base + ARR, *BH = base + 2 * ARR, *BL = base + 3 * ARR, *CH = base + 4 * ARR,
             *CL = base + 5 * ARR;
    // this lane's view of a 128-float vector: registers 4g..4g+3 <-> vl[off + 8g .. 8g+3]
    const float *vl = vecs + 32 * w + 4 * h;
    auto vec4 = [&](int off, int g) { return *reinterpret_cast<const f32x4 *>(vl + off + 8 * g); };
    const int rd = q * LDH + 8 * h;                // operand read offset: [token q][8h]
    const int wr = q * LDH + 32 * w + 4 * h;       // split store offset: [token q][32w + 4h]

#ifdef PROFILE_STAMPS
    unsigned long long st_[32];
    int nst = 0;
#endif
    float range_mx = 0.0f;                             // range guard: largest magnitude this lane has split (track_range)
    STAMP();
    // Everything that is not a matrix of a block (embedding matrix and bias, the blocks' bias vectors, the anchors'
    // biases, the final layer: 20 KB) goes to LDS with 16-byte loads, requested first so that they return first; then
    // the rows' token tiles and the first fragments of the weight stream.  (Read straight from memory, the embedding
    // cost every lane 64 more loads and the whole phase 9.4k cycles.)
    static_assert(NHEAD % 4 == 0, "the staged head is copied in 16-byte pieces");
    WeightRing<PF16> ring;
    f32x16 xres[ROWS];
    {
        constexpr int PIECES = NHEAD / 4, PER = (PIECES + 256 * GROUPS - 1) / (256 * GROUPS);
        f32x4 hv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = threadIdx.x + e * 256 * GROUPS;
            hv[e] = reinterpret_cast<const f32x4 *>(P)[min(i, PIECES - 1)];
        }
        f32x4 t[ROWS][3];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int r = row0 + rr;
            const bool live = r < io.R && q < n;
            const f32x4 *tp = reinterpret_cast<const f32x4 *>(io.tile + (size_t)(live ? r : 0) * io.tile_row_stride +
                                                             (live ? q : 0) * DGL_TILE_STRIDE);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t[rr][i] = tp[i];
                if (!live) t[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        ring.fill(reinterpret_cast<const f32x4 *>(P + PK_STREAM16 + (size_t)w * WAVE_FRAGS * FRAG) + lane);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = threadIdx.x + e * 256 * GROUPS;
            if (i < PIECES) reinterpret_cast<f32x4 *>(headv)[i] = hv[e];
        }
        __syncthreads();

        // ---- embedding, directly in the residual layout: xres[rr][r] = X[token q][feature f(r)],
        //      f(r) = 32w + (r&3) + 8(r>>2) + 4h;  X = b + sum_f tile[token][f] * We[c][f]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(headv + PK_EMB_W + c * EMB_IN);
            const f32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
            const float be = headv[PK_EMB_B + c];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                float acc = 0.0f;
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[rr][0][f], w0[f], acc);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[rr][1][f], w1[f], acc);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[rr][2][f], w2[f], acc);
                const bool live = (row0 + rr) < io.R && q < n;
                xres[rr][r] = live ? acc + be : 0.0f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) store_split(xres[rr], XH + rr * ROWH + wr, XL + rr * ROWH + wr, range_mx);
    }
    __syncthreads();
    STAMP();

    const int anchor0 = io.k_last, anchor1 = n - 1;    // TSP: k_last == n-1

#ifndef DGL_DEBUG_LAYERS
#define DGL_DEBUG_LAYERS 4                  // debugging builds: stop after this many blocks and apply the final layer
#endif
    for (int layer = 0; layer < DGL_DEBUG_LAYERS; ++layer) {
        const int PV = layer * PKV_SIZE;                 // offset of this block's bias vectors in vecs
        STAMP2();  // block start

        if (layer == 1) {
            // decoder entry: re-embed the anchor token(s) (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-446)
            // (the stream always carries two embeddings; TSP's second equals its first and is discarded)
            f32x16 a0[ROWS], a1[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) { a0[rr] = zero16(); a1[rr] = zero16(); }
            gemm16_t<ROWS, ROWH, 0>(ring, XH + rd, XL + rd, a0);
            gemm16_t<ROWS, ROWH, 16>(ring, XH + rd, XL + rd, a1);
            __syncthreads();                           // every wave has finished reading X
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int tok = (a == 0) ? anchor0 : anchor1;
                if (q == tok && (a == 0 || io.problem == DGL_PROBLEM_CVRP)) {
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 bb = vec4(PK_ANCHOR_B - PK_VEC0 + a * D_MODEL, g);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                xres[rr][4 * g + e] = (a == 0 ? a0[rr][4 * g + e] : a1[rr][4 * g + e]) + bb[e];
                        }
                        store_split(xres[rr], XH + rr * ROWH + wr, XL + rr * ROWH + wr, range_mx);
                    }
                }
            }
            __syncthreads();
        }

        // ---- Q^T, K^T (transposed form) and V (normal form) for heads 2w, 2w+1 ----
        f32x16 aq[ROWS], ak[ROWS], av[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) { aq[rr] = zero16(); ak[rr] = zero16(); av[rr] = zero16(); }
        {
            f16x8 nh[ROWS], nl[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                nh[rr] = *reinterpret_cast<const f16x8 *>(XH + rr * ROWH + rd);
                nl[rr] = *reinterpret_cast<const f16x8 *>(XL + rr * ROWH + rd);
            }
            Unroll<0, 8>::run([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                const f16x8 qh = as_h8(ring.template take<(6 * s) % PF16>());
                const f16x8 ql = as_h8(ring.template take<(6 * s + 1) % PF16>());
                const f16x8 kh = as_h8(ring.template take<(6 * s + 2) % PF16>());
                const f16x8 kl = as_h8(ring.template take<(6 * s + 3) % PF16>());
                const f16x8 vh = as_h8(ring.template take<(6 * s + 4) % PF16>());
                const f16x8 vl = as_h8(ring.template take<(6 * s + 5) % PF16>());
                f16x8 xh[ROWS], xl[ROWS];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) { xh[rr] = nh[rr]; xl[rr] = nl[rr]; }
                if constexpr (s + 1 < 8) {
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        nh[rr] = *reinterpret_cast<const f16x8 *>(XH + rr * ROWH + rd + 16 * (s + 1));
                        nl[rr] = *reinterpret_cast<const f16x8 *>(XL + rr * ROWH + rd + 16 * (s + 1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    aq[rr] = MFMA16(ql, xh[rr], aq[rr]);
                    aq[rr] = MFMA16(qh, xl[rr], aq[rr]);
                    aq[rr] = MFMA16(qh, xh[rr], aq[rr]);
                    ak[rr] = MFMA16(kl, xh[rr], ak[rr]);
                    ak[rr] = MFMA16(kh, xl[rr], ak[rr]);
                    ak[rr] = MFMA16(kh, xh[rr], ak[rr]);
                    av[rr] = MFMA16(xh[rr], vl, av[rr]);
                    av[rr] = MFMA16(xl[rr], vh, av[rr]);
                    av[rr] = MFMA16(xh[rr], vh, av[rr]);
#if DGL_PIN_PRODUCTS
                    pin_acc(aq[rr]); pin_acc(ak[rr]); pin_acc(av[rr]);
#endif
                }
            });
        }

        STAMP();   // QKV done
        STAMP2();
        // ---- attention, both local heads, straight from the accumulators --------------------
        // Accumulator registers serve as split f16 operands without any lane movement: registers
        // 8t..8t+7 of a tile whose summed index runs over its rows are one K = 16 fragment, with the
        // same (permuted) k order in both operands.
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            f32x16 o_sel = zero16();
            float pmx = 0.0f;                                // (attention weights lie in [0, 1]: their range is not tracked)
#if DGL_ATTN_PASSES == 3
            f16x8 vh[2], vl[2];                              // V[key][feature]: keys summed, two K-steps
            split8(av[rr], 0, vh[0], vl[0], range_mx);
            split8(av[rr], 8, vh[1], vl[1], range_mx);
#else
            f16x8 vh[2];
            vh[0] = round8(av[rr], 0, range_mx);
            vh[1] = round8(av[rr], 8, range_mx);
#endif
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                // S^T[key j][query i] = sum_d K[j][d] Q[i][d]: the 16 d of this head are one K-step
                f32x16 s = zero16();
#if DGL_ATTN_PASSES == 3
                f16x8 kh, kl, qh, ql;
                split8(ak[rr], 8 * hd, kh, kl, range_mx);
                split8(aq[rr], 8 * hd, qh, ql, range_mx);
                s = MFMA16(kl, qh, s);
                s = MFMA16(kh, ql, s);
                s = MFMA16(kh, qh, s);
#else
                const f16x8 kh = round8(ak[rr], 8 * hd, range_mx), qh = round8(aq[rr], 8 * hd, range_mx);
                s = MFMA16(kh, qh, s);
#endif
                // lane (i, h) holds keys j = (m&3) + 8(m>>2) + 4h
                // softmax over keys; 1/4 = 1/sqrt(16) is folded into the exponent:
                //   exp((s - max)/4) = 2^((s - max) * log2(e)/4), one v_exp_f32 per key
                if (n < 32) {                                    // wave-uniform: a full tile has no padded key
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int j = (m & 3) + 8 * (m >> 2) + 4 * h;
                        s[m] = (j < n) ? s[m] : -INFINITY;
                    }
                }
                float mx = s[0];
#pragma unroll
                for (int m = 1; m < 16; ++m) mx = fmaxf(mx, s[m]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float off = -mx * 0.36067376022224085f;
                float sum = 0.0f;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    s[m] = __builtin_amdgcn_exp2f(fmaf(s[m], 0.36067376022224085f, off));
                    sum += s[m];
                }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
                // O^T[d][i] = sum_j V[j][d] P[i][j]; rows d of this head are registers 8hd..8hd+7.  The weights
                // enter unnormalised (largest = 1); the eight outputs are divided by their sum afterwards.
                f32x16 o = zero16();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#if DGL_ATTN_PASSES == 3
                    f16x8 ph, pl;
                    split8(s, 8 * t, ph, pl, pmx);
                    o = MFMA16(vl[t], ph, o);
                    o = MFMA16(vh[t], pl, o);
                    o = MFMA16(vh[t], ph, o);
#else
                    o = MFMA16(vh[t], round8(s, 8 * t, pmx), o);
#endif
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) o_sel[8 * hd + m] = o[8 * hd + m] * inv;
            }
            store_split(o_sel, BH + rr * ROWH + wr, BL + rr * ROWH + wr, range_mx);
        }
        STAMP();   // attention done
        STAMP2();
        __syncthreads();
        STAMP();   // barrier
        STAMP2();

        // ---- multi_head_combine + residual -> o1 -------------------------------------
        // (every wave passed the barrier above after its Q/K/V reads of X: X may be replaced)
        {
            f32x16 acc[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) acc[rr] = zero16();
            gemm16_t<ROWS, ROWH, 48>(ring, BH + rd, BL + rd, acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = vec4(PV + PKV_BC, g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) xres[rr][4 * g + e] = xres[rr][4 * g + e] + (acc[rr][4 * g + e] + bb[e]);
            }
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) store_split(xres[rr], XH + rr * ROWH + wr, XL + rr * ROWH + wr, range_mx);
        }
        STAMP();   // combine done
        STAMP2();
        __syncthreads();
        STAMP();   // barrier
        STAMP2();

        // ---- feed forward: 4 hidden chunks of 128, hidden images alternate between B and C ----
        // Software pipeline: the epilogue E(hc) of a chunk's first GEMM (bias, ReLU, split, LDS stores: VALU work)
        // is written between the MFMAs of GEMMs that do not depend on it - the second GEMM of the chunk before and
        // the first GEMM of the chunk after, half of it in each:
        //   G1(0) [G1(1) || E(0)] | [G2(0) || E(1)a] [G1(2) || E(1)b] | [G2(1) || E(2)a] [G1(3) || E(2)b] |
        //   [G2(2) || E(3)] | G2(3)                                   (| = workgroup barrier)
        // The weight stream is packed in this order: W1.0 W1.1 W2.0 W1.2 W2.1 W1.3 W2.2 W2.3.
#ifdef DGL_DEBUG_SKIP_FF
        if (layer != DGL_DEBUG_LAYERS - 1)
#endif
        {
            f32x16 af[ROWS], ahA[ROWS], ahB[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) { af[rr] = zero16(); ahA[rr] = zero16(); ahB[rr] = zero16(); }
            // registers of an epilogue in flight
            float v0[ROWS], v1[ROWS], t0[ROWS], t1[ROWS];
            unsigned hp[ROWS], lp[ROWS], hq[ROWS], lq[ROWS], lw0[ROWS], lw1[ROWS];
            f32x4 bb[4];
            auto load_bias = [&](int hc) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bb[g] = vec4(PV + PKV_B1 + 128 * hc, g);
            };
            // piece (st, rr, part) of the epilogue of one chunk: step st = 0..7 handles registers 2 st, 2 st + 1 of
            // row rr in three parts (bias + ReLU | hi | lo and, every second step, the 8-byte stores)
            auto piece = [&](auto stc, auto rc, auto pp, const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
                constexpr int st = decltype(stc)::value, rr = decltype(rc)::value, part = decltype(pp)::value;
                constexpr int g = st / 2, e = 2 * (st & 1);
                if constexpr (part == 0) {
                    v0[rr] = fmaxf(ah[rr][2 * st] + bb[g][e], 0.0f);
                    v1[rr] = fmaxf(ah[rr][2 * st + 1] + bb[g][e + 1], 0.0f);
                } else if constexpr (part == 1) {
                    hq[rr] = pack_rtz(v0[rr], v1[rr]);
                    track_range(v0[rr], v1[rr], range_mx);
                    t0[rr] = minus_half_lo(v0[rr], hq[rr]);
                    t1[rr] = minus_half_hi(v1[rr], hq[rr]);
                } else {
                    lq[rr] = pack_rtz(t0[rr], t1[rr]);
                    // one LDS store per gap (a second one costs 24 cycles beside an MFMA): the hi pair of steps
                    // (st - 1, st) goes out at the odd step, its lo pair one step later; both at the last step
                    if constexpr (st & 1) {
                        *reinterpret_cast<u32x2 *>(HH + rr * ROWH + wr + 8 * g) = u32x2{hp[rr], hq[rr]};
                        if constexpr (st == 7) {
                            *reinterpret_cast<u32x2 *>(HL + rr * ROWH + wr + 8 * g) = u32x2{lp[rr], lq[rr]};
                        } else {
                            lw0[rr] = lp[rr];
                            lw1[rr] = lq[rr];
                        }
                    } else {
                        if constexpr (st >= 2) *reinterpret_cast<u32x2 *>(HL + rr * ROWH + wr + 8 * (g - 1)) = u32x2{lw0[rr], lw1[rr]};
                        hp[rr] = hq[rr];
                        lp[rr] = lq[rr];
                    }
                }
            };
            // a whole epilogue inside one GEMM (one step per K-step), or half of one (a step on every second K-step)
            auto whole = [&](const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
                return [&, HH, HL](auto ss, auto rc, auto pp) { piece(ss, rc, pp, ah, HH, HL); };
            };
            auto half = [&](auto first, const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
                return [&, HH, HL](auto ss, auto rc, auto pp) {
                    constexpr int s = decltype(ss)::value, st0 = decltype(first)::value;
                    if constexpr ((s & 1) == 0) piece(std::integral_constant<int, st0 + s / 2>{}, rc, pp, ah, HH, HL);
                };
            };
            using I0 = std::integral_constant<int, 0>;
            using I4 = std::integral_constant<int, 4>;
            gemm16_t<ROWS, ROWH, 64>(ring, XH + rd, XL + rd, ahA);                                     // G1(0)
            STAMP2();
            load_bias(0);
            gemm16_t_with<ROWS, ROWH, 80>(ring, XH + rd, XL + rd, ahB, whole(ahA, BH, BL));            // G1(1) || E(0)
            STAMP2();
            __syncthreads();
            STAMP2();
            load_bias(1);
            gemm16_t_with<ROWS, ROWH, 96, DGL_W2_PASSES>(ring, BH + rd, BL + rd, af, half(I0{}, ahB, CH, CL));        // G2(0) || E(1)a
            STAMP2();
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) ahA[rr] = zero16();
            gemm16_t_with<ROWS, ROWH, 112>(ring, XH + rd, XL + rd, ahA, half(I4{}, ahB, CH, CL));       // G1(2) || E(1)b
            STAMP2();
            __syncthreads();
            STAMP2();
            load_bias(2);
            gemm16_t_with<ROWS, ROWH, 128, DGL_W2_PASSES>(ring, CH + rd, CL + rd, af, half(I0{}, ahA, BH, BL));        // G2(1) || E(2)a
            STAMP2();
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) ahB[rr] = zero16();
            gemm16_t_with<ROWS, ROWH, 144>(ring, XH + rd, XL + rd, ahB, half(I4{}, ahA, BH, BL));       // G1(3) || E(2)b
            STAMP2();
            __syncthreads();
            STAMP2();
            load_bias(3);
            gemm16_t_with<ROWS, ROWH, 160, DGL_W2_PASSES>(ring, BH + rd, BL + rd, af, whole(ahB, CH, CL));             // G2(2) || E(3)
            STAMP2();
            __syncthreads();
            STAMP2();
            // the last barrier came after every wave's last read of X (W1 of chunk 3): X may be replaced
            gemm16_t<ROWS, ROWH, 176, DGL_W2_PASSES>(ring, CH + rd, CL + rd, af);                                      // G2(3)
            STAMP2();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b2 = vec4(PV + PKV_B2, g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) xres[rr][4 * g + e] = xres[rr][4 * g + e] + (af[rr][4 * g + e] + b2[e]);
            }
            if (layer < 3) {
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) store_split(xres[rr], XH + rr * ROWH + wr, XL + rr * ROWH + wr, range_mx);
            }
        }
        STAMP();   // FF done
        STAMP2();
        __syncthreads();
        STAMP();   // barrier
        STAMP2();
    }
    // an operand reached the binary16 range somewhere in this row's network, or a weight did when it was packed: tell
    // the host (it repeats the work with the binary32 kernel); every writer stores the same value.  (The maxima of
    // track_range drop NaN operands; a NaN activation stays in the residual stream and shows in the logits below.)
    if (io.overflow_out != nullptr && (!(range_mx < 65504.0f) || headv[PK_WEIGHTS_UNFIT] != 0.0f)) *io.overflow_out = 1;
#ifdef PROFILE_STAMPS
#ifdef FUSED_STAMPS
    if (tid == 0 && io.prob_out)
        for (int i = 0; i < 32; ++i) io.prob_out[(size_t)row0 * DGL_MAX_TOKENS + i] += i < nst ? (float)(st_[i] - st_[0]) : 0.f;
    return;
#else
    if (tid == 0 && io.logits_out)
        for (int i = 0; i < 32; ++i) io.logits_out[(size_t)row0 * DGL_MAX_TOKENS + i] = i < nst ? (float)(st_[i] - st_[0]) : 0.f;
    return;
#endif
#endif

    // ---- Linear_final: logit[token] = bf + sum_c X[token][c] wf[c] ---------------------
    {
        float wf[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = vec4(PK_FINAL_W - PK_VEC0, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) wf[4 * g + e] = v[e];
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = fmaf(xres[rr][r], wf[r], acc);
            acc += __shfl_xor(acc, 32, 64);
            if (h == 0) part[(w * ROWS + rr) * 32 + q] = acc;
        }
    }
    __syncthreads();

    // ---- masked softmax, floor, greedy pick: wave w handles rows w, w+4, ... ------------
    for (int rr = w; rr < ROWS; rr += 4) {
        const int r = row0 + rr;
        if (r >= io.R) continue;
        float logit = P[PK_FINAL_B];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) logit += part[(ww * ROWS + rr) * 32 + q];
        if (io.overflow_out != nullptr && q < n && logit != logit) *io.overflow_out = 1;
        policy_row_output(io, r, logit);
    }
}


template <int ROWS, int GROUPS>
constexpr size_t policy16_lds_bytes() {
    return (size_t)GROUPS * ROWS * 6 * ARR * sizeof(_Float16) + (size_t)(GROUPS * 4 * ROWS * 32 + NHEAD) * sizeof(float);
}
