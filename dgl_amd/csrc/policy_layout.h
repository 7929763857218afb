// Packed weight layout shared by the two policy kernels (policy.hip: f32-input MFMA;
// policy16.hip: f16 split-operand MFMA) and by dgl_policy_pack.
//
// Small vectors first, then one weight STREAM per wave and per kernel: the 1 KB fragments
// (64 lanes x 16 B) a wave consumes, in exactly the order it consumes them, so that a wave's
// weight traffic is one linear read which a register ring prefetches ahead across GEMM boundaries.
//   per block : 48 fragments q,k,v interleaved by K-step; 16 combine; 4 x (16 W1 + 16 W2)
//               (f16 stream: W1.0 | W1.1 W2.0 | W1.2 W2.1 | W1.3 W2.2 | W2.3, its pipeline's order)
//   stream    : block 0 | anchor embedding(s) 2 x 16 (TSP repeats its single one) | blocks 1..3
// f32 stream : a fragment holds W[32 ot + (lane&31)][8 c + 4 (lane>>5) + e], e = 0..3, as binary32:
//              four K=2 MFMA steps.
// f16 stream : fragments come in (hi, lo) pairs; a pair holds W[32 ot + (lane&31)][16 s + 8 (lane>>5) + e],
//              e = 0..7, as hi = binary16(W), lo = binary16(W - hi): one K=16 MFMA step each.
//              Same fragment count and bytes as the f32 stream.
#pragma once
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define D_MODEL 128
#define FF_DIM 512
#define EMB_IN DGL_TILE_STRIDE        // padded input features

#define PK_EMB_W 0
#define PK_EMB_B (PK_EMB_W + D_MODEL * EMB_IN)
#define PK_VEC0 (PK_EMB_B + D_MODEL)               // per block: bc[128] b1[512] b2[128]
#define PKV_BC 0
#define PKV_B1 (PKV_BC + D_MODEL)
#define PKV_B2 (PKV_B1 + FF_DIM)
#define PKV_SIZE (PKV_B2 + D_MODEL)
#define PK_ANCHOR_B (PK_VEC0 + 4 * PKV_SIZE)       // [2][128]
#define PK_FINAL_W (PK_ANCHOR_B + 2 * D_MODEL)
#define PK_FINAL_B (PK_FINAL_W + D_MODEL)
#define PK_WEIGHTS_UNFIT (PK_FINAL_B + 1)            // != 0: a matrix entry is NaN or beyond binary16's range (set by dgl_policy_pack)
#define PK_STREAM0 (PK_FINAL_B + 4)
#define FRAG 256                                    // floats (= 1 KB) per fragment
#define BLK_FRAGS 192                               // fragments per wave per block
#define ANCHOR_FRAGS 32
#define WAVE_FRAGS (4 * BLK_FRAGS + ANCHOR_FRAGS)   // 800
#define PF 8                                        // f32 kernel: fragments in flight per wave
#ifndef PF16
#define PF16 16                                     // f16 kernel: ring depth (every GEMM starts at a multiple of 16 of the stream; blocks and the anchor section are multiples of 32 long, so 32 works too)
#endif
#define PK_STREAM_FRAGS (4 * WAVE_FRAGS + 32)       // + zero tail the prefetch rings may read
#define PK_STREAM16 (PK_STREAM0 + PK_STREAM_FRAGS * FRAG)
#define PK_TOTAL (PK_STREAM16 + PK_STREAM_FRAGS * FRAG)

template <int C, int N>
struct Unroll {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, C>{});
        Unroll<C + 1, N>::run(f);
    }
};
template <int N>
struct Unroll<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// Register ring holding the next DEPTH weight fragments of this wave's stream.  take<SLOT>() returns
// the oldest fragment and immediately re-issues the slot's load DEPTH fragments ahead, so weight
// loads stay in flight across MFMAs, epilogues and barriers.
template <int DEPTH>
struct WeightRing {
    f32x4 r[DEPTH];
    const f32x4 *next;      // this lane's 16 bytes of the next fragment to request
    __device__ __forceinline__ void fill(const f32x4 *p) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) r[i] = p[i * 64];
        next = p + DEPTH * 64;
    }
    template <int SLOT>
    __device__ __forceinline__ f32x4 take() {
        const f32x4 v = r[SLOT];
        r[SLOT] = *next;
        next += 64;
        return v;
    }
};

// masked softmax over the candidate tokens, probability floor, greedy pick and outputs for one
// row; called by a whole wave (both half-waves compute the same values).  logit = this lane's
// token (lane & 31) logit.
__device__ __forceinline__ void policy_row_output(const dgl_policy_io_t &io, int r, float logit) {
    const int lane = threadIdx.x & 63, tok = lane & 31, h = lane >> 5;
    const int n = io.n, anchor0 = io.k_last, anchor1 = n - 1;      // TSP: k_last == n-1
    const bool cand = tok < n && tok != anchor0 && tok != anchor1;
    bool open = cand;
    if (io.problem == DGL_PROBLEM_CVRP && io.infeasible != nullptr && tok < io.k_last)
        open = cand && (io.infeasible[(size_t)r * io.idx_stride + tok] == 0);
    const float ml = open ? logit : -INFINITY;
    const float mx = wave_max_f32(ml);
    const float ex = open ? expf(ml - mx) : 0.0f;
    float sum = ex;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);   // within a 32-lane half
    float p = ex / sum;
    if (p <= 1e-5f) p = p + 1e-7f;                   // TSP/TSPModel.py:323-325
    if (!cand) p = 0.0f;
    if (h == 0) {
        if (io.logits_out) io.logits_out[(size_t)r * DGL_MAX_TOKENS + tok] = tok < n ? logit : 0.0f;
        if (io.prob_out) io.prob_out[(size_t)r * DGL_MAX_TOKENS + tok] = p;
    }
    if (io.selected_out) {
        const float best = wave_max_f32(cand ? p : -1.0f);
        const unsigned long long m = __ballot(cand && p == best) & 0xFFFFFFFFull;
        const int btok = __ffsll((long long)m) - 1;
        if (lane == 0) {
            int act;
            if (io.problem == DGL_PROBLEM_TSP || btok < io.k_last)
                act = io.idx_last[(size_t)r * io.idx_stride + btok];
            else
                act = io.idx_depot[(size_t)r * io.idx_dep_stride + (btok - io.k_last - 1)] + io.N + 1;
            io.selected_out[r] = act;
        }
    }
}

// defined in policy16.hip
int dgl_launch_policy16(const float *packed, const dgl_policy_io_t *io, hipStream_t stream);
int dgl_launch_policy16_wide(const float *packed, const dgl_policy_io_t *io, hipStream_t stream);   // 33..DGL_MAX_WIDE_TOKENS tokens
