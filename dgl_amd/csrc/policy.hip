// Policy network of the DGL decode step on gfx950: encoder block + 3 decoder blocks,
// d = 128, 8 heads x 16, FF 512, no LayerNorm, fp32 end to end.
//
// Execution shape
//   * one workgroup = 4 waves handles ROWS rows (partial tours); a row is one 32-token
//     tile (TSP: 30 neighbours + current node; CVRP: 15 + 1 + 15 + 1), so a row maps
//     exactly onto the 32-wide dimension of v_mfma_f32_32x32x2_f32;
//   * wave w owns output features [32w, 32w+32) of every 128-wide projection, i.e. heads
//     2w and 2w+1: Q, K, V and the attention of those heads never leave its registers;
//   * activations live in LDS as X[token][feature] (row stride 132 floats so that a
//     16-byte read per lane is bank-conflict free); weights are streamed from L2 in a
//     pre-packed layout where one 16-byte load per lane feeds four MFMAs.
//
// Operand trick: for the MFMA D[32x32] += A[32x2] * B[2x32], lane l supplies A[l&31][l>>5]
// and B[l>>5][l&31]; D[(r&3)+8(r>>2)+4(l>>5)][l&31] is register r of lane l.  A lane
// (q, h) therefore always fetches X[token q][k_h] and W[out q][k_h]; passing (W, X) gives
// the transposed product Y^T (lane = token, registers = features), passing (X, W) gives
// Y (lane = feature, registers = tokens).  Q^T, K^T come out in exactly the layout the
// score MFMA wants for its two operands, V in the layout the P.V MFMA wants, and the
// softmax probabilities stay in the accumulator layout that the P.V MFMA consumes as its
// B operand: the attention needs no LDS traffic and no cross-lane shuffles beyond one
// exchange between the two half-waves.
#include <cstdlib>

#include "policy_layout.h"

#define LDX 132                       // LDS row stride in floats

extern "C" int64_t dgl_policy_packed_floats(int32_t) { return PK_TOTAL; }

// Fragment j (j = 0..nc-1) of out tile `ot`, K-chunks c0..c0+nc-1 of W[OUT][IN]:
//   dst[(j*stride)*256 + lane*4 + e] = W[32*ot + (lane&31)][8(c0+j) + 4(lane>>5) + e]
__global__ void pack_tile_kernel(const float *__restrict__ W, int IN, int ot, int c0, int nc, float *__restrict__ dst,
                                 int stride) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < nc * FRAG; o += gridDim.x * blockDim.x) {
        const int e = o & 3, lane = (o >> 2) & 63, j = o >> 8;
        dst[(size_t)j * stride * FRAG + lane * 4 + e] =
            W[(size_t)(32 * ot + (lane & 31)) * IN + 8 * (c0 + j) + 4 * (lane >> 5) + e];
    }
}
// f16 stream: fragment pair j (K-step s0 + j) of out tile `ot`:
//   dst[(2j*stride + part)*512 + lane*8 + e] = part(W[32*ot + (lane&31)][16(s0+j) + 8(lane>>5) + e]),
// part 0 = hi = binary16(W), part 1 = lo = binary16(W - hi)
// *unfit is set when a weight cannot be split (NaN, or beyond binary16's range)
__global__ void pack_tile16_kernel(const float *__restrict__ W, int IN, int ot, int s0, int ns, _Float16 *__restrict__ dst,
                                   int stride, float *__restrict__ unfit) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < ns * 512; o += gridDim.x * blockDim.x) {
        const int e = o & 7, lane = (o >> 3) & 63, j = o >> 9;
        const float w = W[(size_t)(32 * ot + (lane & 31)) * IN + 16 * (s0 + j) + 8 * (lane >> 5) + e];
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        if (!(fabsf(w) < 65504.0f)) *unfit = 1.0f;
        dst[(size_t)(2 * j * stride) * 512 + lane * 8 + e] = hi;
        dst[(size_t)(2 * j * stride + 1) * 512 + lane * 8 + e] = lo;
    }
}
__global__ void pack_copy_kernel(const float *__restrict__ src, int n, float *__restrict__ dst) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) dst[o] = src ? src[o] : 0.0f;
}
// encoder.embedding [128][F] -> [128][EMB_IN], zero padded
__global__ void pack_emb_kernel(const float *__restrict__ W, int F, float *__restrict__ dst) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < D_MODEL * EMB_IN; o += gridDim.x * blockDim.x) {
        const int c = o / EMB_IN, f = o % EMB_IN;
        dst[o] = f < F ? W[c * F + f] : 0.0f;
    }
}

extern "C" int dgl_policy_pack(const dgl_policy_weights_t *w, float *packed, dgl_stream_t stream) {
    DGL_CHECK_ARG(w != nullptr && packed != nullptr, "NULL argument");
    DGL_CHECK_ARG(w->problem == DGL_PROBLEM_TSP || w->problem == DGL_PROBLEM_CVRP, "unknown problem %d", w->problem);
    hipStream_t s = (hipStream_t)stream;
    const int F = w->problem == DGL_PROBLEM_TSP ? 7 : 9;
    DGL_CHECK_ARG(w->emb_w && w->emb_b && w->final_w && w->final_b && w->anchor_w[0] && w->anchor_b[0], "NULL weight");
    for (int b = 0; b < 4; ++b) {
        const dgl_block_weights_t &k = w->block[b];
        DGL_CHECK_ARG(k.wq && k.wk && k.wv && k.wc && k.bc && k.w1 && k.b1 && k.w2 && k.b2, "NULL weight in block %d", b);
    }
    if (w->problem == DGL_PROBLEM_CVRP) DGL_CHECK_ARG(w->anchor_w[1] && w->anchor_b[1], "NULL second anchor embedding");
    auto tile = [&](const float *W, int IN, int ot, int c0, int nc, size_t frag, int stride) {
        hipLaunchKernelGGL(pack_tile_kernel, dim3(4), dim3(256), 0, s, W, IN, ot, c0, nc,
                           packed + PK_STREAM0 + frag * FRAG, stride);
    };
    // f16 stream: `frag` = fragment index of the tile's first (hi) fragment, consecutive (hi, lo)
    // pairs `stride` fragments apart
    auto tile16 = [&](const float *W, int IN, int ot, int s0, int ns, size_t frag, int stride) {
        hipLaunchKernelGGL(pack_tile16_kernel, dim3(4), dim3(256), 0, s, W, IN, ot, s0, ns,
                           reinterpret_cast<_Float16 *>(packed + PK_STREAM16 + frag * FRAG), stride, packed + PK_WEIGHTS_UNFIT);
    };
    auto vec = [&](const float *v, int n, int off) {
        hipLaunchKernelGGL(pack_copy_kernel, dim3(1), dim3(256), 0, s, v, n, packed + off);
    };
    hipLaunchKernelGGL(pack_emb_kernel, dim3(8), dim3(256), 0, s, w->emb_w, F, packed + PK_EMB_W);
    vec(w->emb_b, D_MODEL, PK_EMB_B);
    for (int b = 0; b < 4; ++b) {
        const dgl_block_weights_t &k = w->block[b];
        vec(k.bc, D_MODEL, PK_VEC0 + b * PKV_SIZE + PKV_BC);
        vec(k.b1, FF_DIM, PK_VEC0 + b * PKV_SIZE + PKV_B1);
        vec(k.b2, D_MODEL, PK_VEC0 + b * PKV_SIZE + PKV_B2);
    }
    for (int a = 0; a < 2; ++a) vec(w->anchor_b[w->problem == DGL_PROBLEM_CVRP ? a : 0], D_MODEL, PK_ANCHOR_B + a * D_MODEL);
    vec(w->final_w, D_MODEL, PK_FINAL_W);
    vec(w->final_b, 1, PK_FINAL_B);
    vec(nullptr, 3, PK_FINAL_B + 1);              // PK_WEIGHTS_UNFIT and two spare words, before the tiles are packed
    for (int wv = 0; wv < 4; ++wv) {
        for (int b = 0; b < 4; ++b) {
            const dgl_block_weights_t &k = w->block[b];
            size_t f = (size_t)wv * WAVE_FRAGS + (size_t)b * BLK_FRAGS + (b > 0 ? ANCHOR_FRAGS : 0);
            tile(k.wq, D_MODEL, wv, 0, 16, f + 0, 3);
            tile(k.wk, D_MODEL, wv, 0, 16, f + 1, 3);
            tile(k.wv, D_MODEL, wv, 0, 16, f + 2, 3);
            tile(k.wc, D_MODEL, wv, 0, 16, f + 48, 1);
            // f16: per K-step q.hi q.lo k.hi k.lo v.hi v.lo
            tile16(k.wq, D_MODEL, wv, 0, 8, f + 0, 3);
            tile16(k.wk, D_MODEL, wv, 0, 8, f + 2, 3);
            tile16(k.wv, D_MODEL, wv, 0, 8, f + 4, 3);
            tile16(k.wc, D_MODEL, wv, 0, 8, f + 48, 1);
            for (int hc = 0; hc < 4; ++hc) {
                tile(k.w1, D_MODEL, 4 * hc + wv, 0, 16, f + 64 + 32 * hc, 1);
                tile(k.w2, FF_DIM, wv, 16 * hc, 16, f + 64 + 32 * hc + 16, 1);
                // f16 kernel's software pipeline: W1.0 | W1.1 W2.0 | W1.2 W2.1 | W1.3 W2.2 | W2.3
                tile16(k.w1, D_MODEL, 4 * hc + wv, 0, 8, f + (hc == 0 ? 64 : 48 + 32 * hc), 1);
                tile16(k.w2, FF_DIM, wv, 8 * hc, 8, f + (hc == 3 ? 176 : 96 + 32 * hc), 1);
            }
        }
        for (int a = 0; a < 2; ++a) {
            const float *wa = w->anchor_w[w->problem == DGL_PROBLEM_CVRP ? a : 0];
            tile(wa, D_MODEL, wv, 0, 16, (size_t)wv * WAVE_FRAGS + BLK_FRAGS + 16 * a, 1);
            tile16(wa, D_MODEL, wv, 0, 8, (size_t)wv * WAVE_FRAGS + BLK_FRAGS + 16 * a, 1);
        }
    }
    // zero tails the prefetch rings may read
    vec(nullptr, (PK_STREAM_FRAGS - 4 * WAVE_FRAGS) * FRAG, PK_STREAM0 + 4 * WAVE_FRAGS * FRAG);
    vec(nullptr, (PK_STREAM_FRAGS - 4 * WAVE_FRAGS) * FRAG, PK_STREAM16 + 4 * WAVE_FRAGS * FRAG);
    DGL_CHECK_LAUNCH("dgl_policy_pack");
    return DGL_OK;
}

// ---- device building blocks ----------------------------------------------------
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// acc[rr] += W(out tile) * X[rr]^T over 16 K-chunks (transposed product: lane = token,
// registers = features).  xp: LDS pointer X[row 0][token q][4h]; xrow = floats between rows.
template <int ROWS>
__device__ __forceinline__ void gemm_t(WeightRing<PF> &ring, const float *xp, int xrow, f32x16 (&acc)[ROWS]) {
    // X operands are read from LDS one K-chunk ahead of the MFMAs that consume them, so the
    // ~100+ cycle LDS latency hides behind eight 64-cycle MFMAs instead of stalling each chunk.
    f32x4 xn[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) xn[rr] = *reinterpret_cast<const f32x4 *>(xp + rr * xrow);
    Unroll<0, 16>::run([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const f32x4 w = ring.template take<c % PF>();
        f32x4 x[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) x[rr] = xn[rr];
        if constexpr (c + 1 < 16) {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) xn[rr] = *reinterpret_cast<const f32x4 *>(xp + rr * xrow + 8 * (c + 1));
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the operand reads ahead of this chunk's MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) acc[rr] = MFMA(w[e], x[rr][e], acc[rr]);
    });
}

template <int ROWS>
__global__ __launch_bounds__(256) void policy_kernel_f32(const float *__restrict__ P, dgl_policy_io_t io) {
    extern __shared__ float lds[];
    float *XA = lds;                           // [ROWS][32][LDX]
    float *XB = XA + ROWS * 32 * LDX;          // [ROWS][32][LDX]
    float *part = XB + ROWS * 32 * LDX;        // [4][ROWS][32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * ROWS;
    const int n = io.n;
    constexpr int XROW = 32 * LDX;

    WeightRing<PF> ring;
    ring.fill(reinterpret_cast<const f32x4 *>(P + PK_STREAM0 + (size_t)w * WAVE_FRAGS * FRAG) + lane);

    // ---- embedding: X[token][c] = b[c] + sum_f tile[token][f] * We[c][f] ----------
    {
        const int c = tid & 127;
        float we[EMB_IN];
#pragma unroll
        for (int f = 0; f < EMB_IN; ++f) we[f] = P[PK_EMB_W + c * EMB_IN + f];
        const float be = P[PK_EMB_B + c];
        for (int rr = 0; rr < ROWS; ++rr) {
            const int r = row0 + rr;
            const float *trow = io.tile + (size_t)r * io.tile_row_stride;
            for (int tk = tid >> 7; tk < 32; tk += 2) {
                float v = 0.0f;
                if (r < io.R && tk < n) {
                    const f32x4 *tp = reinterpret_cast<const f32x4 *>(trow + tk * DGL_TILE_STRIDE);
                    f32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    float acc = 0.0f;
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc = fmaf(t0[f], we[f], acc);
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc = fmaf(t1[f], we[4 + f], acc);
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc = fmaf(t2[f], we[8 + f], acc);
                    v = acc + be;
                }
                XA[rr * XROW + tk * LDX + c] = v;
            }
        }
    }
    __syncthreads();

    const float *xa_lane = XA + q * LDX + 4 * h;      // operand fetch base (row 0)
    const float *xb_lane = XB + q * LDX + 4 * h;
    const int anchor0 = io.k_last, anchor1 = n - 1;    // TSP: k_last == n-1

    for (int layer = 0; layer < 4; ++layer) {
        const float *PV = P + PK_VEC0 + layer * PKV_SIZE;

        if (layer == 1) {
            // decoder entry: re-embed the anchor token(s) (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-446)
            // (the stream always carries two embeddings; TSP's second equals its first and is discarded)
            for (int a = 0; a < 2; ++a) {
                f32x16 acc[ROWS];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) acc[rr] = zero16();
                gemm_t<ROWS>(ring, xa_lane, XROW, acc);
                __syncthreads();                       // every wave has finished reading XA
                const int tok = (a == 0) ? anchor0 : anchor1;
                if (q == tok && (a == 0 || io.problem == DGL_PROBLEM_CVRP)) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = 32 * w + 8 * g + 4 * h;
                        const f32x4 bb = *reinterpret_cast<const f32x4 *>(P + PK_ANCHOR_B + a * D_MODEL + f);
#pragma unroll
                        for (int rr = 0; rr < ROWS; ++rr) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[rr][4 * g + e] + bb[e];
                            *reinterpret_cast<f32x4 *>(XA + rr * XROW + tok * LDX + f) = o;
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- Q^T, K^T (transposed form) and V (normal form) for heads 2w, 2w+1 ----
        f32x16 aq[ROWS], ak[ROWS], av[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) { aq[rr] = zero16(); ak[rr] = zero16(); av[rr] = zero16(); }
        {
            f32x4 xn[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) xn[rr] = *reinterpret_cast<const f32x4 *>(xa_lane + rr * XROW);
            Unroll<0, 16>::run([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const f32x4 fq = ring.template take<(3 * c) % PF>();
                const f32x4 fk = ring.template take<(3 * c + 1) % PF>();
                const f32x4 fv = ring.template take<(3 * c + 2) % PF>();
                f32x4 x[ROWS];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) x[rr] = xn[rr];
                if constexpr (c + 1 < 16) {
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr)
                        xn[rr] = *reinterpret_cast<const f32x4 *>(xa_lane + rr * XROW + 8 * (c + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        aq[rr] = MFMA(fq[e], x[rr][e], aq[rr]);
                        ak[rr] = MFMA(fk[e], x[rr][e], ak[rr]);
                        av[rr] = MFMA(x[rr][e], fv[e], av[rr]);
                    }
                }
            });
        }

        // ---- attention, both local heads, straight from the accumulators -----------
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            f32x16 o_sel = zero16();
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                // S^T[key j][query i] = sum_d K[j][d] Q[i][d]
                f32x16 s = zero16();
#pragma unroll
                for (int m = 0; m < 8; ++m) s = MFMA(ak[rr][8 * hd + m], aq[rr][8 * hd + m], s);
                // lane (i, h) holds keys j = (m&3) + 8(m>>2) + 4h
                // softmax over keys; 1/4 = 1/sqrt(16) is folded into the exponent:
                //   exp((s - max)/4) = 2^((s - max) * log2(e)/4), one v_exp_f32 per key
                float mx = -INFINITY;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int j = (m & 3) + 8 * (m >> 2) + 4 * h;
                    s[m] = (j < n) ? s[m] : -INFINITY;
                    mx = fmaxf(mx, s[m]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.0f;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    s[m] = __builtin_amdgcn_exp2f((s[m] - mx) * 0.36067376022224085f);
                    sum += s[m];
                }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int m = 0; m < 16; ++m) s[m] = s[m] * inv;
                // O^T[d][i] = sum_j V[j][d] P[i][j]; rows d of this head are registers 8hd..8hd+7
                f32x16 o = zero16();
#pragma unroll
                for (int m = 0; m < 16; ++m) o = MFMA(av[rr][m], s[m], o);
#pragma unroll
                for (int m = 0; m < 8; ++m) o_sel[8 * hd + m] = o[8 * hd + m];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o_sel[4 * g + e];
                *reinterpret_cast<f32x4 *>(XB + rr * XROW + q * LDX + 32 * w + 8 * g + 4 * h) = v;
            }
        }
        __syncthreads();

        // ---- multi_head_combine + residual -> o1 (in place in XA, own columns) -------
        {
            f32x16 acc[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) acc[rr] = zero16();
            gemm_t<ROWS>(ring, xb_lane, XROW, acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * w + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(PV + PKV_BC + f);
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    float *xp = XA + rr * XROW + q * LDX + f;
                    f32x4 x = *reinterpret_cast<const f32x4 *>(xp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = x[e] + (acc[rr][4 * g + e] + bb[e]);
                    *reinterpret_cast<f32x4 *>(xp) = x;
                }
            }
        }
        __syncthreads();

        // ---- feed forward: 4 hidden chunks of 128 -----------------------------------
        {
            f32x16 af[ROWS];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) af[rr] = zero16();
            for (int hc = 0; hc < 4; ++hc) {
                f32x16 ah[ROWS];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) ah[rr] = zero16();
                gemm_t<ROWS>(ring, xa_lane, XROW, ah);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = 32 * w + 8 * g + 4 * h;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(PV + PKV_B1 + 128 * hc + f);
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(ah[rr][4 * g + e] + bb[e], 0.0f);
                        *reinterpret_cast<f32x4 *>(XB + rr * XROW + q * LDX + f) = v;
                    }
                }
                __syncthreads();
                gemm_t<ROWS>(ring, xb_lane, XROW, af);
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * w + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(PV + PKV_B2 + f);
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    float *xp = XA + rr * XROW + q * LDX + f;
                    f32x4 x = *reinterpret_cast<const f32x4 *>(xp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = x[e] + (af[rr][4 * g + e] + bb[e]);
                    *reinterpret_cast<f32x4 *>(xp) = x;
                }
            }
        }
        __syncthreads();
    }

    // ---- Linear_final: logit[token] = bf + sum_c X[token][c] wf[c] ---------------------
    {
        const int f0 = 32 * w + 16 * h;
        f32x4 wf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wf[g] = *reinterpret_cast<const f32x4 *>(P + PK_FINAL_W + f0 + 4 * g);
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            float acc = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(XA + rr * XROW + q * LDX + f0 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(x[e], wf[g][e], acc);
            }
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
        policy_row_output(io, r, logit);
    }
}

// dense[r][a] = 1e-20 except the candidate actions of row r
__global__ void scatter_probs_kernel(dgl_policy_io_t io, const float *__restrict__ prob, float *__restrict__ dense, int A) {
    const int r = blockIdx.x;
    float *drow = dense + (size_t)r * A;
    for (int a = threadIdx.x; a < A; a += blockDim.x) drow[a] = 1e-20f;
    __syncthreads();
    const int n = io.n;
    for (int tok = threadIdx.x; tok < n; tok += blockDim.x) {
        if (tok == io.k_last || tok == n - 1) continue;
        int act;
        if (io.problem == DGL_PROBLEM_TSP || tok < io.k_last)
            act = io.idx_last[(size_t)r * io.idx_stride + tok];
        else
            act = io.idx_depot[(size_t)r * io.idx_dep_stride + (tok - io.k_last - 1)] + io.N + 1;
        drow[act] = prob[(size_t)r * (io.out_stride > 0 ? io.out_stride : DGL_MAX_TOKENS) + tok];
    }
}

static int check_io(const dgl_policy_io_t *io, int max_n = DGL_MAX_TOKENS) {
    DGL_CHECK_ARG(io != nullptr, "io is NULL");
    DGL_CHECK_ARG(io->problem == DGL_PROBLEM_TSP || io->problem == DGL_PROBLEM_CVRP, "unknown problem %d", io->problem);
    DGL_CHECK_ARG(io->R > 0, "R=%d", io->R);
    DGL_CHECK_ARG(io->n >= 2 && io->n <= max_n, "n=%d tokens per row outside [2,%d]", io->n, max_n);
    if (io->problem == DGL_PROBLEM_TSP) {
        DGL_CHECK_ARG(io->k_last == io->n - 1, "TSP needs k_last == n-1 (k_last=%d n=%d)", io->k_last, io->n);
    } else {
        DGL_CHECK_ARG(io->k_last >= 1 && io->k_last <= io->n - 3, "CVRP k_last=%d invalid for n=%d", io->k_last, io->n);
        DGL_CHECK_ARG(io->idx_depot != nullptr && io->idx_dep_stride >= io->n - io->k_last - 2, "CVRP needs idx_depot");
    }
    DGL_CHECK_ARG(io->idx_last != nullptr && io->idx_stride >= io->k_last, "idx_last missing or stride %d < k_last %d",
                  io->idx_stride, io->k_last);
    return DGL_OK;
}

template <int ROWS>
static int launch_policy(const float *P, const dgl_policy_io_t *io, hipStream_t s) {
    const size_t lds = (size_t)(2 * ROWS * 32 * LDX + 4 * ROWS * 32) * sizeof(float);
    // (set on every launch: the attribute belongs to the current device's copy of the kernel)
    if (lds > 48 * 1024)
        dgl_max_dynamic_lds((const void *)policy_kernel_f32<ROWS>, (int)lds);
    hipLaunchKernelGGL(policy_kernel_f32<ROWS>, dim3((io->R + ROWS - 1) / ROWS), dim3(256), lds, s, P, *io);
    DGL_CHECK_LAUNCH("dgl_policy_forward");
    return DGL_OK;
}

extern "C" int dgl_policy_forward(const float *packed_weights, const dgl_policy_io_t *io, dgl_stream_t stream) {
    DGL_CHECK_ARG(packed_weights != nullptr, "packed_weights is NULL");
    const bool wide = io != nullptr && io->n > DGL_MAX_TOKENS;
    if (int e = check_io(io, wide && io->arithmetic == DGL_ARITH_SPLIT_F16 ? DGL_MAX_WIDE_TOKENS : DGL_MAX_TOKENS)) return e;
    DGL_CHECK_ARG(io->tile != nullptr, "tile is NULL");
    DGL_CHECK_ARG(io->tile_row_stride >= io->n * DGL_TILE_STRIDE, "tile_row_stride %d < n*%d", io->tile_row_stride,
                  DGL_TILE_STRIDE);
    if (wide) {
        // 33..128 tokens per row, split-f16 arithmetic: one workgroup per row (policy16_wide.h)
        DGL_CHECK_ARG(io->out_stride >= io->n, "out_stride %d < n=%d tokens per row", io->out_stride, io->n);
        return dgl_launch_policy16_wide(packed_weights, io, (hipStream_t)stream);
    }
    DGL_CHECK_ARG(io->out_stride == 0 || io->out_stride == DGL_MAX_TOKENS, "the 32-token kernels write %d floats per row",
                  DGL_MAX_TOKENS);
    // two rows per workgroup share every weight fragment; a single row per workgroup
    // keeps all 256 CUs busy when there are few rows
    if (io->arithmetic == DGL_ARITH_SPLIT_F16) return dgl_launch_policy16(packed_weights, io, (hipStream_t)stream);
    DGL_CHECK_ARG(io->arithmetic == DGL_ARITH_F32, "unknown arithmetic %d", io->arithmetic);
    static const int forced = getenv("DGL_POLICY_ROWS") ? atoi(getenv("DGL_POLICY_ROWS")) : 0;
    const int rows = forced ? forced : (io->R >= 512 ? 2 : 1);
    if (rows == 2) return launch_policy<2>(packed_weights, io, (hipStream_t)stream);
    return launch_policy<1>(packed_weights, io, (hipStream_t)stream);
}

extern "C" int dgl_policy_scatter_probs(const dgl_policy_io_t *io, const float *prob, float *dense_out, int32_t A,
                                        dgl_stream_t stream) {
    if (int e = check_io(io, io && io->out_stride > 0 ? io->out_stride : DGL_MAX_TOKENS)) return e;
    DGL_CHECK_ARG(prob != nullptr && dense_out != nullptr, "NULL argument");
    const int need = io->problem == DGL_PROBLEM_TSP ? io->N : 2 * (io->N + 1);
    DGL_CHECK_ARG(A == need, "A=%d but the action space is %d", A, need);
    hipLaunchKernelGGL(scatter_probs_kernel, dim3(io->R), dim3(256), 0, (hipStream_t)stream, *io, prob, dense_out, A);
    DGL_CHECK_LAUNCH("dgl_policy_scatter_probs");
    return DGL_OK;
}
