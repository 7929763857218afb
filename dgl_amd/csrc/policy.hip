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
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define D_MODEL 128
#define FF_DIM 512
#define LDX 132                       // LDS row stride in floats
#define EMB_IN DGL_TILE_STRIDE        // padded input features

// ---- packed weight layout (float offsets) ---------------------------------------
#define PK_LIN128 (D_MODEL * D_MODEL)               // 16384: one packed 128x128 matrix
#define PK_EMB_W 0
#define PK_EMB_B (PK_EMB_W + D_MODEL * EMB_IN)
#define PK_BLOCK0 (PK_EMB_B + D_MODEL)
#define PKB_WQ 0
#define PKB_WK (PKB_WQ + PK_LIN128)
#define PKB_WV (PKB_WK + PK_LIN128)
#define PKB_WC (PKB_WV + PK_LIN128)
#define PKB_BC (PKB_WC + PK_LIN128)
#define PKB_W1 (PKB_BC + D_MODEL)
#define PKB_B1 (PKB_W1 + FF_DIM * D_MODEL)
#define PKB_W2 (PKB_B1 + FF_DIM)
#define PKB_B2 (PKB_W2 + FF_DIM * D_MODEL)
#define PK_BLOCK_SIZE (PKB_B2 + D_MODEL)
#define PK_ANCHOR0 (PK_BLOCK0 + 4 * PK_BLOCK_SIZE)
#define PK_ANCHOR_SIZE (PK_LIN128 + D_MODEL)
#define PK_FINAL_W (PK_ANCHOR0 + 2 * PK_ANCHOR_SIZE)
#define PK_FINAL_B (PK_FINAL_W + D_MODEL)
#define PK_TOTAL (PK_FINAL_B + 4)

extern "C" int64_t dgl_policy_packed_floats(int32_t) { return PK_TOTAL; }

// dst[((ot*C + c)*64 + lane)*4 + e] = W[32*ot + (lane&31)][8c + 4(lane>>5) + e],  C = IN/8
__global__ void pack_linear_kernel(const float *__restrict__ W, int OUT, int IN, float *__restrict__ dst) {
    const int total = OUT * IN;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int e = o & 3, lane = (o >> 2) & 63, blk = o >> 8;
        const int C = IN / 8, c = blk % C, ot = blk / C;
        dst[o] = W[(size_t)(32 * ot + (lane & 31)) * IN + 8 * c + 4 * (lane >> 5) + e];
    }
}
__global__ void pack_copy_kernel(const float *__restrict__ src, int n, float *__restrict__ dst) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) dst[o] = src[o];
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
    auto lin = [&](const float *W, int OUT, int IN, int off) {
        hipLaunchKernelGGL(pack_linear_kernel, dim3(64), dim3(256), 0, s, W, OUT, IN, packed + off);
    };
    auto vec = [&](const float *v, int n, int off) {
        hipLaunchKernelGGL(pack_copy_kernel, dim3(1), dim3(256), 0, s, v, n, packed + off);
    };
    hipLaunchKernelGGL(pack_emb_kernel, dim3(8), dim3(256), 0, s, w->emb_w, F, packed + PK_EMB_W);
    vec(w->emb_b, D_MODEL, PK_EMB_B);
    for (int b = 0; b < 4; ++b) {
        const dgl_block_weights_t &k = w->block[b];
        DGL_CHECK_ARG(k.wq && k.wk && k.wv && k.wc && k.bc && k.w1 && k.b1 && k.w2 && k.b2, "NULL weight in block %d", b);
        const int o = PK_BLOCK0 + b * PK_BLOCK_SIZE;
        lin(k.wq, D_MODEL, D_MODEL, o + PKB_WQ);
        lin(k.wk, D_MODEL, D_MODEL, o + PKB_WK);
        lin(k.wv, D_MODEL, D_MODEL, o + PKB_WV);
        lin(k.wc, D_MODEL, D_MODEL, o + PKB_WC);
        vec(k.bc, D_MODEL, o + PKB_BC);
        lin(k.w1, FF_DIM, D_MODEL, o + PKB_W1);
        vec(k.b1, FF_DIM, o + PKB_B1);
        lin(k.w2, D_MODEL, FF_DIM, o + PKB_W2);
        vec(k.b2, D_MODEL, o + PKB_B2);
    }
    for (int a = 0; a < 2; ++a) {
        const int src = (w->problem == DGL_PROBLEM_CVRP) ? a : 0;
        DGL_CHECK_ARG(w->anchor_w[src] && w->anchor_b[src], "NULL anchor embedding %d", src);
        lin(w->anchor_w[src], D_MODEL, D_MODEL, PK_ANCHOR0 + a * PK_ANCHOR_SIZE);
        vec(w->anchor_b[src], D_MODEL, PK_ANCHOR0 + a * PK_ANCHOR_SIZE + PK_LIN128);
    }
    vec(w->final_w, D_MODEL, PK_FINAL_W);
    vec(w->final_b, 1, PK_FINAL_B);
    DGL_CHECK_LAUNCH("dgl_policy_pack");
    return DGL_OK;
}

// ---- device building blocks ----------------------------------------------------
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// acc[rr] += W(out tile) * X[rr]^T   (transposed product: lane = token, registers = features)
//   wp: packed weights of this out tile, already offset by lane (float4 units), chunk stride 64
//   xp: LDS pointer X[row 0][token q][4h]; xrow = floats between rows
template <int ROWS, int NCHUNK>
__device__ __forceinline__ void gemm_t(const f32x4 *__restrict__ wp, const float *xp, int xrow, f32x16 (&acc)[ROWS]) {
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const f32x4 w = wp[c * 64];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(xp + rr * xrow + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rr] = MFMA(w[e], x[e], acc[rr]);
        }
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

template <int ROWS>
__global__ __launch_bounds__(256) void policy_kernel(const float *__restrict__ P, dgl_policy_io_t io) {
    extern __shared__ float lds[];
    float *XA = lds;                           // [ROWS][32][LDX]
    float *XB = XA + ROWS * 32 * LDX;          // [ROWS][32][LDX]
    float *part = XB + ROWS * 32 * LDX;        // [4][ROWS][32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * ROWS;
    const int n = io.n;
    constexpr int XROW = 32 * LDX;

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
        const float *PB = P + PK_BLOCK0 + layer * PK_BLOCK_SIZE;

        if (layer == 1) {
            // decoder entry: re-embed the anchor token(s) (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-446)
            const int na = (io.problem == DGL_PROBLEM_CVRP) ? 2 : 1;
            for (int a = 0; a < na; ++a) {
                const float *PA = P + PK_ANCHOR0 + a * PK_ANCHOR_SIZE;
                f32x16 acc[ROWS];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) acc[rr] = zero16();
                gemm_t<ROWS, 16>(reinterpret_cast<const f32x4 *>(PA + w * 16 * 256) + lane, xa_lane, XROW, acc);
                __syncthreads();                       // every wave has finished reading XA
                const int tok = (a == 0) ? anchor0 : anchor1;
                if (q == tok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = 32 * w + 8 * g + 4 * h;
                        const f32x4 bb = *reinterpret_cast<const f32x4 *>(PA + PK_LIN128 + f);
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
            const f32x4 *wq = reinterpret_cast<const f32x4 *>(PB + PKB_WQ + w * 16 * 256) + lane;
            const f32x4 *wk = reinterpret_cast<const f32x4 *>(PB + PKB_WK + w * 16 * 256) + lane;
            const f32x4 *wv = reinterpret_cast<const f32x4 *>(PB + PKB_WV + w * 16 * 256) + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 fq = wq[c * 64], fk = wk[c * 64], fv = wv[c * 64];
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(xa_lane + rr * XROW + 8 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        aq[rr] = MFMA(fq[e], x[e], aq[rr]);
                        ak[rr] = MFMA(fk[e], x[e], ak[rr]);
                        av[rr] = MFMA(x[e], fv[e], av[rr]);
                    }
                }
            }
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
                float mx = -INFINITY;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int j = (m & 3) + 8 * (m >> 2) + 4 * h;
                    s[m] = (j < n) ? s[m] * 0.25f : -INFINITY;
                    mx = fmaxf(mx, s[m]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.0f;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    s[m] = expf(s[m] - mx);
                    sum += s[m];
                }
                sum += __shfl_xor(sum, 32, 64);
#pragma unroll
                for (int m = 0; m < 16; ++m) s[m] = s[m] / sum;
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
            gemm_t<ROWS, 16>(reinterpret_cast<const f32x4 *>(PB + PKB_WC + w * 16 * 256) + lane, xb_lane, XROW, acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * w + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(PB + PKB_BC + f);
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
                gemm_t<ROWS, 16>(reinterpret_cast<const f32x4 *>(PB + PKB_W1 + (4 * hc + w) * 16 * 256) + lane, xa_lane,
                                 XROW, ah);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = 32 * w + 8 * g + 4 * h;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(PB + PKB_B1 + 128 * hc + f);
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(ah[rr][4 * g + e] + bb[e], 0.0f);
                        *reinterpret_cast<f32x4 *>(XB + rr * XROW + q * LDX + f) = v;
                    }
                }
                __syncthreads();
                gemm_t<ROWS, 16>(reinterpret_cast<const f32x4 *>(PB + PKB_W2 + (w * 64 + 16 * hc) * 256) + lane, xb_lane,
                                 XROW, af);
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * w + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(PB + PKB_B2 + f);
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
        const int tok = q;                               // both half-waves compute the same values
        float logit = P[PK_FINAL_B];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) logit += part[(ww * ROWS + rr) * 32 + tok];
        bool cand = tok < n && tok != anchor0 && tok != anchor1;
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
        drow[act] = prob[(size_t)r * DGL_MAX_TOKENS + tok];
    }
}

static int check_io(const dgl_policy_io_t *io) {
    DGL_CHECK_ARG(io != nullptr, "io is NULL");
    DGL_CHECK_ARG(io->problem == DGL_PROBLEM_TSP || io->problem == DGL_PROBLEM_CVRP, "unknown problem %d", io->problem);
    DGL_CHECK_ARG(io->R > 0, "R=%d", io->R);
    DGL_CHECK_ARG(io->n >= 2 && io->n <= DGL_MAX_TOKENS, "n=%d tokens per row outside [2,%d]", io->n, DGL_MAX_TOKENS);
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
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)policy_kernel<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(policy_kernel<ROWS>, dim3((io->R + ROWS - 1) / ROWS), dim3(256), lds, s, P, *io);
    DGL_CHECK_LAUNCH("dgl_policy_forward");
    return DGL_OK;
}

extern "C" int dgl_policy_forward(const float *packed_weights, const dgl_policy_io_t *io, dgl_stream_t stream) {
    DGL_CHECK_ARG(packed_weights != nullptr, "packed_weights is NULL");
    if (int e = check_io(io)) return e;
    DGL_CHECK_ARG(io->tile != nullptr, "tile is NULL");
    DGL_CHECK_ARG(io->tile_row_stride >= io->n * DGL_TILE_STRIDE, "tile_row_stride %d < n*%d", io->tile_row_stride,
                  DGL_TILE_STRIDE);
    // two rows per workgroup share every weight fragment; a single row per workgroup
    // keeps all 256 CUs busy when there are few rows
    if (io->R >= 512) return launch_policy<2>(packed_weights, io, (hipStream_t)stream);
    return launch_policy<1>(packed_weights, io, (hipStream_t)stream);
}

extern "C" int dgl_policy_scatter_probs(const dgl_policy_io_t *io, const float *prob, float *dense_out, int32_t A,
                                        dgl_stream_t stream) {
    if (int e = check_io(io)) return e;
    DGL_CHECK_ARG(prob != nullptr && dense_out != nullptr, "NULL argument");
    const int need = io->problem == DGL_PROBLEM_TSP ? io->N : 2 * (io->N + 1);
    DGL_CHECK_ARG(A == need, "A=%d but the action space is %d", A, need);
    hipLaunchKernelGGL(scatter_probs_kernel, dim3(io->R), dim3(256), 0, (hipStream_t)stream, *io, prob, dense_out, A);
    DGL_CHECK_LAUNCH("dgl_policy_scatter_probs");
    return DGL_OK;
}
