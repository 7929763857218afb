// General (any number of tokens per row) forward of the policy network and its backward pass:
// the teacher-forced training step of CVRPTrainer._train_one_batch (CVRP/CVRPTrainer.py:232-282,
// model train branch CVRP/CVRPModel.py:311-345; TSP/TSPModel.py:206-217), and the same forward for
// inference when a row has more than one 32-token tile (knn + depot_knn + 2 > 32).
//
// Unlike the fused decode kernels this path is a sequence of plain kernels over the flattened
// token axis (M = R*n tokens): one strided f32-MFMA GEMM kernel serves the three products of a
// linear layer (Y = X W^T, dX = dY W, dW += dY^T X), attention runs per (row, head) with the
// probabilities kept for the backward pass, the output stage computes the teacher's probability
// and the gradient of  -mean(log p_teacher)  with respect to the logits.  Everything is binary32
// (v_mfma_f32_32x32x2_f32 = FMA chains); weight gradients are accumulated with float atomics.
#include "policy_layout.h"

#include <mutex>

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define HEADS 8
#define HDIM 16

// ------------------------------------------------------------------------------- GEMM
// C[i][j] (op)= sum_kk A(i,kk) B(kk,j),  A(i,kk) = A[i*sai + kk*sak],  B(kk,j) = B[kk*sbk + j*sbn],
// i < I, j < J, kk in the z-th chunk of [0, KK).  64x64 output tile per workgroup (4 waves, each a
// 32x32 MFMA tile), K staged through LDS 32 at a time.
//   mode 0: C = acc (+ bias[j]) (ReLU) (+ add[i][j]);  mode 1: C += acc;  mode 2: atomicAdd(C, acc)
struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias, *add;
    int I, J, KK;
    long sai, sak, sbk, sbn;
    int ldc, ldadd;
    int mode, relu, kchunk;
};

#define GBK 32      // K-extent of one LDS stage

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

// Several independent products share one launch: workgroup `blockIdx.x` belongs to the product whose block
// range [first[b], first[b+1]) contains it.
#define GEMM_BATCH_MAX 6
struct GemmBatch {
    GemmArgs g[GEMM_BATCH_MAX];
    int first[GEMM_BATCH_MAX + 1];
    int count;
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch batch) {
    __shared__ float As[64][GBK + 1], Bs[64][GBK + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 31, h = lane >> 5;
    int b = 0;
    while (b + 1 < batch.count && (int)blockIdx.x >= batch.first[b + 1]) ++b;
    const GemmArgs g = batch.g[b];
    const int local = blockIdx.x - batch.first[b];
    const int gx = (g.I + 63) / 64, gy = (g.J + 63) / 64;
    const int i0 = (local % gx) * 64, j0 = ((local / gx) % gy) * 64;
    const int k_lo = (local / (gx * gy)) * g.kchunk, k_hi = min(g.KK, k_lo + g.kchunk);
    const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
    f32x16 acc = zero16();
    // staging map: each thread moves two groups of four elements per operand and stage; the four run along
    // whichever index is contiguous in memory.  The next stage is fetched into registers while the MFMAs of the
    // current one run.
    const bool a_k_fast = g.sak == 1, b_k_fast = g.sbk == 1;
    const bool a_vec = ((reinterpret_cast<size_t>(g.A) & 15) == 0) && (((a_k_fast ? g.sai : g.sak) & 3) == 0);
    const bool b_vec = ((reinterpret_cast<size_t>(g.B) & 15) == 0) && (((b_k_fast ? g.sbn : g.sbk) & 3) == 0);
    int am[2], ak[2], bm[2], bk[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int o = tid + 256 * e;
        if (a_k_fast) { am[e] = o >> 3; ak[e] = (o & 7) * 4; } else { ak[e] = o >> 4; am[e] = (o & 15) * 4; }
        if (b_k_fast) { bm[e] = o >> 3; bk[e] = (o & 7) * 4; } else { bk[e] = o >> 4; bm[e] = (o & 15) * 4; }
    }
    float4 ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ra[e] = gemm_load4(g.A, g.sai, g.sak, a_k_fast, a_vec, i0 + am[e], k0 + ak[e], g.I, k_hi);
            rb[e] = gemm_load4(g.B, g.sbn, g.sbk, b_k_fast, b_vec, j0 + bm[e], k0 + bk[e], g.J, k_hi);
        }
    };
    auto stage = [&](float (&S)[64][GBK + 1], bool k_fast, int m, int k, const float4 &v) {
        if (k_fast) { S[m][k] = v.x; S[m][k + 1] = v.y; S[m][k + 2] = v.z; S[m][k + 3] = v.w; }
        else { S[m][k] = v.x; S[m + 1][k] = v.y; S[m + 2][k] = v.z; S[m + 3][k] = v.w; }
    };
    fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += GBK) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            stage(As, a_k_fast, am[e], ak[e], ra[e]);
            stage(Bs, b_k_fast, bm[e], bk[e], rb[e]);
        }
        __syncthreads();
        if (k0 + GBK < k_hi) fetch(k0 + GBK);
#pragma unroll
        for (int t = 0; t < GBK / 2; ++t) acc = MFMA32(As[wi + q][2 * t + h], Bs[wj + q][2 * t + h], acc);
        __syncthreads();
    }
    const int j = j0 + wj + q;
    if (j >= g.J) return;
    const float bias_j = (g.mode == 0 && g.bias) ? g.bias[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (i >= g.I) continue;
        float *c = g.C + (long)i * g.ldc + j;
        if (g.mode == 0) {
            float v = acc[r] + bias_j;
            if (g.relu) v = fmaxf(v, 0.0f);
            if (g.add) v = g.add[(long)i * g.ldadd + j] + v;
            *c = v;
        } else if (g.mode == 1) {
            *c = *c + acc[r];
        } else {
            atomicAdd(c, acc[r]);
        }
    }
}

struct GemmLaunch {
    GemmBatch b;
    int blocks = 0;
    GemmLaunch() { b.count = 0; }
    GemmLaunch &add(const GemmArgs &g) {
        const int zs = (g.KK + g.kchunk - 1) / g.kchunk;
        b.g[b.count] = g;
        b.first[b.count++] = blocks;
        blocks += ((g.I + 63) / 64) * ((g.J + 63) / 64) * zs;
        return *this;
    }
    void run(hipStream_t s) {
        b.first[b.count] = blocks;
        hipLaunchKernelGGL(gemm_kernel, dim3(blocks), dim3(256), 0, s, b);
    }
};
// Y[M][N] = X[M][K] W[N][K]^T + b (ReLU) (+ add)
static GemmArgs linear_fwd(const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, int M, int N, int K,
                           bool relu, const float *add, int ldadd) {
    return GemmArgs{X, W, Y, b, add, M, N, K, ldx, 1, 1, K, ldy, ldadd, 0, relu ? 1 : 0, K};
}
// dX[M][K] (+)= dY[M][N] W[N][K]; mode 0 store, 1 add in place (no other writer of dX in flight), 2 atomic add
static GemmArgs linear_bwd_x(const float *dY, int ldy, const float *W, float *dX, int ldx, int M, int N, int K, int mode) {
    return GemmArgs{dY, W, dX, nullptr, nullptr, M, K, N, ldy, 1, K, 1, ldx, 0, mode, 0, N};
}
// dW[N][K] += dY[M][N]^T X[M][K]   (atomics over chunks of M)
static GemmArgs linear_bwd_w(const float *dY, int ldy, const float *X, int ldx, float *dW, int M, int N, int K) {
    return GemmArgs{dY, X, dW, nullptr, nullptr, N, K, M, 1, ldy, ldx, 1, K, 0, 2, 0, 512};
}

// db[j] += sum_i dY[i][j]: a workgroup sums 32 rows (thread = column, 4 row groups of 8, LDS reduction)
__global__ __launch_bounds__(512) void colsum_kernel(const float *__restrict__ dY, int ld, int M, int N, float *__restrict__ db) {
    __shared__ float part[4][128];
    const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const int j = blockIdx.x * 128 + c;
    const int i_lo = blockIdx.y * 32 + grp * 8, i_hi = min(M, i_lo + 8);
    float s = 0.0f;
    if (j < N)
        for (int i = i_lo; i < i_hi; ++i) s += dY[(long)i * ld + j];
    part[grp][c] = s;
    __syncthreads();
    if (grp == 0 && j < N) atomicAdd(db + j, (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}
static void bias_bwd(hipStream_t s, const float *dY, int ld, int M, int N, float *db) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(512), 0, s, dY, ld, M, N, db);
}

// dH *= (H > 0)
__global__ void relu_bwd_kernel(float *__restrict__ dH, const float *__restrict__ H, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dH[i] = H[i] > 0.0f ? dH[i] : 0.0f;
}

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
// One workgroup per (row, head), n <= 256 tokens, head h = columns 16h..16h+15 of Q, K, V [M][128]
// (TSP/TSPModel.py:420-441): P[i][j] = softmax_j(q_i.k_j / 4), O_i = sum_j P[i][j] V_j.
// The n x n probabilities are never stored: every pass recomputes q.k from the head's K (V, Q, dO)
// rows held in LDS and read as broadcasts; the backward pass gets the row maximum and the row sum
// of exponentials from the forward pass (stats [R][8][n][2]) and uses
//     sum_j P[i][j] dP[i][j] = dO_i . O_i.
__device__ __forceinline__ float dot16(const float (&a)[HDIM], const float *__restrict__ b) {
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < HDIM / 4; ++v) {
        const float4 t = b4[v];
        s = fmaf(a[4 * v], t.x, s);
        s = fmaf(a[4 * v + 1], t.y, s);
        s = fmaf(a[4 * v + 2], t.z, s);
        s = fmaf(a[4 * v + 3], t.w, s);
    }
    return s;
}
__device__ __forceinline__ void axpy16(float (&acc)[HDIM], float a, const float *__restrict__ b) {
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
#pragma unroll
    for (int v = 0; v < HDIM / 4; ++v) {
        const float4 t = b4[v];
        acc[4 * v] = fmaf(a, t.x, acc[4 * v]);
        acc[4 * v + 1] = fmaf(a, t.y, acc[4 * v + 1]);
        acc[4 * v + 2] = fmaf(a, t.z, acc[4 * v + 2]);
        acc[4 * v + 3] = fmaf(a, t.w, acc[4 * v + 3]);
    }
}
// copies the head's 16 columns of `src` rows tok0..tok0+n-1 into dst[n][16]
__device__ __forceinline__ void stage_head(const float *__restrict__ src, long tok0, int n, int hd, float *dst) {
    for (int e = threadIdx.x; e < n * (HDIM / 4); e += blockDim.x) {
        const int t = e / (HDIM / 4), v = e % (HDIM / 4);
        reinterpret_cast<float4 *>(dst)[e] =
            *reinterpret_cast<const float4 *>(src + (tok0 + t) * D_MODEL + hd * HDIM + 4 * v);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float *__restrict__ Q, const float *__restrict__ K,
                                                       const float *__restrict__ V, int n, float *__restrict__ stats,
                                                       float *__restrict__ O) {
    extern __shared__ float sm[];              // K[n][16], V[n][16]
    float *Ks = sm, *Vs = sm + n * HDIM;
    const int r = blockIdx.x, hd = blockIdx.y, i = threadIdx.x;
    const long tok0 = (long)r * n;
    stage_head(K, tok0, n, hd, Ks);
    stage_head(V, tok0, n, hd, Vs);
    __syncthreads();
    if (i >= n) return;
    float q[HDIM];
#pragma unroll
    for (int d = 0; d < HDIM; ++d) q[d] = Q[(tok0 + i) * D_MODEL + hd * HDIM + d];
    // One pass over the keys with a running maximum: whenever it grows, the sum and the output accumulated so
    // far are rescaled (exp(old - new) <= 1).  Equal to softmax(s) V up to rounding.
    float mx = -INFINITY, sum = 0.0f;
    float o[HDIM];
#pragma unroll
    for (int d = 0; d < HDIM; ++d) o[d] = 0.0f;
    for (int j = 0; j < n; ++j) {
        const float sc = dot16(q, Ks + j * HDIM) / 4.0f;
        if (sc > mx) {
            const float f = expf(mx - sc);          // 0 on the first key (mx = -inf)
            sum *= f;
#pragma unroll
            for (int d = 0; d < HDIM; ++d) o[d] *= f;
            mx = sc;
        }
        const float p = expf(sc - mx);
        sum += p;
        axpy16(o, p, Vs + j * HDIM);
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int d = 0; d < HDIM; ++d) O[(tok0 + i) * D_MODEL + hd * HDIM + d] = o[d] * inv;
    float *st = stats + (((long)r * HEADS + hd) * n + i) * 2;
    st[0] = mx;
    st[1] = sum;
}

// Backward of the above.  dO, O [M][128] -> dQ, dK, dV [M][128].
//   dS[i][j] = P[i][j] (dO_i . V_j - dO_i . O_i) / 4 (gradient with respect to q.k, the 1/4 included)
//   dQ_i = sum_j dS[i][j] K_j;  dK_j = sum_i dS[i][j] Q_i;  dV_j = sum_i P[i][j] dO_i
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float *__restrict__ Q, const float *__restrict__ K,
                                                       const float *__restrict__ V, const float *__restrict__ O,
                                                       const float *__restrict__ dO, int n,
                                                       const float *__restrict__ stats, float *__restrict__ dQ,
                                                       float *__restrict__ dK, float *__restrict__ dV) {
    extern __shared__ float sm[];              // Q, K, V, dO: [n][16] each; row max, 1 / row sum, dO.O: [n] each
    float *Qs = sm, *Ks = Qs + n * HDIM, *Vs = Ks + n * HDIM, *Gs = Vs + n * HDIM;
    float *Mx = Gs + n * HDIM, *Inv = Mx + n, *Dt = Inv + n;
    const int r = blockIdx.x, hd = blockIdx.y, i = threadIdx.x;
    const long tok0 = (long)r * n;
    stage_head(Q, tok0, n, hd, Qs);
    stage_head(K, tok0, n, hd, Ks);
    stage_head(V, tok0, n, hd, Vs);
    stage_head(dO, tok0, n, hd, Gs);
    const bool live = i < n;
    if (live) {
        const float *st = stats + (((long)r * HEADS + hd) * n + i) * 2;
        Mx[i] = st[0];
        Inv[i] = 1.0f / st[1];
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < HDIM; ++c) {
            const long g = (tok0 + i) * D_MODEL + hd * HDIM + c;
            d = fmaf(dO[g], O[g], d);
        }
        Dt[i] = d;
    }
    __syncthreads();
    if (!live) return;
    {   // thread = query i
        float q[HDIM], go[HDIM], a[HDIM];
#pragma unroll
        for (int d = 0; d < HDIM; ++d) { q[d] = Qs[i * HDIM + d]; go[d] = Gs[i * HDIM + d]; a[d] = 0.0f; }
        const float mx = Mx[i], inv = Inv[i], dt = Dt[i];
        for (int j = 0; j < n; ++j) {
            const float p = expf(dot16(q, Ks + j * HDIM) / 4.0f - mx) * inv;
            const float ds = p * (dot16(go, Vs + j * HDIM) - dt) / 4.0f;
            axpy16(a, ds, Ks + j * HDIM);
        }
#pragma unroll
        for (int d = 0; d < HDIM; ++d) dQ[(tok0 + i) * D_MODEL + hd * HDIM + d] = a[d];
    }
    {   // thread = key j (= i)
        float k[HDIM], v[HDIM], ak[HDIM], av[HDIM];
#pragma unroll
        for (int d = 0; d < HDIM; ++d) { k[d] = Ks[i * HDIM + d]; v[d] = Vs[i * HDIM + d]; ak[d] = 0.0f; av[d] = 0.0f; }
        for (int t = 0; t < n; ++t) {
            const float p = expf(dot16(k, Qs + t * HDIM) / 4.0f - Mx[t]) * Inv[t];
            const float ds = p * (dot16(v, Gs + t * HDIM) - Dt[t]) / 4.0f;
            axpy16(av, p, Gs + t * HDIM);
            axpy16(ak, ds, Qs + t * HDIM);
        }
#pragma unroll
        for (int d = 0; d < HDIM; ++d) {
            dK[(tok0 + i) * D_MODEL + hd * HDIM + d] = ak[d];
            dV[(tok0 + i) * D_MODEL + hd * HDIM + d] = av[d];
        }
    }
}

// ------------------------------------------------------------------------------ anchors
// Decoder entry (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-449): Y = X with the anchor token(s)
// of every row replaced by  W_a x_anchor + b_a.  One workgroup per (row, anchor), 128 threads.
__global__ __launch_bounds__(128) void anchor_fwd_kernel(const float *__restrict__ X, float *__restrict__ Y, int n, int tok0_,
                                                         int tok1_, const float *__restrict__ W0, const float *__restrict__ b0,
                                                         const float *__restrict__ W1, const float *__restrict__ b1) {
    __shared__ float xs[D_MODEL];
    const int r = blockIdx.x, a = blockIdx.y, c = threadIdx.x;
    const long tk = (long)r * n + (a == 0 ? tok0_ : tok1_);
    const float *W = a == 0 ? W0 : W1, *b = a == 0 ? b0 : b1;
    xs[c] = X[tk * D_MODEL + c];
    __syncthreads();
    float s = 0.0f;
    for (int k = 0; k < D_MODEL; ++k) s = fmaf(W[c * D_MODEL + k], xs[k], s);
    Y[tk * D_MODEL + c] = s + b[c];
}
// dX = dY except at the anchors, where dX_anchor = W_a^T dY_anchor; db_a += dY_anchor
// (dW_a += dY_anchor x_anchor^T runs before this kernel as a strided GEMM over the rows' anchor tokens)
__global__ __launch_bounds__(128) void anchor_bwd_kernel(const float *__restrict__ X, float *__restrict__ dY, int n, int tok0_,
                                                         int tok1_, const float *__restrict__ W0, const float *__restrict__ W1,
                                                         float *__restrict__ db0, float *__restrict__ db1) {
    __shared__ float gs[D_MODEL];
    const int r = blockIdx.x, a = blockIdx.y, c = threadIdx.x;
    const long tk = (long)r * n + (a == 0 ? tok0_ : tok1_);
    const float *W = a == 0 ? W0 : W1;
    float *db = a == 0 ? db0 : db1;
    gs[c] = dY[tk * D_MODEL + c];
    __syncthreads();
    float s = 0.0f;
    for (int o = 0; o < D_MODEL; ++o) s = fmaf(W[o * D_MODEL + c], gs[o], s);
    dY[tk * D_MODEL + c] = s;                                     // becomes dX in place
    atomicAdd(db + c, gs[c]);
}

// -------------------------------------------------------------------------- output stage
// Per row: logits = X wf + bf, masking, softmax, floor (TSP/TSPModel.py:312-326,
// CVRP/CVRPModel.py:462-493), the outputs of dgl_policy_forward, and for training the teacher's
// probability (1e-20 when the teacher's action is not a candidate, as in the dense tensor the
// reference indexes) and dX = d(-mean log p_teacher)/dX, dwf, dbf.  One wave per row.
__global__ __launch_bounds__(64) void output_kernel(const float *__restrict__ X, const float *__restrict__ wf,
                                                    const float *__restrict__ bf, dgl_policy_io_t io,
                                                    const int32_t *__restrict__ teacher, float grad_scale,
                                                    float *__restrict__ teacher_prob, float *__restrict__ dX,
                                                    float *__restrict__ dwf, float *__restrict__ dbf) {
    extern __shared__ float lg[];              // [n] logits, then probabilities
    const int r = blockIdx.x, lane = threadIdx.x, n = io.n;
    const int stride = io.out_stride > 0 ? io.out_stride : DGL_MAX_TOKENS;
    const int anchor0 = io.k_last, anchor1 = n - 1;
    const float *xr = X + (long)r * n * D_MODEL;
    auto is_cand = [&](int t) { return t != anchor0 && t != anchor1; };
    auto is_open = [&](int t) {
        if (!is_cand(t)) return false;
        if (io.problem == DGL_PROBLEM_CVRP && io.infeasible != nullptr && t < io.k_last)
            return io.infeasible[(size_t)r * io.idx_stride + t] == 0;
        return true;
    };
    auto action = [&](int t) {
        if (io.problem == DGL_PROBLEM_TSP || t < io.k_last) return io.idx_last[(size_t)r * io.idx_stride + t];
        return io.idx_depot[(size_t)r * io.idx_dep_stride + (t - io.k_last - 1)] + io.N + 1;
    };
    float mx = -INFINITY;
    for (int t = lane; t < n; t += 64) {
        float s = 0.0f;
        for (int c = 0; c < D_MODEL; ++c) s = fmaf(xr[t * D_MODEL + c], wf[c], s);
        s += bf[0];
        lg[t] = s;
        if (io.logits_out) io.logits_out[(size_t)r * stride + t] = s;
        if (is_open(t)) mx = fmaxf(mx, s);
    }
    mx = wave_max_f32(mx);
    float sum = 0.0f;
    for (int t = lane; t < n; t += 64) {
        const float e = is_open(t) ? expf(lg[t] - mx) : 0.0f;
        lg[t] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float best = -1.0f;
    int best_act = 0x7FFFFFFF, t_teacher = -1;
    const int want = teacher ? teacher[r] : -1;
    for (int t = lane; t < n; t += 64) {
        const float p = lg[t] / sum;                 // pure softmax (kept for the gradient)
        lg[t] = p;
        float pf = p;
        if (pf <= 1e-5f) pf = pf + 1e-7f;
        if (!is_cand(t)) pf = 0.0f;
        if (io.prob_out) io.prob_out[(size_t)r * stride + t] = pf;
        if (is_cand(t)) {
            const int act = action(t);
            if (pf > best || (pf == best && act < best_act)) { best = pf; best_act = act; }
            if (act == want) t_teacher = t;
        }
    }
    // greedy pick: highest probability, lowest action id on ties
    float wbest = wave_max_f32(best);
    int cand_act = (best == wbest) ? best_act : 0x7FFFFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand_act = min(cand_act, __shfl_xor(cand_act, o, 64));
    if (lane == 0 && io.selected_out) io.selected_out[r] = cand_act;
    if (!teacher) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t_teacher = max(t_teacher, __shfl_xor(t_teacher, o, 64));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float pt = 1e-20f, g_logit_scale = 0.0f, p_soft = 0.0f;
    if (t_teacher >= 0) {
        p_soft = lg[t_teacher];
        pt = p_soft <= 1e-5f ? p_soft + 1e-7f : p_soft;
        // L = -grad_scale^-1... : dL/dp_t = -grad_scale / p_t (grad_scale = 1 / rows of the global batch)
        g_logit_scale = -grad_scale / pt;
    }
    if (lane == 0 && teacher_prob) teacher_prob[r] = pt;
    if (!dX) return;
    float dbsum = 0.0f;
    for (int t = lane; t < n; t += 64) {
        // d p_t / d logit_j = p_t (delta_tj - p_j) over the open tokens
        float dl = 0.0f;
        if (t_teacher >= 0 && is_open(t)) dl = g_logit_scale * p_soft * ((t == t_teacher ? 1.0f : 0.0f) - lg[t]);
        lg[t] = dl;
        dbsum += dl;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dbsum += __shfl_xor(dbsum, o, 64);
    if (lane == 0 && dbsum != 0.0f) atomicAdd(dbf, dbsum);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < D_MODEL; c += 64) {
        float dw = 0.0f;
        const float w = wf[c];
        for (int t = 0; t < n; ++t) {
            const float dl = lg[t];
            dX[((long)r * n + t) * D_MODEL + c] = dl * w;
            dw = fmaf(dl, xr[t * D_MODEL + c], dw);
        }
        if (dw != 0.0f) atomicAdd(dwf + c, dw);
    }
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
//   T [M][12] | X[0..4] [M][128] (input of block l; X[4] = final) | E [M][128] (encoder output before the
//   anchor re-embedding) | per block: Q K V AO O1 [M][128] each, H [M][512], attention row statistics [R][8][n][2]
//   | backward scratch: dA dB dQ dK dV [M][128] each, dH [M][512]
static size_t ws_floats(int R, int n) {
    const size_t M = (size_t)R * n;
    const size_t per_block = M * (5 * D_MODEL + FF_DIM) + (size_t)R * HEADS * n * 2;
    return M * DGL_TILE_STRIDE + 6 * M * D_MODEL + 4 * per_block + 5 * M * D_MODEL + M * FF_DIM;
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

extern "C" int dgl_policy_general(const dgl_policy_weights_t *w, const dgl_policy_weights_t *grad,
                                  const dgl_policy_io_t *io, const int32_t *teacher, float grad_scale, float *ws,
                                  float *teacher_prob_out, dgl_stream_t stream) {
    if (int e = check_general(w, io, ws)) return e;
    DGL_CHECK_ARG(grad == nullptr || teacher != nullptr, "gradients need the teacher's actions");
    hipStream_t s = (hipStream_t)stream;
    const int R = io->R, n = io->n, F = io->problem == DGL_PROBLEM_TSP ? 7 : 9;
    const int M = R * n;
    const size_t MD = (size_t)M * D_MODEL, MF = (size_t)M * FF_DIM, PP = (size_t)R * HEADS * n * 2;
    float *T = ws;
    float *X[5];
    float *p = T + (size_t)M * DGL_TILE_STRIDE;
    for (int l = 0; l < 5; ++l) { X[l] = p; p += MD; }
    float *E = p; p += MD;
    float *Qb[4], *Kb[4], *Vb[4], *AO[4], *O1[4], *H[4], *P[4];
    for (int l = 0; l < 4; ++l) {
        Qb[l] = p; p += MD; Kb[l] = p; p += MD; Vb[l] = p; p += MD; AO[l] = p; p += MD; O1[l] = p; p += MD;
        H[l] = p; p += MF; P[l] = p; p += PP;
    }
    float *dA = p; p += MD;
    float *dB = p; p += MD;
    float *dQ = p; p += MD;
    float *dK = p; p += MD;
    float *dV = p; p += MD;
    float *dH = p; p += MF;
    const int anchor0 = io->k_last, anchor1 = n - 1;
    const bool cvrp = io->problem == DGL_PROBLEM_CVRP;
    const int n_anchor = cvrp ? 2 : 1;
    const size_t attn_lds = (size_t)2 * n * HDIM * sizeof(float), attn_bwd_lds = 2 * attn_lds + (size_t)3 * n * sizeof(float);
    if (attn_bwd_lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);

    // ---- forward
    hipLaunchKernelGGL(gather_tile_kernel, dim3(min((M * DGL_TILE_STRIDE + 255) / 256, 2048)), dim3(256), 0, s, io->tile,
                       io->tile_row_stride, R, n, T);
    GemmLaunch().add(linear_fwd(T, DGL_TILE_STRIDE, w->emb_w, w->emb_b, X[0], D_MODEL, M, D_MODEL, F, false, nullptr, 0)).run(s);
    for (int l = 0; l < 4; ++l) {
        const dgl_block_weights_t &k = w->block[l];
        const float *Xin = X[l];
        if (l == 1) {
            // E keeps the encoder output; X[1] = E with re-embedded anchors
            (void)hipMemcpyAsync(E, X[1], MD * sizeof(float), hipMemcpyDeviceToDevice, s);
            hipLaunchKernelGGL(anchor_fwd_kernel, dim3(R, n_anchor), dim3(128), 0, s, E, X[1], n, anchor0, anchor1,
                               w->anchor_w[0], w->anchor_b[0], w->anchor_w[cvrp ? 1 : 0], w->anchor_b[cvrp ? 1 : 0]);
        }
        GemmLaunch()
            .add(linear_fwd(Xin, D_MODEL, k.wq, nullptr, Qb[l], D_MODEL, M, D_MODEL, D_MODEL, false, nullptr, 0))
            .add(linear_fwd(Xin, D_MODEL, k.wk, nullptr, Kb[l], D_MODEL, M, D_MODEL, D_MODEL, false, nullptr, 0))
            .add(linear_fwd(Xin, D_MODEL, k.wv, nullptr, Vb[l], D_MODEL, M, D_MODEL, D_MODEL, false, nullptr, 0))
            .run(s);
        hipLaunchKernelGGL(attn_fwd_kernel, dim3(R, HEADS), dim3(256), attn_lds, s, Qb[l], Kb[l], Vb[l], n, P[l], AO[l]);
        GemmLaunch().add(linear_fwd(AO[l], D_MODEL, k.wc, k.bc, O1[l], D_MODEL, M, D_MODEL, D_MODEL, false, Xin, D_MODEL)).run(s);
        GemmLaunch().add(linear_fwd(O1[l], D_MODEL, k.w1, k.b1, H[l], FF_DIM, M, FF_DIM, D_MODEL, true, nullptr, 0)).run(s);
        GemmLaunch().add(linear_fwd(H[l], FF_DIM, k.w2, k.b2, X[l + 1], D_MODEL, M, D_MODEL, FF_DIM, false, O1[l], D_MODEL)).run(s);
    }
    const bool train = grad != nullptr;
    hipLaunchKernelGGL(output_kernel, dim3(R), dim3(64), (size_t)n * sizeof(float), s, X[4], w->final_w, w->final_b, *io,
                       teacher, grad_scale, teacher_prob_out, train ? dA : nullptr, train ? (float *)grad->final_w : nullptr,
                       train ? (float *)grad->final_b : nullptr);
    DGL_CHECK_LAUNCH("dgl_policy_general (forward)");
    if (!train) return DGL_OK;

    // ---- backward: dA holds the gradient of the current block's output
    for (int l = 3; l >= 0; --l) {
        const dgl_block_weights_t &k = w->block[l];
        const dgl_block_weights_t &g = grad->block[l];
        const float *Xin = X[l];
        // Products that share no output run in one launch each.
        // X[l+1] = O1 + W2 H + b2
        bias_bwd(s, dA, D_MODEL, M, D_MODEL, (float *)g.b2);
        GemmLaunch()
            .add(linear_bwd_w(dA, D_MODEL, H[l], FF_DIM, (float *)g.w2, M, D_MODEL, FF_DIM))
            .add(linear_bwd_x(dA, D_MODEL, k.w2, dH, FF_DIM, M, D_MODEL, FF_DIM, 0))
            .run(s);
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(2048), dim3(256), 0, s, dH, H[l], (long)MF);
        // H = relu(W1 O1 + b1);  dO1 = dA + dHpre W1   (accumulated into dA)
        bias_bwd(s, dH, FF_DIM, M, FF_DIM, (float *)g.b1);
        GemmLaunch()
            .add(linear_bwd_w(dH, FF_DIM, O1[l], D_MODEL, (float *)g.w1, M, FF_DIM, D_MODEL))
            .add(linear_bwd_x(dH, FF_DIM, k.w1, dA, D_MODEL, M, FF_DIM, D_MODEL, 1))
            .run(s);
        // O1 = Xin + Wc AO + bc
        bias_bwd(s, dA, D_MODEL, M, D_MODEL, (float *)g.bc);
        GemmLaunch()
            .add(linear_bwd_w(dA, D_MODEL, AO[l], D_MODEL, (float *)g.wc, M, D_MODEL, D_MODEL))
            .add(linear_bwd_x(dA, D_MODEL, k.wc, dB, D_MODEL, M, D_MODEL, D_MODEL, 0))       // dB = dAO
            .run(s);
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(R, HEADS), dim3(256), attn_bwd_lds, s, Qb[l], Kb[l], Vb[l], AO[l], dB, n, P[l], dQ,
                           dK, dV);
        // dXin = dA + dQ Wq + dK Wk + dV Wv: three products into one output, added atomically
        GemmLaunch()
            .add(linear_bwd_w(dQ, D_MODEL, Xin, D_MODEL, (float *)g.wq, M, D_MODEL, D_MODEL))
            .add(linear_bwd_w(dK, D_MODEL, Xin, D_MODEL, (float *)g.wk, M, D_MODEL, D_MODEL))
            .add(linear_bwd_w(dV, D_MODEL, Xin, D_MODEL, (float *)g.wv, M, D_MODEL, D_MODEL))
            .add(linear_bwd_x(dQ, D_MODEL, k.wq, dA, D_MODEL, M, D_MODEL, D_MODEL, 2))
            .add(linear_bwd_x(dK, D_MODEL, k.wk, dA, D_MODEL, M, D_MODEL, D_MODEL, 2))
            .add(linear_bwd_x(dV, D_MODEL, k.wv, dA, D_MODEL, M, D_MODEL, D_MODEL, 2))
            .run(s);
        if (l == 1) {
            // dW_a[c][k] += sum_r dA[anchor of row r][c] E[anchor of row r][k]: the rows' anchor tokens are n tokens apart
            GemmLaunch gl;
            for (int a = 0; a < n_anchor; ++a) {
                const long off = (long)(a == 0 ? anchor0 : anchor1) * D_MODEL;
                gl.add(GemmArgs{dA + off, E + off, (float *)grad->anchor_w[a], nullptr, nullptr, D_MODEL, D_MODEL, R, 1,
                                (long)n * D_MODEL, (long)n * D_MODEL, 1, D_MODEL, 0, 2, 0, 512});
            }
            gl.run(s);
            hipLaunchKernelGGL(anchor_bwd_kernel, dim3(R, n_anchor), dim3(128), 0, s, E, dA, n, anchor0, anchor1, w->anchor_w[0],
                               w->anchor_w[cvrp ? 1 : 0], (float *)grad->anchor_b[0], (float *)grad->anchor_b[cvrp ? 1 : 0]);
        }
    }
    // X[0] = T We^T + be
    GemmLaunch().add(linear_bwd_w(dA, D_MODEL, T, DGL_TILE_STRIDE, (float *)grad->emb_w, M, D_MODEL, F)).run(s);
    bias_bwd(s, dA, D_MODEL, M, D_MODEL, (float *)grad->emb_b);
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
// All calls on one device share the array: calls that overlap in time on one device race.  Nothing
// in the workspace has to start at zero (every buffer is written in full before it is read), so the
// previous call's contents are simply overwritten.
static __device__ __align__(256) float scratch[DGL_STATIC_WORKSPACE_FLOATS];

#define DGL_MAX_DEVICES 64
static float *scratch_ptr[DGL_MAX_DEVICES];          // per device, filled on the first call on that device
static std::mutex scratch_mutex;

static int static_workspace(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
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
    if (int e = static_workspace(&ws)) return e;
    return dgl_policy_general(w, grad, io, teacher, grad_scale, ws, teacher_prob_out, stream);
}
