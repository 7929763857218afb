// Label handling of the self-improvement training loop: the nearest-neighbour construction the
// trainers start from (TSP/TSPEnv.py:140-169, CVRP/CVRPEnv.py:340-409) and the label augmentation
// of the CVRP trainer (sub-route reversal and rotation, CVRP/CVRPEnv.py:325-338).  The reference
// runs these as host loops over a dense distance matrix; here one workgroup owns an instance and
// keeps its coordinates in LDS.
#include "common.h"
#include "state_common.h"

namespace {

constexpr int NN_THREADS = 256;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}

// Nearest unvisited node, repeated until every node is placed.  Distances are the entries of the
// reference's cdist matrix (row = current node), so the choices are the reference's; equal
// distances go to the lower index.  M nodes per instance; CVRP: node 0 is the depot and never
// chosen, a move goes via the depot (flag 1, capacity refilled) when the load does not fit
// (capacity + 1e-6 < demand), the first move always does.
// dynamic LDS: float2 pts[M], float nrm[M] (|p|^2, -1 once visited), float edge[M] (TSP under torch's summation).
// M <= DGL_DIRECT_MAX_POINTS: cdist's direct evaluation (common.h, dgl_direct_sq); nrm then only marks the visited.
// SQ: the square-root model of the matrix entries (common.h).  Under the reference-bits model the TSP label's length is
// torch.sum's float over the chosen entries in tour order (TSP/TSPEnv.py:167: dgl_torch_row_sum), else their binary64
// sum rounded once.
template <bool CVRP, class SQ>
__global__ __launch_bounds__(NN_THREADS) void nearest_neighbour_kernel(const float *__restrict__ xy,
                                                                       const float *__restrict__ demand, int M,
                                                                       int32_t *__restrict__ nodes,
                                                                       int8_t *__restrict__ flags,
                                                                       float *__restrict__ length, SQ sq) {
    constexpr bool TORCH_SUM = SQ::WAVE_WIDE;                  // (the reference-bits model)
    extern __shared__ float lds[];
    __shared__ unsigned long long red[2][NN_THREADS / 64];
    float2 *pts = reinterpret_cast<float2 *>(lds);
    float *nrm = lds + 2 * (size_t)M;
    float *edge = nrm + M;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float2 *in = reinterpret_cast<const float2 *>(xy) + (size_t)b * M;
    const bool direct = M <= DGL_DIRECT_MAX_POINTS;
    for (int j = tid; j < M; j += NN_THREADS) {
        const float2 p = in[j];
        pts[j] = p;
        nrm[j] = j == 0 ? -1.0f : dgl_sqnorm(p.x, p.y);       // node 0: start of the tour / the depot
    }
    __syncthreads();
    const int count = CVRP ? M - 1 : M;                        // entries of the label
    int32_t *out = nodes + (size_t)b * count;
    int8_t *fl = CVRP ? flags + (size_t)b * count : nullptr;
    const float *dem = CVRP ? demand + (size_t)b * M : nullptr;
    int pos = 0;
    if (!CVRP) {
        if (tid == 0) out[0] = 0;
        pos = 1;
    }
    int cur = 0;
    float capacity = 1.0f;
    double total = 0.0;
    for (int it = 0; pos < count; ++pos, ++it) {
        const float2 pc = pts[cur];
        const float nc = dgl_sqnorm(pc.x, pc.y);
        unsigned long long best = ~0ull;
        for (int j = tid; j < M; j += NN_THREADS) {
            const float nj = nrm[j];
            if (nj < 0.0f) continue;
            const float2 pj = pts[j];
            const float d = direct ? dgl_sqrt(dgl_direct_sq(pc.x, pc.y, pj.x, pj.y))
                                   : sq(dgl_chain_sq(dgl_chain_cross(pc.x, pc.y, pj.x, pj.y), nc, nj));
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
            best = key < best ? key : best;
        }
        best = wave_min_u64(best);
        if ((tid & 63) == 0) red[it & 1][tid >> 6] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NN_THREADS / 64; ++w) {
            const unsigned long long u = red[it & 1][w];
            best = u < best ? u : best;
        }
        const int j = (int)(unsigned)best;                     // every thread holds the same choice
        if (j % NN_THREADS == tid) nrm[j] = -1.0f;             // only this thread ever reads nrm[j] again
        if (CVRP) {
            const float d = dem[j];
            const bool via = pos == 0 || __fadd_rn(capacity, 0.000001f) < d;
            capacity = via ? __fsub_rn(1.0f, d) : __fsub_rn(capacity, d);
            if (tid == 0) {
                out[pos] = j;
                fl[pos] = via ? 1 : 0;
            }
        } else {
            total += (double)__uint_as_float((unsigned)(best >> 32));
            if (tid == 0) {
                out[pos] = j;
                if (TORCH_SUM) edge[pos - 1] = __uint_as_float((unsigned)(best >> 32));
            }
        }
        cur = j;
    }
    if (!CVRP) {
        const float2 pc = pts[cur], p0 = pts[0];               // closing edge: row = last node, column = node 0
        const float closing = direct ? dgl_sqrt(dgl_direct_sq(pc.x, pc.y, p0.x, p0.y))
                                     : sq(dgl_chain_sq(dgl_chain_cross(pc.x, pc.y, p0.x, p0.y), dgl_sqnorm(pc.x, pc.y),
                                                       dgl_sqnorm(p0.x, p0.y)));
        if (!TORCH_SUM) {
            if (tid == 0) length[b] = (float)(total + (double)closing);
        } else {
            if (tid == 0) edge[M - 1] = closing;
            __syncthreads();
            if (tid < 64) {                                    // (both halves of the wave sum the same row)
                const float sum = dgl_torch_row_sum(M, [&](int p, auto) { return edge[p]; });
                if (tid == 0) length[b] = sum;
            }
        }
    }
}

// Sub-route reversal and rotation of a CVRP label.  Sub-route s (s = 1..S) spans the positions from
// the s-th set flag up to the next one; sub-routes 1..S-1 are reversed when flip[b][s] is set (the
// last one never is), then label and flags are rotated so that sub-route 1 + (shift[b] mod S)
// comes first.  dynamic LDS: N node ids, then ceil(N/64)+1 64-bit words (flag bits, their prefix counts).
__global__ __launch_bounds__(256) void solution_augment_kernel(const int32_t *__restrict__ nodes,
                                                               const int8_t *__restrict__ flags, int N,
                                                               const uint8_t *__restrict__ flip,
                                                               const int32_t *__restrict__ shift,
                                                               int32_t *__restrict__ nodes_out,
                                                               int8_t *__restrict__ flags_out) {
    extern __shared__ unsigned long long lds64[];
    const int W = (N + 63) / 64;
    unsigned long long *bits = lds64;                          // [W]
    int *prefix = reinterpret_cast<int *>(lds64 + W);          // [W + 1] set flags before word w
    int *moved = prefix + W + 2;                               // [N] label after the reversals
    __shared__ int offset;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int32_t *in = nodes + (size_t)b * N;
    const int8_t *fin = flags + (size_t)b * N;
    for (int w0 = (tid >> 6); w0 < W; w0 += 4) {
        const int p = w0 * 64 + lane;
        const unsigned long long m = __ballot(p < N && fin[p] != 0);
        if (lane == 0) bits[w0] = m;
    }
    if (tid == 0) offset = 0;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int w = 0; w < W; ++w) {
            prefix[w] = s;
            s += __popcll(bits[w]);
        }
        prefix[W] = s;
    }
    __syncthreads();
    const int S = prefix[W];
    const int want = S > 0 ? (int)((unsigned)shift[b] % (unsigned)S) : 0;     // 0-based sub-route that comes first
    for (int p = tid; p < N; p += 256) {
        const int w = p >> 6, o = p & 63;
        const unsigned long long below = bits[w] & ((o == 63) ? ~0ull : ((2ull << o) - 1));   // flags at positions <= p
        const int rank = prefix[w] + __popcll(below);          // 1-based sub-route of position p (0: before the first flag)
        int src = p;
        if (rank >= 1 && rank < S && flip[(size_t)b * N + rank]) {
            int s = p;                                         // first position of the sub-route
            while (!((bits[s >> 6] >> (s & 63)) & 1ull)) --s;
            int e = p + 1;                                     // one past its last position
            while (e < N && !((bits[e >> 6] >> (e & 63)) & 1ull)) ++e;
            src = s + (e - 1 - p);
        }
        moved[p] = in[src];
        if (((bits[w] >> o) & 1ull) && rank - 1 == want) offset = p;
    }
    __syncthreads();
    const int off = offset;
    for (int q = tid; q < N; q += 256) {
        int p = q + off;
        if (p >= N) p -= N;
        nodes_out[(size_t)b * N + q] = moved[p];
        flags_out[(size_t)b * N + q] = fin[p];
    }
}

}  // namespace

extern "C" int dgl_tsp_nearest_neighbour(const float *xy, int32_t B, int32_t N, int32_t *solution, float *length,
                                         dgl_stream_t stream) {
    DGL_CHECK_ARG(xy && solution && length && B > 0 && N > 1, "bad argument");
    const size_t lds = (size_t)N * 16;
    DGL_CHECK_ARG(lds <= 150 * 1024, "more than 9600 nodes per instance do not fit the workgroup's LDS");
    if (dgl_sqrt_reference_bits()) {
        dgl_max_dynamic_lds(reinterpret_cast<const void *>(nearest_neighbour_kernel<false, DglSqrtMkl>), (int)lds);
        hipLaunchKernelGGL((nearest_neighbour_kernel<false, DglSqrtMkl>), dim3(B), dim3(NN_THREADS), lds, (hipStream_t)stream, xy,
                           (const float *)nullptr, N, solution, (int8_t *)nullptr, length, DglSqrtMkl{});
    } else {
        dgl_max_dynamic_lds(reinterpret_cast<const void *>(nearest_neighbour_kernel<false, DglSqrtIeee>), (int)lds);
        hipLaunchKernelGGL((nearest_neighbour_kernel<false, DglSqrtIeee>), dim3(B), dim3(NN_THREADS), lds, (hipStream_t)stream, xy,
                           (const float *)nullptr, N, solution, (int8_t *)nullptr, length, DglSqrtIeee{});
    }
    DGL_CHECK_LAUNCH("dgl_tsp_nearest_neighbour");
    return DGL_OK;
}

extern "C" int dgl_cvrp_nearest_neighbour(const float *xy, const float *demand, int32_t B, int32_t N, int32_t *nodes,
                                          int8_t *flags, dgl_stream_t stream) {
    DGL_CHECK_ARG(xy && demand && nodes && flags && B > 0 && N > 0, "bad argument");
    const size_t lds = (size_t)(N + 1) * 16;
    DGL_CHECK_ARG(lds <= 150 * 1024, "more than 9600 nodes per instance do not fit the workgroup's LDS");
    if (dgl_sqrt_reference_bits()) {
        dgl_max_dynamic_lds(reinterpret_cast<const void *>(nearest_neighbour_kernel<true, DglSqrtMkl>), (int)lds);
        hipLaunchKernelGGL((nearest_neighbour_kernel<true, DglSqrtMkl>), dim3(B), dim3(NN_THREADS), lds, (hipStream_t)stream, xy, demand,
                           N + 1, nodes, flags, (float *)nullptr, DglSqrtMkl{});
    } else {
        dgl_max_dynamic_lds(reinterpret_cast<const void *>(nearest_neighbour_kernel<true, DglSqrtIeee>), (int)lds);
        hipLaunchKernelGGL((nearest_neighbour_kernel<true, DglSqrtIeee>), dim3(B), dim3(NN_THREADS), lds, (hipStream_t)stream, xy, demand,
                           N + 1, nodes, flags, (float *)nullptr, DglSqrtIeee{});
    }
    DGL_CHECK_LAUNCH("dgl_cvrp_nearest_neighbour");
    return DGL_OK;
}

extern "C" int dgl_cvrp_solution_augment(const int32_t *nodes, const int8_t *flags, int32_t B, int32_t N,
                                         const uint8_t *flip, const int32_t *shift, int32_t *nodes_out,
                                         int8_t *flags_out, dgl_stream_t stream) {
    DGL_CHECK_ARG(nodes && flags && flip && shift && nodes_out && flags_out && B > 0 && N > 0, "bad argument");
    DGL_CHECK_ARG(nodes != nodes_out && flags != flags_out, "the augmentation does not run in place");
    const int W = (N + 63) / 64;
    const size_t lds = (size_t)W * 8 + (size_t)(W + 2) * 4 + (size_t)N * 4;
    DGL_CHECK_ARG(lds <= 150 * 1024, "label too long for the workgroup's LDS");
    dgl_max_dynamic_lds(reinterpret_cast<const void *>(solution_augment_kernel), (int)lds);
    hipLaunchKernelGGL(solution_augment_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, nodes, flags, N, flip,
                       shift, nodes_out, flags_out);
    DGL_CHECK_LAUNCH("dgl_cvrp_solution_augment");
    return DGL_OK;
}
