// TSP environment state on the device: reset, fused step + neighbourhood + token tile,
// dense state materialisation, tour length, beam permutation.
//
// Layout: one wave64 owns one row (one partial tour).  Coordinates are read from the
// instance ([B][N][2], shared by the rows of an instance), the running features avg/sd
// are [R][N] binary32, the visited set is a bitmask.  No [N][N] distance matrix and no
// [R][N][7] feature tensor exist: distances are recomputed with the reference's exact
// arithmetic wherever they are needed.
#include <cstdlib>

#include "tsp_step_body.h"

// --------------------------------------------------------------------------- reset
// grid: (ceil(N / 256), B), block 256 (see dgl_reset_rows in state_common.h)
__global__ __launch_bounds__(256) void tsp_reset_kernel(dgl_tsp_state_t st) {
    dgl_reset_rows(st.xy, nullptr, st.N, st.rows_per_instance, st.avg, st.sd, st.visited);
}

// the reset of the square-root models that reproduce the reference's torch build bit for bit: grid (ceil(N / 8), B)
__global__ __launch_bounds__(256) void tsp_reset_torch_kernel(dgl_tsp_state_t st, DglSqrtMkl sq) {
    dgl_reset_rows_torch(st.xy, nullptr, st.N, st.rows_per_instance, st.avg, st.sd, st.visited, sq);
}

// ---------------------------------------------------------------------------- step
// (body: tsp_step_body.h)  One wave per row, or the WAVES waves of a workgroup sharing one row (SPLIT).
template <class SQ, int WAVES, bool UPDATE, bool SPLIT = false, bool STREAM = false>
__global__ __launch_bounds__(WAVES * 64) void tsp_step_kernel(dgl_tsp_state_t st, const int32_t *__restrict__ selected,
                                                             int t, int knn, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ tile_out, int lds_per_wave, int mom_off,
                                                             TspBeamLink link, SQ sq) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar registers
    const int r = SPLIT ? blockIdx.x : blockIdx.x * WAVES + wave;
    if (r >= st.R) return;                                  // uniform over the waves that meet at a barrier
    tsp_step_row<WAVES, UPDATE, SPLIT, STREAM, false, SQ>(st, selected, t, knn, idx_out, tile_out, lds_per_wave, mom_off, link,
                                                          lds, r, wave, sq);
}

// ---------------------------------------------------------------- dense state (API parity)
template <class SQ>
__global__ void tsp_state_data_kernel(dgl_tsp_state_t st, int t, float *__restrict__ out, SQ sq) {
    const int r = blockIdx.y;
    const int N = st.N;
    TspRowCtx cx;
    cx.xyb = st.xy + (size_t)(r / st.rows_per_instance) * N * 2;
    cx.rowN = (size_t)r * N;
    const int cur = st.tour[cx.rowN + (t - 1)], start = st.tour[cx.rowN];
    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.nc = dgl_sqnorm_t<SQ::DIRECT>(cx.xc, cx.yc);
    cx.xs = cx.xyb[2 * start]; cx.ys = cx.xyb[2 * start + 1]; cx.ns = dgl_sqnorm_t<SQ::DIRECT>(cx.xs, cx.ys);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        float f[TSP_F];
        tsp_features(sq, st, cx, j, f);
        float *o = out + (cx.rowN + j) * TSP_F;
#pragma unroll
        for (int c = 0; c < TSP_F; ++c) o[c] = f[c];
    }
}

// --------------------------------------------------------------------- tour length
// one wave per row; segment = sqrt(fl(dx^2) + fl(dy^2)) as TSP/TSPEnv.py:559-560, correctly rounded roots, the sum in
// binary64 rounded once
__global__ __launch_bounds__(256) void tsp_length_kernel(dgl_tsp_state_t st, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= st.R) return;
    const int N = st.N;
    const float *xyb = st.xy + (size_t)(r / st.rows_per_instance) * N * 2;
    const int32_t *tour = st.tour + (size_t)r * N;
    double s = 0.0;
    for (int p = lane; p < N; p += 64) {
        int a = tour[p], c = tour[p + 1 == N ? 0 : p + 1];
        float dx = __fsub_rn(xyb[2 * a], xyb[2 * c]), dy = __fsub_rn(xyb[2 * a + 1], xyb[2 * c + 1]);
        s += (double)sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
    s = wave_sum_f64(s);
    if (lane == 0) out[r] = (float)s;
}

// The same length as the reference's torch build returns it: every segment's root through the reference-bits model (a
// .sqrt() over a contiguous tensor, whatever the instance's size), summed over the tour in torch.sum's order
// (dgl_torch_row_sum).  Half a wave per row; rows past the end repeat the last one and store nothing.
__global__ __launch_bounds__(256) void tsp_length_torch_kernel(dgl_tsp_state_t st, float *__restrict__ out, DglSqrtMkl sq) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), r = min(row, st.R - 1);
    const int N = st.N;
    const float *xyb = st.xy + (size_t)(r / st.rows_per_instance) * N * 2;
    const int32_t *tour = st.tour + (size_t)r * N;
    const DglSqrtMkl::Lanes ln = sq.lanes();
    const float total = dgl_torch_row_sum(N, [&](int p, auto all_lanes) {
        const int a = min(max(tour[p], 0), N - 1), c = min(max(tour[p + 1 == N ? 0 : p + 1], 0), N - 1);
        const float dx = __fsub_rn(xyb[2 * a], xyb[2 * c]), dy = __fsub_rn(xyb[2 * a + 1], xyb[2 * c + 1]);
        const float r2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        return decltype(all_lanes)::value ? sq.all(ln, r2) : sq(r2);
    });
    if (row < st.R && (threadIdx.x & 31) == 0) out[row] = total;
}

// ------------------------------------------------------------------ beam permutation
// dst row (g, b) <- src row (g, prev_k[g][b]); copies avg, sd, visited and the first t tour entries.
__global__ __launch_bounds__(256) void tsp_permute_kernel(dgl_tsp_state_t src, dgl_tsp_state_t dst,
                                                         const int32_t *__restrict__ prev_k, int beam, int t) {
    const int r = blockIdx.x;
    const int g = r / beam;
    const int sr = g * beam + min(max(prev_k[r], 0), beam - 1);
    const int N = src.N, W = (N + 63) / 64;
    const size_t so = (size_t)sr * N, d0 = (size_t)r * N;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        dst.avg[d0 + j] = src.avg[so + j];
        dst.sd[d0 + j] = src.sd[so + j];
        if (j < t) dst.tour[d0 + j] = src.tour[so + j];
    }
    for (int w = threadIdx.x; w < W; w += blockDim.x) dst.visited[(size_t)r * W + w] = src.visited[(size_t)sr * W + w];
}

// ------------------------------------------------------------------ square-root model check
// whole waves take the model's wave-wide root (common.h), the ragged tail the one that works under any lane mask
template <class SQ>
__global__ void sqrt_apply_kernel(const float *__restrict__ x, float *__restrict__ out, long long n, SQ sq) {
    const typename SQ::Lanes ln = sq.lanes();
    for (long long w0 = blockIdx.x * (long long)blockDim.x + (threadIdx.x & ~63u); w0 < n; w0 += (long long)gridDim.x * blockDim.x) {
        const long long i = w0 + (threadIdx.x & 63u);
        if (w0 + 64 <= n)
            out[i] = sq.all(ln, x[i]);
        else if (i < n)
            out[i] = sq(x[i]);
    }
}

extern "C" int dgl_sqrt_apply(const float *x, float *out, int64_t n, dgl_stream_t stream) {
    DGL_CHECK_ARG(x != nullptr && out != nullptr && n >= 0, "NULL argument or n < 0");
    if (n == 0) return DGL_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dgl_sqrt_reference_bits())
        hipLaunchKernelGGL(sqrt_apply_kernel<DglSqrtMkl>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n, DglSqrtMkl{});
    else
        hipLaunchKernelGGL(sqrt_apply_kernel<DglSqrtIeee>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n, DglSqrtIeee{});
    DGL_CHECK_LAUNCH("dgl_sqrt_apply");
    return DGL_OK;
}

// ------------------------------------------------------------------------ launchers
static int check_tsp_state(const dgl_tsp_state_t *st) {
    DGL_CHECK_ARG(st != nullptr, "state is NULL");
    DGL_CHECK_ARG(st->B > 0 && st->N > 1 && st->rows_per_instance > 0, "bad sizes B=%d N=%d rows_per_instance=%d",
                  st->B, st->N, st->rows_per_instance);
    DGL_CHECK_ARG(st->R == st->B * st->rows_per_instance, "R=%d != B*rows_per_instance", st->R);
    DGL_CHECK_ARG(st->xy && st->avg && st->sd && st->visited && st->tour, "NULL state buffer");
    return DGL_OK;
}

extern "C" int dgl_tsp_reset(const dgl_tsp_state_t *st, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    size_t lds = (size_t)st->N * 2 * sizeof(float);
    DGL_CHECK_ARG(lds <= 150 * 1024, "N=%d too large for the LDS-staged reset (max 19200)", st->N);
    if (lds > 48 * 1024)
        dgl_max_dynamic_lds((const void *)tsp_reset_kernel, (int)lds);
    if (dgl_sqrt_reference_bits()) {
        if (lds > 48 * 1024)
            dgl_max_dynamic_lds((const void *)tsp_reset_torch_kernel, (int)lds);
        hipLaunchKernelGGL(tsp_reset_torch_kernel, dim3((st->N + 7) / 8, st->B), dim3(256), lds, (hipStream_t)stream, *st,
                           DglSqrtMkl{});
        DGL_CHECK_LAUNCH("dgl_tsp_reset");
        return DGL_OK;
    }
    dim3 grid((st->N + 255) / 256, st->B);
    hipLaunchKernelGGL(tsp_reset_kernel, grid, dim3(256), lds, (hipStream_t)stream, *st);
    DGL_CHECK_LAUNCH("dgl_tsp_reset");
    return DGL_OK;
}

// Instances of at most 25 points (torch.cdist's direct evaluation, common.h): one wave per row, one row per workgroup.
template <bool UPDATE, class SQ>
static int launch_tsp_step_small(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                                 float *tile_out, dgl_stream_t stream, const char *what, const TspBeamLink &link, SQ sq) {
    static_assert(SQ::DIRECT, "the small-instance shape is the direct evaluation's");
    const int per_wave = st->N + (knn + 1) * DGL_TILE_STRIDE + 128;
    const size_t lds1 = (size_t)per_wave * sizeof(float);
    DGL_CHECK_ARG(lds1 <= 150 * 1024, "N=%d, knn=%d need %zu bytes of LDS per row (max 153600)", st->N, knn, lds1);
    if (lds1 > 48 * 1024)
        dgl_max_dynamic_lds((const void *)tsp_step_kernel<SQ, 1, UPDATE>, (int)lds1);
    hipLaunchKernelGGL((tsp_step_kernel<SQ, 1, UPDATE>), dim3(st->R), dim3(64), lds1, (hipStream_t)stream, *st,
                       selected, t, knn, idx_out, tile_out, per_wave, 0, link, sq);
    DGL_CHECK_LAUNCH(what);
    return DGL_OK;
}

template <bool UPDATE, class SQ>
static int launch_tsp_step_sq(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                              float *tile_out, dgl_stream_t stream, const char *what, const TspBeamLink &link, SQ sq) {
    const int per_wave = st->N + (knn + 1) * DGL_TILE_STRIDE + 128;
    size_t lds4 = (size_t)per_wave * 4 * sizeof(float);
    // shared-row shape, rows of up to 2048 nodes: 8 N more bytes of LDS hold the new (avg, sd) pairs
    static const int force_mom = getenv("DGL_STEP_MOM") ? atoi(getenv("DGL_STEP_MOM")) : -1;                // experiments
    const bool with_mom = UPDATE && (force_mom >= 0 ? force_mom != 0 : st->N <= 2048);
    // long rows, few of them: eight waves share a row (more loads in flight per CU: the scan of a 10 000-node row
    // is bandwidth-bound, and four waves per row leave a CU with eight waves at 512 rows)
    static const int force_waves = getenv("DGL_STEP_WAVES") ? atoi(getenv("DGL_STEP_WAVES")) : 0;            // experiments
    // (measured: 512 x 10 000 nodes 27 -> 25 us, 64 x 18 512 nodes 29 -> 20 us; no gain at 2 500 nodes, 16 waves lose)
    const int waves = force_waves == 4 || force_waves == 8 ? force_waves : (st->N >= 4096 && st->R <= 1024 ? 8 : 4);
    const int split_floats = per_wave + waves * (64 + 128) + waves;
    const int mom_off = with_mom ? (split_floats + 1) & ~1 : 0;
    const size_t lds_split = (size_t)(with_mom ? mom_off + 2 * st->N : split_floats) * sizeof(float);
    static const int force_split = getenv("DGL_STEP_SPLIT") ? atoi(getenv("DGL_STEP_SPLIT")) : -1;   // experiments
    // measured at N = 1000: sharing a row wins below 2048 rows (latency) and from 8192 rows on (6-10 %)
    const bool split = force_split >= 0 ? force_split != 0 : ((st->R < 2048 || st->R >= 8192) && st->N >= 512);
    if (split && lds_split <= 150 * 1024) {
        // few rows: four waves share a row
        if (lds_split > 48 * 1024)
            dgl_max_dynamic_lds((const void *)tsp_step_kernel<SQ, 4, UPDATE, true>, (int)lds_split);
        static const int force_stream = getenv("DGL_STEP_STREAM") ? atoi(getenv("DGL_STEP_STREAM")) : -1;          // experiments
        if (waves == 8) {
            if (lds_split > 48 * 1024)
                dgl_max_dynamic_lds((const void *)tsp_step_kernel<SQ, 8, UPDATE, true>, (int)lds_split);
            hipLaunchKernelGGL((tsp_step_kernel<SQ, 8, UPDATE, true>), dim3(st->R), dim3(512), lds_split, (hipStream_t)stream,
                               *st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, link, sq);
        } else if (with_mom && lds_split <= 48 * 1024 && (force_stream >= 0 ? force_stream != 0 : (long long)st->R * st->N >= 16000000ll))
            hipLaunchKernelGGL((tsp_step_kernel<SQ, 4, UPDATE, true, true>), dim3(st->R), dim3(256), lds_split,
                               (hipStream_t)stream, *st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, link, sq);
        else
            hipLaunchKernelGGL((tsp_step_kernel<SQ, 4, UPDATE, true>), dim3(st->R), dim3(256), lds_split, (hipStream_t)stream,
                               *st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, link, sq);
    } else if (lds4 <= 64 * 1024 && st->R >= 2048) {
        hipLaunchKernelGGL((tsp_step_kernel<SQ, 4, UPDATE>), dim3((st->R + 3) / 4), dim3(256), lds4, (hipStream_t)stream,
                           *st, selected, t, knn, idx_out, tile_out, per_wave, 0, link, sq);
    } else {
        size_t lds1 = (size_t)per_wave * sizeof(float);
        DGL_CHECK_ARG(lds1 <= 150 * 1024, "N=%d, knn=%d need %zu bytes of LDS per row (max 153600)", st->N, knn, lds1);
        if (lds1 > 48 * 1024)
            dgl_max_dynamic_lds((const void *)tsp_step_kernel<SQ, 1, UPDATE>, (int)lds1);
        hipLaunchKernelGGL((tsp_step_kernel<SQ, 1, UPDATE>), dim3(st->R), dim3(64), lds1, (hipStream_t)stream, *st,
                           selected, t, knn, idx_out, tile_out, per_wave, 0, link, sq);
    }
    DGL_CHECK_LAUNCH(what);
    return DGL_OK;
}

// the square-root model in force on the current device picks the instantiation (dgl_set_sqrt_model)
template <bool UPDATE>
static int launch_tsp_step(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                           float *tile_out, dgl_stream_t stream, const char *what, const TspBeamLink &link = TspBeamLink{}) {
    if (st->N <= DGL_DIRECT_MAX_POINTS) {
        if (dgl_sqrt_reference_bits())
            return launch_tsp_step_small<UPDATE>(st, selected, t, knn, idx_out, tile_out, stream, what, link, DglSqrtMklT<true>{});
        return launch_tsp_step_small<UPDATE>(st, selected, t, knn, idx_out, tile_out, stream, what, link, DglSqrtIeeeT<true>{});
    }
    if (dgl_sqrt_reference_bits())
        return launch_tsp_step_sq<UPDATE>(st, selected, t, knn, idx_out, tile_out, stream, what, link, DglSqrtMkl{});
    return launch_tsp_step_sq<UPDATE>(st, selected, t, knn, idx_out, tile_out, stream, what, link, DglSqrtIeee{});
}

extern "C" int dgl_tsp_step(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn,
                            int32_t *idx_out, float *tile_out, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(selected != nullptr, "selected is NULL");
    DGL_CHECK_ARG(t >= 1 && t <= st->N, "t=%d outside [1,N=%d]", t, st->N);
    DGL_CHECK_ARG((idx_out == nullptr) == (tile_out == nullptr), "idx_out and tile_out must both be given or both be NULL");
    DGL_CHECK_ARG(idx_out == nullptr || knn >= 1, "knn=%d must be >= 1", knn);
    if (idx_out == nullptr) knn = 1;
    return launch_tsp_step<true>(st, selected, t, knn, idx_out, tile_out, stream, "dgl_tsp_step");
}

extern "C" int dgl_tsp_step_beam(const dgl_tsp_state_t *src, const dgl_tsp_state_t *dst, const int32_t *prev_k,
                                 int32_t beam, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                                 float *tile_out, int32_t *hist_node, int32_t *hist_parent, dgl_stream_t stream) {
    if (int e = check_tsp_state(src)) return e;
    if (int e = check_tsp_state(dst)) return e;
    DGL_CHECK_ARG(selected != nullptr && prev_k != nullptr && hist_node != nullptr && hist_parent != nullptr, "NULL argument");
    DGL_CHECK_ARG(beam >= 1 && src->R == dst->R && src->N == dst->N && src->R % beam == 0 && src->xy == dst->xy,
                  "src/dst shapes differ or R %% beam != 0");
    DGL_CHECK_ARG(src->avg != dst->avg && src->sd != dst->sd && src->visited != dst->visited && src->tour != dst->tour,
                  "src and dst must be different buffers");
    DGL_CHECK_ARG(t >= 2 && t <= dst->N, "t=%d outside [2,N=%d] (the first step has no back-pointer)", t, dst->N);
    DGL_CHECK_ARG((idx_out == nullptr) == (tile_out == nullptr), "idx_out and tile_out must both be given or both be NULL");
    DGL_CHECK_ARG(idx_out == nullptr || knn >= 1, "knn=%d must be >= 1", knn);
    if (idx_out == nullptr) knn = 1;
    const TspBeamLink link{src->avg, src->sd, src->visited, src->tour, prev_k, beam, hist_node, hist_parent};
    return launch_tsp_step<true>(dst, selected, t, knn, idx_out, tile_out, stream, "dgl_tsp_step_beam", link);
}

// tour[r][t - 1] for t = t_last .. t_first from the history of linked steps: one thread walks one row's back-pointers
__global__ void tsp_backtrack_kernel(dgl_tsp_state_t st, const int32_t *__restrict__ hist_node,
                                     const int32_t *__restrict__ hist_parent, int t_first, int t_last) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= st.R) return;
    int p = r;
    for (int t = t_last; t >= t_first; --t) {
        const size_t e = (size_t)(t - 1) * st.R + p;
        st.tour[(size_t)r * st.N + (t - 1)] = hist_node[e];
        p = min(max(hist_parent[e], 0), st.R - 1);
    }
}

extern "C" int dgl_tsp_beam_backtrack(const dgl_tsp_state_t *st, const int32_t *hist_node, const int32_t *hist_parent,
                                      int32_t t_first, int32_t t_last, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(hist_node != nullptr && hist_parent != nullptr, "NULL history");
    DGL_CHECK_ARG(t_first >= 2 && t_last <= st->N && t_first <= t_last, "steps [%d,%d] outside [2,N=%d]", t_first, t_last, st->N);
    hipLaunchKernelGGL(tsp_backtrack_kernel, dim3((st->R + 63) / 64), dim3(64), 0, (hipStream_t)stream, *st, hist_node,
                       hist_parent, t_first, t_last);
    DGL_CHECK_LAUNCH("dgl_tsp_beam_backtrack");
    return DGL_OK;
}

extern "C" int dgl_tsp_knn(const dgl_tsp_state_t *st, int32_t t, int32_t knn, int32_t *idx_out, float *tile_out,
                           dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(t >= 1 && t < st->N, "t=%d outside [1,N-1=%d]", t, st->N - 1);
    DGL_CHECK_ARG(idx_out != nullptr && tile_out != nullptr && knn >= 1, "idx_out/tile_out NULL or knn=%d < 1", knn);
    return launch_tsp_step<false>(st, nullptr, t, knn, idx_out, tile_out, stream, "dgl_tsp_knn");
}

extern "C" int dgl_tsp_state_data(const dgl_tsp_state_t *st, int32_t t, float *data_out, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(data_out != nullptr, "data_out is NULL");
    DGL_CHECK_ARG(t >= 1 && t <= st->N, "t=%d outside [1,N=%d]", t, st->N);
    dim3 grid(min((st->N + 255) / 256, 64), st->R);
    const bool tab = dgl_sqrt_reference_bits();
    if (st->N <= DGL_DIRECT_MAX_POINTS && tab)
        hipLaunchKernelGGL(tsp_state_data_kernel<DglSqrtMklT<true>>, grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, DglSqrtMklT<true>{});
    else if (st->N <= DGL_DIRECT_MAX_POINTS)
        hipLaunchKernelGGL(tsp_state_data_kernel<DglSqrtIeeeT<true>>, grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, DglSqrtIeeeT<true>{});
    else if (tab)
        hipLaunchKernelGGL(tsp_state_data_kernel<DglSqrtMkl>, grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, DglSqrtMkl{});
    else
        hipLaunchKernelGGL(tsp_state_data_kernel<DglSqrtIeee>, grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, DglSqrtIeee{});
    DGL_CHECK_LAUNCH("dgl_tsp_state_data");
    return DGL_OK;
}

extern "C" int dgl_tsp_tour_length(const dgl_tsp_state_t *st, float *length_out, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(length_out != nullptr, "length_out is NULL");
    if (dgl_sqrt_reference_bits())
        hipLaunchKernelGGL(tsp_length_torch_kernel, dim3((st->R + 7) / 8), dim3(256), 0, (hipStream_t)stream, *st, length_out, DglSqrtMkl{});
    else
        hipLaunchKernelGGL(tsp_length_kernel, dim3((st->R + 3) / 4), dim3(256), 0, (hipStream_t)stream, *st, length_out);
    DGL_CHECK_LAUNCH("dgl_tsp_tour_length");
    return DGL_OK;
}

extern "C" int dgl_tsp_permute(const dgl_tsp_state_t *src, const dgl_tsp_state_t *dst, const int32_t *prev_k,
                               int32_t beam, int32_t t, dgl_stream_t stream) {
    if (int e = check_tsp_state(src)) return e;
    if (int e = check_tsp_state(dst)) return e;
    DGL_CHECK_ARG(prev_k != nullptr && beam >= 1, "bad prev_k/beam");
    DGL_CHECK_ARG(src->R == dst->R && src->N == dst->N && src->R % beam == 0, "src/dst shapes differ or R %% beam != 0");
    DGL_CHECK_ARG(src->avg != dst->avg && src->sd != dst->sd && src->tour != dst->tour && src->visited != dst->visited,
                  "src and dst must be distinct buffers");
    hipLaunchKernelGGL(tsp_permute_kernel, dim3(src->R), dim3(256), 0, (hipStream_t)stream, *src, *dst, prev_k, beam, t);
    DGL_CHECK_LAUNCH("dgl_tsp_permute");
    return DGL_OK;
}
