// TSP environment state on the device: reset, fused step + neighbourhood + token tile,
// dense state materialisation, tour length, beam permutation.
//
// Layout: one wave64 owns one row (one partial tour).  Coordinates are read from the
// instance ([B][N][2], shared by the rows of an instance), the running features avg/sd
// are [R][N] binary32, the visited set is a bitmask.  No [N][N] distance matrix and no
// [R][N][7] feature tensor exist: distances are recomputed with the reference's exact
// arithmetic wherever they are needed.
#include <cstdlib>

#include "common.h"
#include "select.h"
#include "state_common.h"

#define TSP_F 7

// --------------------------------------------------------------------------- reset
// grid: (ceil(N / 256), B), block 256 (see dgl_reset_rows in state_common.h)
__global__ __launch_bounds__(256) void tsp_reset_kernel(dgl_tsp_state_t st) {
    dgl_reset_rows(st.xy, nullptr, st.N, st.rows_per_instance, st.avg, st.sd, st.visited);
}

// ---------------------------------------------------------------------------- step
// One wave per row.  LDS per wave: keys[N] + tile[(knn+1)*STRIDE] + scratch[128].
struct TspRowCtx {
    const float *xyb;
    float xc, yc, nc, xs, ys, ns;
    size_t rowN;
};

// raw feature row of a node at p with running statistics (a, sd): x, y, d(j,cur), avg, sd and
// sin/cos of the direction towards the start node (TSP/TSPEnv.py:371,393,415,430-475)
__device__ __forceinline__ void tsp_row(const TspRowCtx &cx, float2 p, float a, float sd, float *f) {
    const float nj = dgl_sqnorm(p.x, p.y);
    const float ds = dgl_chain_dist(p.x, p.y, nj, cx.xs, cx.ys, cx.ns);       // d(j, start)
    const float den = __fadd_rn(ds, 1e-7f);
    f[0] = p.x;
    f[1] = p.y;
    f[2] = dgl_chain_dist(p.x, p.y, nj, cx.xc, cx.yc, cx.nc);                  // d(j, current)
    f[3] = a;
    f[4] = sd;
    f[5] = __fdiv_rn(__fsub_rn(cx.ys, p.y), den);
    f[6] = __fdiv_rn(__fsub_rn(cx.xs, p.x), den);
}

__device__ __forceinline__ void tsp_features(const dgl_tsp_state_t &st, const TspRowCtx &cx, int j, float *f) {
    tsp_row(cx, reinterpret_cast<const float2 *>(cx.xyb)[j], st.avg[cx.rowN + j], st.sd[cx.rowN + j], f);
}

#define STEP_UNROLL 4     // node chunks (of 64) whose loads are issued together: bytes in flight per wave

// SPLIT: the WAVES waves of a workgroup share ONE row: each scans a contiguous quarter of the nodes
// (state update + keys into the shared LDS array), then wave 0 alone selects and writes the tile.
// With few rows this is what hides the memory latency of the scan (one wave per row leaves most
// SIMDs with a single wave and every load exposed).  LDS: the row's arrays + 64 * WAVES lane minima + per-wave
// candidate lists (long rows: the waves share the compaction too) + optionally the row's new (avg, sd) pairs.
// Beam search: row r of the state being written continues row (r / beam) * beam + prev_k[r] of another copy of the
// state (its back-pointer).  The step then reads avg / sd / visited of that row and writes row r, which replaces
// the separate gather of the whole state; the chosen node and the parent row go to a history from which the tours
// are rebuilt once at the end (tsp_backtrack_kernel).  prev_k == NULL: the ordinary in-place step.
struct TspBeamLink {
    const float *avg_in, *sd_in;
    const uint64_t *vis_in;
    const int32_t *tour_in;
    const int32_t *prev_k;
    int beam;
    int32_t *hist_node, *hist_parent;      // [N][R], entry (t - 1) * R + r
};

// STREAM: the state arrays are read and written with non-temporal hints.  With tens of thousands of rows the rows in
// flight exceed the L2; streaming the arrays that are read once and written once past it keeps the coordinates resident
// for the candidates' feature rows.  Measured: it pays from about 16 million row-nodes on (16 384 x 1000: 10 % less fetch
// traffic, 4 % faster; 24 576 x 1000 and 16 384 x 2000: 10-11 % faster) and costs 7-25 % with fewer (the launcher's rule).
template <int WAVES, bool UPDATE, bool SPLIT = false, bool STREAM = false>
__global__ __launch_bounds__(WAVES * 64) void tsp_step_kernel(dgl_tsp_state_t st, const int32_t *__restrict__ selected,
                                                             int t, int knn, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ tile_out, int lds_per_wave, int mom_off,
                                                             TspBeamLink link) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar registers
    const int r = SPLIT ? blockIdx.x : blockIdx.x * WAVES + wave;
    if (r >= st.R) return;                                  // uniform over the waves that meet at a barrier below
    const int N = st.N, W = (N + 63) / 64;
    float *keys = lds + (SPLIT ? 0 : (size_t)wave * lds_per_wave);
    float *tile = keys + N;
    float *scratch = tile + (knn + 1) * DGL_TILE_STRIDE;
    // mom_off != 0 (UPDATE only): the row's new (avg, sd) pairs are also kept in LDS, so that the feature rows of the
    // candidates do not fetch them back from memory (at saturating batches half of those lines have left the L2)
    float2 *mom = (UPDATE && mom_off) ? reinterpret_cast<float2 *>(keys + mom_off) : nullptr;
    // this wave's node range [j_begin, j_end): whole groups of STEP_UNROLL chunks
    int j_begin = 0, j_end = N;
    if (SPLIT) {
        const int per = ((W + WAVES - 1) / WAVES + STEP_UNROLL - 1) / STEP_UNROLL * STEP_UNROLL;
        j_begin = wave * per * 64;
        j_end = min(N, j_begin + per * 64);
    }

    const int b = r / st.rows_per_instance;
    TspRowCtx cx;
    cx.xyb = st.xy + (size_t)b * N * 2;
    cx.rowN = (size_t)r * N;
    const int u = N - t;                                    // unvisited after this step
    const float up1 = (float)(u + 1);
    const DglConstDiv du = dgl_const_div((float)u);
    const bool want_knn = (idx_out != nullptr) && (t < N);
    float lane_min = INFINITY;
    uint64_t *vis = st.visited + (size_t)r * W;
    const float2 *xy2 = reinterpret_cast<const float2 *>(cx.xyb);
    float *avg = st.avg + cx.rowN, *sd = st.sd + cx.rowN;
    const bool linked = UPDATE && link.prev_k != nullptr;
    int sr = r;                                             // the row this one continues
    if (linked) sr = (r / link.beam) * link.beam + min(max(__builtin_amdgcn_readfirstlane(link.prev_k[r]), 0), link.beam - 1);
    const float *avg_in = linked ? link.avg_in + (size_t)sr * N : avg, *sd_in = linked ? link.sd_in + (size_t)sr * N : sd;
    const uint64_t *vis_in = linked ? link.vis_in + (size_t)sr * W : vis;
    const int32_t *tour_in = linked ? link.tour_in + (size_t)sr * N : st.tour + cx.rowN;
    // The loads of a group of STEP_UNROLL chunks are issued one group ahead of its arithmetic, so
    // every wave keeps a group of requests in flight while it computes (clamped addresses keep the
    // loads unconditional).
    struct Group {
        float2 p[STEP_UNROLL];
        float a0[STEP_UNROLL], s0[STEP_UNROLL];
    };
    auto request = [&](int j0, Group &g) {
#pragma unroll
        for (int q = 0; q < STEP_UNROLL; ++q) {
            const int j = min(j0 + 64 * q + lane, N - 1);
            g.p[q] = xy2[j];
            if (UPDATE) {
                g.a0[q] = STREAM ? __builtin_nontemporal_load(avg_in + j) : avg_in[j];
                g.s0[q] = STREAM ? __builtin_nontemporal_load(sd_in + j) : sd_in[j];
            }
        }
    };
    Group nxt;
    request(j_begin, nxt);
    // The current node and its coordinates are two dependent round trips; they start only now, behind the first
    // group's loads, which do not depend on them.  Wave-uniform values: kept in scalar registers.
    // (clamped: a node id outside [0, N) from a faulty caller must not turn into a wild address)
    const int cur = min(max(__builtin_amdgcn_readfirstlane(UPDATE ? selected[r] : st.tour[cx.rowN + (t - 1)]), 0), N - 1);
    const int start = (t == 1) ? cur : __builtin_amdgcn_readfirstlane(tour_in[0]);
    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.nc = dgl_sqnorm(cx.xc, cx.yc);
    cx.xs = cx.xyb[2 * start]; cx.ys = cx.xyb[2 * start + 1]; cx.ns = dgl_sqnorm(cx.xs, cx.ys);
    // visited words of 64 consecutive chunks, one per lane; the word of a chunk is read back as a
    // wave-uniform lane mask (bit = lane) with the excluded nodes of that chunk added
    unsigned long long vis_words = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += 64 * STEP_UNROLL) {
        const Group g = nxt;
        if (j0 + 64 * STEP_UNROLL < j_end) request(j0 + 64 * STEP_UNROLL, nxt);
        if (want_knn && (j0 == j_begin || ((j0 >> 6) & 63) == 0)) vis_words = vis_in[min(((j0 >> 6) & ~63) + lane, W - 1)];
#pragma unroll
        for (int q = 0; q < STEP_UNROLL; ++q) {
            const int j = j0 + 64 * q + lane;
            if (j < j_end) {
                const float x = g.p[q].x, y = g.p[q].y, nj = dgl_sqnorm(x, y);
                const float cross = dgl_chain_cross(cx.xc, cx.yc, x, y);     // shared by dis[j][cur] and dis[cur][j]
                float a1 = 0.0f, s1 = 0.0f;
                if (UPDATE) {
                    const float dcol = dgl_sqrt_pk(dgl_chain_sq(cross, nj, cx.nc));       // dis[j][cur]
                    dgl_update_moments<false>(dcol, g.a0[q], g.s0[q], up1, du, &a1, &s1);   // TSP/TSPEnv.py:386, 408-409
                    if (STREAM) {
                        __builtin_nontemporal_store(a1, avg + j);
                        __builtin_nontemporal_store(s1, sd + j);
                    } else {
                        avg[j] = a1;
                        sd[j] = s1;
                    }
                }
                if (want_knn) {
                    const int chunk = (j0 >> 6) + q;
                    unsigned long long excl = dgl_readlane_u64(vis_words, chunk & 63);
                    if (chunk == (cur >> 6)) excl |= 1ull << (cur & 63);
                    // key = dis[cur][j]^2; the root is taken for the few candidates only.  The reference
                    // adds 1e2 to visited keys; they can never rank among the k <= #unvisited smallest,
                    // so they are simply excluded here
                    const float key = dgl_inf_where(excl, dgl_chain_sq(cross, cx.nc, nj));
                    keys[j] = key;
                    lane_min = fminf(lane_min, key);
                    if (UPDATE && mom) mom[j] = make_float2(a1, s1);
                }
            }
        }
    }
    // (a wave that still scans may read the visited word with this bit already set: it excludes `cur` anyway)
    if (UPDATE && linked && (!SPLIT || wave == 0)) {
        for (int w = lane; w < W; w += 64) vis[w] = vis_in[w] | (w == (cur >> 6) ? 1ull << (cur & 63) : 0ull);
        if (lane == 0) {
            link.hist_node[(size_t)(t - 1) * st.R + r] = cur;
            link.hist_parent[(size_t)(t - 1) * st.R + r] = sr;
            st.tour[cx.rowN] = start;                       // every row of a beam group shares its start node
        }
    } else if (UPDATE && lane == 0 && (!SPLIT || wave == 0)) {
        st.tour[cx.rowN + (t - 1)] = cur;
        vis[cur >> 6] |= 1ull << (cur & 63);
    }
    if (!want_knn) return;

    // LDS keys of all lanes, and (for the staging loads below) the avg/sd stores of all lanes
    const int k = min(knn, u);
    int32_t *idx_row = idx_out + (size_t)r * knn;
    float ckey;
    int cj;
    int c;
    if (SPLIT) {
        // LDS behind the row's arrays: lane minima [64 WAVES] | candidate lists [WAVES][128] | counts [WAVES]
        float *mins = scratch + 128;
        float *lists = mins + 64 * WAVES;
        int *counts = reinterpret_cast<int *>(lists + 128 * WAVES);
        mins[threadIdx.x] = lane_min;
        __syncthreads();                                    // keys, minima and the avg/sd stores of every wave
        // the 64 classes {lane of any wave} partition the keys as the lanes of one wave do
#pragma unroll
        for (int w = 0; w < WAVES; ++w) lane_min = fminf(lane_min, mins[64 * w + lane]);
        // Long rows: every wave derives the same bound and compacts the candidates of the range it scanned (on
        // long rows the compaction is the longest part of the selection); short rows: wave 0 does it alone and the
        // others retire at once, which is what keeps saturating batches moving.
        const bool share = N >= 2048 && k <= 64;
        if (!share) {
            if (wave != 0) return;
            c = wave_candidates(keys, N, k, lane_min, scratch, &ckey, &cj);
        } else {
            const float bound = __uint_as_float(wave_candidate_bound(lane_min, k));
            const int cw = wave_compact(keys, j_begin, j_end, bound, lists + 128 * wave,
                                        reinterpret_cast<int *>(lists + 128 * wave) + 64);
            if (lane == 0) counts[wave] = cw;
            __syncthreads();                                // `share` is uniform over the workgroup
            if (wave != 0) return;
            c = wave_candidates_from_lists(lists, counts, WAVES, keys, N, k, bound, scratch, &ckey, &cj);
        }
    } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        c = wave_candidates(keys, N, k, lane_min, scratch, &ckey, &cj);
    }
    if (c >= 0) {
        // The inputs of every candidate's feature row (and of the anchor token = the current node, on
        // lane 63 when that lane holds no candidate) are requested before the ranking bisection.
        const bool anchor_early = (c < 64) && (lane == 63);
        const int jl = anchor_early ? cur : cj;
        const float2 p = xy2[jl];
        float av, sv;
        if (mom) { const float2 m = mom[jl]; av = m.x; sv = m.y; } else { av = avg[jl]; sv = sd[jl]; }
        int rank;
        const bool pick = wave_pick_k(dgl_sqrt(ckey), c, k, &rank);
        if (pick) idx_row[rank] = cj;
        if (pick || anchor_early) tsp_row(cx, p, av, sv, tile + (pick ? rank : k) * DGL_TILE_STRIDE);
        if (c == 64 && lane == 63) tsp_features(st, cx, cur, tile + k * DGL_TILE_STRIDE);
    } else {
        wave_select_general(keys, N, k, scratch, [&](int j, int rank) {
            idx_row[rank] = j;
            tsp_features(st, cx, j, tile + rank * DGL_TILE_STRIDE);
        });
        if (lane == 63) tsp_features(st, cx, cur, tile + k * DGL_TILE_STRIDE);
    }
    for (int e = k + lane; e < knn; e += 64) idx_row[e] = -1;
    wave_lds_sync();
    wave_minmax_norm_store(tile, k + 1, TSP_F, 0.0f, tile_out + (size_t)r * (knn + 1) * DGL_TILE_STRIDE, scratch);
}

// ---------------------------------------------------------------- dense state (API parity)
__global__ void tsp_state_data_kernel(dgl_tsp_state_t st, int t, float *__restrict__ out) {
    const int r = blockIdx.y;
    const int N = st.N;
    TspRowCtx cx;
    cx.xyb = st.xy + (size_t)(r / st.rows_per_instance) * N * 2;
    cx.rowN = (size_t)r * N;
    const int cur = st.tour[cx.rowN + (t - 1)], start = st.tour[cx.rowN];
    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.nc = dgl_sqnorm(cx.xc, cx.yc);
    cx.xs = cx.xyb[2 * start]; cx.ys = cx.xyb[2 * start + 1]; cx.ns = dgl_sqnorm(cx.xs, cx.ys);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        float f[TSP_F];
        tsp_features(st, cx, j, f);
        float *o = out + (cx.rowN + j) * TSP_F;
#pragma unroll
        for (int c = 0; c < TSP_F; ++c) o[c] = f[c];
    }
}

// --------------------------------------------------------------------- tour length
// one wave per row; segment = sqrt(fl(dx^2) + fl(dy^2)) as TSP/TSPEnv.py:559-560
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

// ------------------------------------------------------------------------ launchers
static int check_tsp_state(const dgl_tsp_state_t *st) {
    DGL_CHECK_ARG(st != nullptr, "state is NULL");
    DGL_CHECK_ARG(st->B > 0 && st->N > 1 && st->rows_per_instance > 0, "bad sizes B=%d N=%d rows_per_instance=%d",
                  st->B, st->N, st->rows_per_instance);
    DGL_CHECK_ARG(st->R == st->B * st->rows_per_instance, "R=%d != B*rows_per_instance", st->R);
    DGL_CHECK_ARG(st->xy && st->avg && st->sd && st->visited && st->tour, "NULL state buffer");
    // torch.cdist evaluates instances of at most 25 points with the direct formula sqrt(dx^2 + dy^2) instead of the
    // matrix-product chain implemented here: results would differ from the reference's, so they are refused
    if (st->N <= 25)
        return dgl_set_error(DGL_ERR_UNSUPPORTED, "N=%d: instances of at most 25 points are not supported (the reference's "
                             "distance arithmetic differs there)", st->N);
    return DGL_OK;
}

extern "C" int dgl_tsp_reset(const dgl_tsp_state_t *st, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    size_t lds = (size_t)st->N * 2 * sizeof(float);
    DGL_CHECK_ARG(lds <= 150 * 1024, "N=%d too large for the LDS-staged reset (max 19200)", st->N);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)tsp_reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((st->N + 255) / 256, st->B);
    hipLaunchKernelGGL(tsp_reset_kernel, grid, dim3(256), lds, (hipStream_t)stream, *st);
    DGL_CHECK_LAUNCH("dgl_tsp_reset");
    return DGL_OK;
}

template <bool UPDATE>
static int launch_tsp_step(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                           float *tile_out, dgl_stream_t stream, const char *what, const TspBeamLink &link = TspBeamLink{}) {
    const int per_wave = st->N + (knn + 1) * DGL_TILE_STRIDE + 128;
    size_t lds4 = (size_t)per_wave * 4 * sizeof(float);
    // shared-row shape, rows of up to 2048 nodes: 8 N more bytes of LDS hold the new (avg, sd) pairs
    static const int force_mom = getenv("DGL_STEP_MOM") ? atoi(getenv("DGL_STEP_MOM")) : -1;                // experiments
    const bool with_mom = UPDATE && (force_mom >= 0 ? force_mom != 0 : st->N <= 2048);
    const int split_floats = per_wave + 4 * (64 + 128) + 4;
    const int mom_off = with_mom ? (split_floats + 1) & ~1 : 0;
    const size_t lds_split = (size_t)(with_mom ? mom_off + 2 * st->N : split_floats) * sizeof(float);
    static const int force_split = getenv("DGL_STEP_SPLIT") ? atoi(getenv("DGL_STEP_SPLIT")) : -1;   // experiments
    // measured at N = 1000: sharing a row wins below 2048 rows (latency) and from 8192 rows on (6-10 %)
    const bool split = force_split >= 0 ? force_split != 0 : ((st->R < 2048 || st->R >= 8192) && st->N >= 512);
    if (split && lds_split <= 150 * 1024) {
        // few rows: four waves share a row
        if (lds_split > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)tsp_step_kernel<4, UPDATE, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split);
        static const int force_stream = getenv("DGL_STEP_STREAM") ? atoi(getenv("DGL_STEP_STREAM")) : -1;          // experiments
        if (with_mom && lds_split <= 48 * 1024 && (force_stream >= 0 ? force_stream != 0 : (long long)st->R * st->N >= 16000000ll))
            hipLaunchKernelGGL((tsp_step_kernel<4, UPDATE, true, true>), dim3(st->R), dim3(256), lds_split,
                               (hipStream_t)stream, *st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, link);
        else
            hipLaunchKernelGGL((tsp_step_kernel<4, UPDATE, true>), dim3(st->R), dim3(256), lds_split, (hipStream_t)stream,
                               *st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, link);
    } else if (lds4 <= 64 * 1024 && st->R >= 2048) {
        hipLaunchKernelGGL((tsp_step_kernel<4, UPDATE>), dim3((st->R + 3) / 4), dim3(256), lds4, (hipStream_t)stream,
                           *st, selected, t, knn, idx_out, tile_out, per_wave, 0, link);
    } else {
        size_t lds1 = (size_t)per_wave * sizeof(float);
        DGL_CHECK_ARG(lds1 <= 150 * 1024, "N=%d, knn=%d need %zu bytes of LDS per row (max 153600)", st->N, knn, lds1);
        if (lds1 > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)tsp_step_kernel<1, UPDATE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        hipLaunchKernelGGL((tsp_step_kernel<1, UPDATE>), dim3(st->R), dim3(64), lds1, (hipStream_t)stream, *st,
                           selected, t, knn, idx_out, tile_out, per_wave, 0, link);
    }
    DGL_CHECK_LAUNCH(what);
    return DGL_OK;
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
    hipLaunchKernelGGL(tsp_state_data_kernel, grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out);
    DGL_CHECK_LAUNCH("dgl_tsp_state_data");
    return DGL_OK;
}

extern "C" int dgl_tsp_tour_length(const dgl_tsp_state_t *st, float *length_out, dgl_stream_t stream) {
    if (int e = check_tsp_state(st)) return e;
    DGL_CHECK_ARG(length_out != nullptr, "length_out is NULL");
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
