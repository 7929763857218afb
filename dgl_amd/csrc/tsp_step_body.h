// Fused TSP step (state update + neighbourhood + token tile) as a device function; see tsp_state.hip.
#pragma once
#include "common.h"
#include "select.h"
#include "state_common.h"

#define TSP_F 7

// ---------------------------------------------------------------------------- step
// One wave per row.  LDS per wave: keys[N] + tile[(knn+1)*STRIDE] + scratch[128].
struct TspRowCtx {
    const float *xyb;
    float xc, yc, nc, xs, ys, ns;
    size_t rowN;
};

// raw feature row of a node at p with running statistics (a, sd): x, y, d(j,cur), avg, sd and
// sin/cos of the direction towards the start node (TSP/TSPEnv.py:371,393,415,430-475)
// ALL: called by all 64 lanes of the wave together (the model's wave-wide roots, common.h)
template <bool ALL = false, class SQ>
__device__ __forceinline__ void tsp_row(const SQ &sq, const TspRowCtx &cx, float2 p, float a, float sd, float *f,
                                        const typename SQ::Lanes &ln = typename SQ::Lanes{}) {
    const float nj = dgl_sqnorm_t<SQ::DIRECT>(p.x, p.y);
    auto dist = [&](float xj, float yj, float n2) {
        const float r2 = dgl_chain_sq(dgl_cross_t<SQ::DIRECT>(p.x, p.y, xj, yj), nj, n2);
        return ALL ? sq.dist_all(ln, r2) : sq.dist(r2);
    };
    const float ds = dist(cx.xs, cx.ys, cx.ns);                                    // d(j, start)
    const float den = __fadd_rn(ds, 1e-7f);
    f[0] = p.x;
    f[1] = p.y;
    f[2] = dist(cx.xc, cx.yc, cx.nc);                                              // d(j, current)
    f[3] = a;
    f[4] = sd;
    f[5] = __fdiv_rn(__fsub_rn(cx.ys, p.y), den);
    f[6] = __fdiv_rn(__fsub_rn(cx.xs, p.x), den);
}

template <class SQ>
__device__ __forceinline__ void tsp_features(const SQ &sq, const dgl_tsp_state_t &st, const TspRowCtx &cx, int j, float *f) {
    tsp_row(sq, cx, reinterpret_cast<const float2 *>(cx.xyb)[j], st.avg[cx.rowN + j], st.sd[cx.rowN + j], f);
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
// The step for row r as a device function (a kernel of its own in tsp_state.hip, a phase of the fused decode kernel
// in rollout.hip).  `lds`: the workgroup's step area; `wave`: this wave's index among the WAVES waves that share the
// row (SPLIT) or among the workgroup's waves (one row per wave); SPLIT contains workgroup barriers: every wave of the
// workgroup must call it the same number of times (the waves of all rows of a workgroup meet at them).
// QUIET: the call takes part in every workgroup barrier but stores nothing to global memory (the spare wave pair of the
// fused decode kernel's last workgroup when the row count is odd: a second in-place update of the same row would read
// moments the first has already replaced).
// SQ: the square-root model (common.h).
template <int WAVES, bool UPDATE, bool SPLIT = false, bool STREAM = false, bool QUIET = false, class SQ = DglSqrtIeee>
__device__ __forceinline__ void tsp_step_row(const dgl_tsp_state_t &st, const int32_t *__restrict__ selected, int t, int knn,
                                             int32_t *__restrict__ idx_out, float *__restrict__ tile_out, int lds_per_wave,
                                             int mom_off, const TspBeamLink &link, float *lds, int r, int wave,
                                             const SQ &sq = SQ{}) {
    const int lane = threadIdx.x & 63;
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
    const typename SQ::Lanes sq_lanes = sq.lanes();         // (all 64 lanes are active here and at the top of the scan loop)
    // The current node and its coordinates are two dependent round trips; they start only now, behind the first
    // group's loads, which do not depend on them.  Wave-uniform values: kept in scalar registers.
    // (clamped: a node id outside [0, N) from a faulty caller must not turn into a wild address)
    const int cur = min(max(__builtin_amdgcn_readfirstlane(UPDATE ? selected[r] : st.tour[cx.rowN + (t - 1)]), 0), N - 1);
    const int start = (t == 1) ? cur : __builtin_amdgcn_readfirstlane(tour_in[0]);
    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.nc = dgl_sqnorm_t<SQ::DIRECT>(cx.xc, cx.yc);
    cx.xs = cx.xyb[2 * start]; cx.ys = cx.xyb[2 * start + 1]; cx.ns = dgl_sqnorm_t<SQ::DIRECT>(cx.xs, cx.ys);
    // visited words of 64 consecutive chunks, one per lane; the word of a chunk is read back as a
    // wave-uniform lane mask (bit = lane) with the excluded nodes of that chunk added
    unsigned long long vis_words = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += 64 * STEP_UNROLL) {
        const Group g = nxt;
        if (j0 + 64 * STEP_UNROLL < j_end) request(j0 + 64 * STEP_UNROLL, nxt);
        if (want_knn && (j0 == j_begin || ((j0 >> 6) & 63) == 0)) vis_words = vis_in[min(((j0 >> 6) & ~63) + lane, W - 1)];
        float a1g[STEP_UNROLL], s1g[STEP_UNROLL];
        if constexpr (SQ::WAVE_WIDE && UPDATE && SQ::GROUPED) {
            float rad[STEP_UNROLL], dcolg[STEP_UNROLL], item[STEP_UNROLL];
#pragma unroll
            for (int q = 0; q < STEP_UNROLL; ++q) {
                const float x = g.p[q].x, y = g.p[q].y;
                rad[q] = dgl_chain_sq(dgl_cross_t<SQ::DIRECT>(cx.xc, cx.yc, x, y), dgl_sqnorm_t<SQ::DIRECT>(x, y), cx.nc);
            }
            sq.dist_all_n(sq_lanes, rad, dcolg);
#pragma unroll
            for (int q = 0; q < STEP_UNROLL; ++q)
                dgl_update_moments_item<false>(dcolg[q], g.a0[q], g.s0[q], up1, du, &a1g[q], &item[q]);
            sq.all_n(sq_lanes, item, s1g);
        }
#pragma unroll
        for (int q = 0; q < STEP_UNROLL; ++q) {
            const int j = j0 + 64 * q + lane;
            // models with wave-wide roots (common.h): both roots by every lane, before the guard (the loads were clamped)
            float a1w = 0.0f, s1w = 0.0f;
            if constexpr (SQ::WAVE_WIDE && UPDATE && SQ::GROUPED) {
                a1w = a1g[q];
                s1w = s1g[q];
            } else if constexpr (SQ::WAVE_WIDE && UPDATE) {
                const float x = g.p[q].x, y = g.p[q].y;
                const float dcol = sq.dist_all(sq_lanes, dgl_chain_sq(dgl_cross_t<SQ::DIRECT>(cx.xc, cx.yc, x, y),
                                                                      dgl_sqnorm_t<SQ::DIRECT>(x, y), cx.nc));
                float item;
                dgl_update_moments_item<false>(dcol, g.a0[q], g.s0[q], up1, du, &a1w, &item);
                s1w = sq.all(sq_lanes, item);
            }
            if (j < j_end) {
                const float x = g.p[q].x, y = g.p[q].y, nj = dgl_sqnorm_t<SQ::DIRECT>(x, y);
                const float cross = dgl_cross_t<SQ::DIRECT>(cx.xc, cx.yc, x, y);     // shared by dis[j][cur] and dis[cur][j]
                float a1 = 0.0f, s1 = 0.0f;
                if constexpr (SQ::WAVE_WIDE && UPDATE) {
                    a1 = a1w;
                    s1 = s1w;
                    if (QUIET) {
                    } else if (STREAM) {
                        __builtin_nontemporal_store(a1, avg + j);
                        __builtin_nontemporal_store(s1, sd + j);
                    } else {
                        avg[j] = a1;
                        sd[j] = s1;
                    }
                } else if (UPDATE) {
                    const float dcol = sq.dist_pk(dgl_chain_sq(cross, nj, cx.nc));             // dis[j][cur]
                    dgl_update_moments<false>(dcol, g.a0[q], g.s0[q], up1, du, &a1, &s1, sq);   // TSP/TSPEnv.py:386, 408-409
                    if (QUIET) {
                    } else if (STREAM) {
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
    if (QUIET) {
    } else if (UPDATE && linked && (!SPLIT || wave == 0)) {
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
        mins[64 * wave + lane] = lane_min;
        __syncthreads();                                    // keys, minima and the avg/sd stores of every wave
        // the 64 classes {lane of any wave} partition the keys as the lanes of one wave do
#pragma unroll
        for (int w = 0; w < WAVES; ++w) lane_min = fminf(lane_min, mins[64 * w + lane]);
        // Long rows: every wave derives the same bound and compacts the candidates of the range it scanned (on
        // long rows the compaction is the longest part of the selection); short rows: wave 0 does it alone and the
        // others retire at once, which is what keeps saturating batches moving.
        const bool share = N >= 2048 && k <= 64;
        if (!share) {
            if (wave != 0 || QUIET) return;
            c = wave_candidates(keys, N, k, lane_min, scratch, &ckey, &cj, SQ::ULPS);
        } else {
            const float bound = __uint_as_float(wave_candidate_bound(lane_min, k, SQ::ULPS));
            const int cw = wave_compact(keys, j_begin, j_end, bound, lists + 128 * wave,
                                        reinterpret_cast<int *>(lists + 128 * wave) + 64);
            if (lane == 0) counts[wave] = cw;
            __syncthreads();                                // `share` is uniform over the workgroup
            if (wave != 0 || QUIET) return;
            c = wave_candidates_from_lists(lists, counts, WAVES, keys, N, k, bound, scratch, &ckey, &cj, SQ::ULPS);
        }
    } else {
        if (QUIET) return;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        c = wave_candidates(keys, N, k, lane_min, scratch, &ckey, &cj, SQ::ULPS);
    }
    // more than 64 neighbours, row shared by the workgroup's waves: one pass with a bound from the scan classes' minima
    bool wide = false;
    if (SPLIT && c < 0 && k > 64 && k <= 256) {
        float *mins = scratch + 128;
        wide = wave_select_wide<WAVES>(sq, keys, N, k, mins, mins + 64 * WAVES, [&](int j, int rank) {
            idx_row[rank] = j;
            tsp_features(sq, st, cx, j, tile + rank * DGL_TILE_STRIDE);
        });
        if (wide && lane == 63) tsp_features(sq, st, cx, cur, tile + k * DGL_TILE_STRIDE);
    }
    if (wide) {
    } else if (c >= 0) {
        // The inputs of every candidate's feature row (and of the anchor token = the current node, on
        // lane 63 when that lane holds no candidate) are requested before the ranking bisection.
        const bool anchor_early = (c < 64) && (lane == 63);
        const int jl = anchor_early ? cur : cj;
        const float2 p = xy2[jl];
        float av, sv;
        if (mom) { const float2 m = mom[jl]; av = m.x; sv = m.y; } else { av = avg[jl]; sv = sd[jl]; }
        int rank;
        const bool pick = wave_pick_k(sq.dist_all(sq_lanes, ckey), c, k, &rank);
        if (pick) idx_row[rank] = cj;
        if constexpr (SQ::WAVE_WIDE) {                      // the rows' roots by every lane, stored by those that hold one
            float f[TSP_F];
            tsp_row<true>(sq, cx, p, av, sv, f, sq_lanes);
            if (pick || anchor_early) {
                float *dst = tile + (pick ? rank : k) * DGL_TILE_STRIDE;
#pragma unroll
                for (int e = 0; e < TSP_F; ++e) dst[e] = f[e];
            }
        } else if (pick || anchor_early) {
            tsp_row(sq, cx, p, av, sv, tile + (pick ? rank : k) * DGL_TILE_STRIDE);
        }
        if (c == 64 && lane == 63) tsp_features(sq, st, cx, cur, tile + k * DGL_TILE_STRIDE);
    } else {
        wave_select_general(sq, keys, N, k, scratch, [&](int j, int rank) {
            idx_row[rank] = j;
            tsp_features(sq, st, cx, j, tile + rank * DGL_TILE_STRIDE);
        });
        if (lane == 63) tsp_features(sq, st, cx, cur, tile + k * DGL_TILE_STRIDE);
    }
    for (int e = k + lane; e < knn; e += 64) idx_row[e] = -1;
    wave_lds_sync();
    wave_minmax_norm_store(tile, k + 1, TSP_F, 0.0f, tile_out + (size_t)r * (knn + 1) * DGL_TILE_STRIDE, scratch);
}

