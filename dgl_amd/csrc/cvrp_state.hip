// CVRP environment state on the device: reset, fused step (action decoding, capacity,
// running features) + the two neighbourhoods (current node, depot) + token tile, dense state
// materialisation, route length, beam permutation.
//
// One wave64 owns one row.  Nodes 0..N (0 = depot), M = N + 1 points per instance.  The
// feasibility mask of the reference (ninf_mask [R][M], CVRP/CVRPEnv.py:633-637) is not stored:
// it is capacity + 1e-6 < demand[j], evaluated where needed.
#include <cstdlib>

#include "common.h"
#include "select.h"
#include "state_common.h"

#define CVRP_F 8       // x, y, demand, d, avg, sd, sin, cos (+ group flag in column 8 of the tile)

__global__ __launch_bounds__(256) void cvrp_reset_kernel(dgl_cvrp_state_t st) {
    dgl_reset_rows(st.xy, st.demand_in_distance ? st.demand : nullptr, st.N + 1, st.rows_per_instance, st.avg, st.sd,
                   st.visited);
    if (blockIdx.x == 0)
        for (int p = threadIdx.x; p < st.rows_per_instance; p += blockDim.x)
            st.capacity[(size_t)blockIdx.y * st.rows_per_instance + p] = 1.0f;
}

// the reset of the square-root models that reproduce the reference's torch build bit for bit: grid (ceil(M / 8), B)
__global__ __launch_bounds__(256) void cvrp_reset_torch_kernel(dgl_cvrp_state_t st, DglSqrtMkl sq) {
    dgl_reset_rows_torch(st.xy, st.demand_in_distance ? st.demand : nullptr, st.N + 1, st.rows_per_instance, st.avg, st.sd,
                         st.visited, sq);
    if (blockIdx.x == 0)
        for (int p = threadIdx.x; p < st.rows_per_instance; p += blockDim.x)
            st.capacity[(size_t)blockIdx.y * st.rows_per_instance + p] = 1.0f;
}

struct CvrpRowCtx {
    const float *xyb, *demb;
    float xc, yc, zc, nc, x0, y0, n0;       // current node (zc = its demand, used when D3), depot (demand 0)
    size_t rowM;
};

// D3: distances over (x, y, demand) as the training reset builds them (CVRP/CVRPEnv.py:519:
// cdist over all three columns of the instance): |p|^2 = (x^2 + y^2) + z^2 and one more FMA in the
// product chain.  The depot's demand is 0, so its products need no third term.
// DIRECT (at most 25 points, depot included: cdist's direct evaluation, common.h): the norms are 0 and the cross term is
// the whole squared distance fma(dz, dz, fma(dy, dy, dx dx)) - the depot's third term then counts like any other.
template <bool D3, bool DIRECT = false>
__device__ __forceinline__ float cvrp_norm(float x, float y, float z) {
    if (DIRECT) return 0.0f;
    const float n = dgl_sqnorm(x, y);
    return D3 ? __fadd_rn(n, __fmul_rn(z, z)) : n;
}
template <bool D3, bool DIRECT = false>
__device__ __forceinline__ float cvrp_cross(float xi, float yi, float zi, float xj, float yj, float zj) {
    if (DIRECT) {
        const float d2 = dgl_direct_sq(xi, yi, xj, yj), dz = __fsub_rn(zi, zj);
        return D3 ? __fmaf_rn(dz, dz, d2) : d2;
    }
    const float c = dgl_chain_cross(xi, yi, xj, yj);
    return D3 ? __fmaf_rn(-2.0f * zi, zj, c) : c;
}
// cross term against the depot (demand 0: the product chain needs no third term)
template <bool D3, bool DIRECT>
__device__ __forceinline__ float cvrp_cross_depot(float x, float y, float z, float x0, float y0) {
    return DIRECT ? cvrp_cross<D3, true>(x, y, z, x0, y0, 0.0f) : dgl_chain_cross(x, y, x0, y0);
}

// raw data row of a node at p with demand dem and running statistics (a, sd)
// (CVRP/CVRPEnv.py:696,718,746,782,801); returns d(j, depot)
// ALL: called by all 64 lanes of the wave together (the model's wave-wide roots, common.h)
template <bool D3, bool ALL = false, class SQ>
__device__ __forceinline__ float cvrp_row(const SQ &sq, const CvrpRowCtx &cx, float2 p, float dem, float a, float sd, float *f,
                                          const typename SQ::Lanes &ln = typename SQ::Lanes{}) {
    const float nj = cvrp_norm<D3, SQ::DIRECT>(p.x, p.y, dem);
    auto root = [&](float r2) { return ALL ? sq.dist_all(ln, r2) : sq.dist(r2); };
    const float d0 = root(dgl_chain_sq(cvrp_cross_depot<D3, SQ::DIRECT>(p.x, p.y, dem, cx.x0, cx.y0), nj, cx.n0));   // d(j, depot)
    const float den = __fadd_rn(d0, 1e-7f);
    f[0] = p.x;
    f[1] = p.y;
    f[2] = dem;
    f[3] = root(dgl_chain_sq(cvrp_cross<D3, SQ::DIRECT>(p.x, p.y, dem, cx.xc, cx.yc, cx.zc), nj, cx.nc));   // d(j, current)
    f[4] = a;
    f[5] = sd;
    f[6] = __fdiv_rn(__fsub_rn(cx.y0, p.y), den);
    f[7] = __fdiv_rn(__fsub_rn(cx.x0, p.x), den);
    return d0;
}

template <bool D3, class SQ>
__device__ __forceinline__ float cvrp_features(const SQ &sq, const dgl_cvrp_state_t &st, const CvrpRowCtx &cx, int j, float *f) {
    return cvrp_row<D3>(sq, cx, reinterpret_cast<const float2 *>(cx.xyb)[j], cx.demb[j], st.avg[cx.rowM + j],
                    st.sd[cx.rowM + j], f);
}

#define STEP_UNROLL 4     // node chunks (of 64) whose loads are issued together: bytes in flight per wave

// One wave per row.  LDS per wave: keysL[M] + keysD[M] + tile[(knn+depot_knn+2)*STRIDE] + scratch[128].
// SPLIT: the WAVES waves of a workgroup share one row (each scans a contiguous part of the nodes, wave 0
// selects and writes the tiles), as in tsp_step_kernel.
// Beam search: row r of the state being written continues row (r / beam) * beam + prev_k[r] of another copy of the
// state; avg / sd / visited / capacity are read from that row and written to row r.  The action and the parent row
// go to a history from which nodes and flags are rebuilt at the end (cvrp_backtrack_kernel).  prev_k == NULL: the
// ordinary in-place step.
struct CvrpBeamLink {
    const float *avg_in, *sd_in, *cap_in;
    const uint64_t *vis_in;
    const int32_t *nodes_in;
    const int8_t *flags_in;
    const int32_t *prev_k;
    int beam;
    int32_t *hist_action, *hist_parent;    // [N][R], entry (t - 1) * R + r
};

template <class SQ, int WAVES, bool UPDATE, bool D3, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_sgpr(96))) void cvrp_step_kernel(dgl_cvrp_state_t st, const int32_t *__restrict__ action,
                                                              int t, int knn, int depot_knn,
                                                              int32_t *__restrict__ idx_last_out,
                                                              int32_t *__restrict__ idx_dep_out,
                                                              uint8_t *__restrict__ infeasible_out,
                                                              float *__restrict__ tile_out, int lds_per_wave,
                                                              CvrpBeamLink link, SQ sq) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar registers
    const int r = SPLIT ? blockIdx.x : blockIdx.x * WAVES + wave;
    if (r >= st.R) return;
    const int N = st.N, M = N + 1, W = (M + 63) / 64;
    float *keysL = lds + (SPLIT ? 0 : (size_t)wave * lds_per_wave);
    float *keysD = keysL + M;
    float *tile = keysD + M;
    float *scratch = tile + (knn + depot_knn + 2) * DGL_TILE_STRIDE;
    int j_begin = 0, j_end = M;                                // this wave's node range: whole groups of chunks
    if (SPLIT) {
        const int per = ((W + WAVES - 1) / WAVES + STEP_UNROLL - 1) / STEP_UNROLL * STEP_UNROLL;
        j_begin = wave * per * 64;
        j_end = min(M, j_begin + per * 64);
    }

    const int b = r / st.rows_per_instance;
    CvrpRowCtx cx;
    cx.xyb = st.xy + (size_t)b * M * 2;
    cx.demb = st.demand + (size_t)b * M;
    cx.rowM = (size_t)r * M;

    // ---- action decoding, capacity (CVRP/CVRPEnv.py:612-631)
    // wave-uniform: scalar registers (clamped: an action id outside [0, 2(N+1)) must not turn into a wild address)
    const int act = UPDATE ? min(max(__builtin_amdgcn_readfirstlane(action[r]), 0), 2 * N + 1) : 0;
    const bool via = act > N;
    const int cur = UPDATE ? (via ? act - (N + 1) : act) : __builtin_amdgcn_readfirstlane(st.nodes[(size_t)r * N + (t - 1)]);
    const float dem_cur = cx.demb[cur];
    const bool linked = UPDATE && link.prev_k != nullptr;
    int sr = r;                                                // the row this one continues
    if (linked) sr = (r / link.beam) * link.beam + min(max(__builtin_amdgcn_readfirstlane(link.prev_k[r]), 0), link.beam - 1);
    const float cap_old = linked ? link.cap_in[sr] : st.capacity[r];
    float cap = !UPDATE ? cap_old : (via ? __fsub_rn(1.0f, dem_cur) : __fsub_rn(cap_old, dem_cur));
    const float cap_eps = __fadd_rn(cap, 0.000001f);          // feasibility uses the un-nudged capacity

    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.zc = dem_cur; cx.nc = cvrp_norm<D3, SQ::DIRECT>(cx.xc, cx.yc, cx.zc);
    cx.x0 = cx.xyb[0]; cx.y0 = cx.xyb[1]; cx.n0 = dgl_sqnorm_t<SQ::DIRECT>(cx.x0, cx.y0);

    const int u = M - t;                                       // CVRP/CVRPEnv.py:662 (>= 1: t <= N)
    const float up1 = (float)(u + 1);
    const DglConstDiv du = dgl_const_div((float)u);
    const bool want_knn = (idx_last_out != nullptr) && (t < N);
    float minL = INFINITY, minD = INFINITY;
    uint64_t *vis = st.visited + (size_t)r * W;
    const float2 *xy2 = reinterpret_cast<const float2 *>(cx.xyb);
    float *avg = st.avg + cx.rowM, *sd = st.sd + cx.rowM;
    const float *avg_in = linked ? link.avg_in + (size_t)sr * M : avg, *sd_in = linked ? link.sd_in + (size_t)sr * M : sd;
    const uint64_t *vis_in = linked ? link.vis_in + (size_t)sr * W : vis;
    // The loads of a group of STEP_UNROLL chunks are issued one group ahead of its arithmetic
    // (clamped addresses keep them unconditional).
    struct Group {
        float2 p[STEP_UNROLL];
        float a0[STEP_UNROLL], s0[STEP_UNROLL], dm[STEP_UNROLL];
    };
    auto request = [&](int j0, Group &g) {
#pragma unroll
        for (int q = 0; q < STEP_UNROLL; ++q) {
            const int j = min(j0 + 64 * q + lane, M - 1);
            g.p[q] = xy2[j];
            if (UPDATE) { g.a0[q] = avg_in[j]; g.s0[q] = sd_in[j]; }
            g.dm[q] = D3 ? cx.demb[j] : 0.0f;
        }
    };
    Group nxt;
    request(j_begin, nxt);
    const typename SQ::Lanes sq_lanes = sq.lanes();         // (all 64 lanes are active here and at the top of the scan loop)
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
                rad[q] = dgl_chain_sq(cvrp_cross<D3, SQ::DIRECT>(cx.xc, cx.yc, cx.zc, x, y, g.dm[q]),
                                      cvrp_norm<D3, SQ::DIRECT>(x, y, g.dm[q]), cx.nc);
            }
            sq.dist_all_n(sq_lanes, rad, dcolg);
#pragma unroll
            for (int q = 0; q < STEP_UNROLL; ++q) {
                const float a0 = g.a0[q], s0 = g.s0[q], dcol = dcolg[q];
                a1g[q] = dgl_div(__fsub_rn(__fmul_rn(a0, up1), dcol), du);                          // :711
                const float e = __fsub_rn(dcol, a0);
                const float it = dgl_div(__fsub_rn(__fmul_rn(__fmul_rn(s0, s0), up1), __fmul_rn(e, e)), du);   // :733
                item[q] = it < 0.0f ? 0.0f : it;                                                    // :735-736
            }
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
                const float dcol = sq.dist_all(sq_lanes, dgl_chain_sq(cvrp_cross<D3, SQ::DIRECT>(cx.xc, cx.yc, cx.zc, x, y, g.dm[q]),
                                                                      cvrp_norm<D3, SQ::DIRECT>(x, y, g.dm[q]), cx.nc));
                const float a0 = g.a0[q], s0 = g.s0[q];
                a1w = dgl_div(__fsub_rn(__fmul_rn(a0, up1), dcol), du);                             // :711
                const float e = __fsub_rn(dcol, a0);
                const float it = dgl_div(__fsub_rn(__fmul_rn(__fmul_rn(s0, s0), up1), __fmul_rn(e, e)), du);   // :733
                s1w = sq.all(sq_lanes, it < 0.0f ? 0.0f : it);                                      // :735-738
            }
            if (j < j_end) {
                const float x = g.p[q].x, y = g.p[q].y, nj = cvrp_norm<D3, SQ::DIRECT>(x, y, g.dm[q]);
                const float cross = cvrp_cross<D3, SQ::DIRECT>(cx.xc, cx.yc, cx.zc, x, y, g.dm[q]);   // shared by dis[j][cur] and dis[cur][j]
                if constexpr (SQ::WAVE_WIDE && UPDATE) {
                    avg[j] = a1w;
                    sd[j] = s1w;
                } else if (UPDATE) {
                    const float dcol = sq.dist(dgl_chain_sq(cross, nj, cx.nc));                       // dis[j][cur]
                    const float a0 = g.a0[q], s0 = g.s0[q];
                    const float a1 = dgl_div(__fsub_rn(__fmul_rn(a0, up1), dcol), du);             // :711
                    const float e = __fsub_rn(dcol, a0);
                    float item = dgl_div(__fsub_rn(__fmul_rn(__fmul_rn(s0, s0), up1), __fmul_rn(e, e)), du);   // :733
                    if (item < 0.0f) item = 0.0f;                                                   // :735-736
                    avg[j] = a1;
                    sd[j] = sq(item);
                }
                if (want_knn) {
                    // keys are squared distances dis[cur][j]^2 and dis[0][j]^2; visited nodes, the current
                    // node and the depot never rank (the reference adds 1e2 to them, CVRP/CVRPModel.py:112-119)
                    const int chunk = (j0 >> 6) + q;
                    unsigned long long excl = dgl_readlane_u64(vis_words, chunk & 63);
                    if (chunk == (cur >> 6)) excl |= 1ull << (cur & 63);
                    if (chunk == 0) excl |= 1ull;
                    const float kl = dgl_inf_where(excl, dgl_chain_sq(cross, cx.nc, nj));
                    const float kd = dgl_inf_where(excl, dgl_chain_sq(cvrp_cross_depot<D3, SQ::DIRECT>(x, y, g.dm[q], cx.x0, cx.y0), cx.n0, nj));
                    keysL[j] = kl;
                    keysD[j] = kd;
                    minL = fminf(minL, kl);
                    minD = fminf(minD, kd);
                }
            }
        }
    }
    if (want_knn) {
        // CVRPModel.get_knn_state nudges near-zero capacities in place before it divides by them
        // (CVRP/CVRPModel.py:164-166); the nudged value is what the next step starts from.
        if (cap < 0.00001f) cap = __fadd_rn(cap, 0.0000001f);
    }
    if (SPLIT) {
        float *mins = scratch + 128;
        mins[threadIdx.x] = minL;
        mins[WAVES * 64 + threadIdx.x] = minD;
        // every wave has read the row's capacity; keys, minima and the avg/sd stores of every wave are complete
        __syncthreads();
    }
    if (linked && (!SPLIT || wave == 0)) {
        for (int w = lane; w < W; w += 64) vis[w] = vis_in[w] | (w == (cur >> 6) ? 1ull << (cur & 63) : 0ull);
        if (lane == 0) {
            st.capacity[r] = cap;
            link.hist_action[(size_t)(t - 1) * st.R + r] = act;
            link.hist_parent[(size_t)(t - 1) * st.R + r] = sr;
            // every row of a beam group shares its first move
            st.nodes[(size_t)r * N] = link.nodes_in[(size_t)sr * N];
            st.flags[(size_t)r * N] = link.flags_in[(size_t)sr * N];
        }
    } else if (lane == 0 && (!SPLIT || wave == 0)) {
        st.capacity[r] = cap;
        if (UPDATE) {
            st.nodes[(size_t)r * N + (t - 1)] = cur;
            st.flags[(size_t)r * N + (t - 1)] = via ? 1 : 0;
            vis[cur >> 6] |= 1ull << (cur & 63);
        }
    }
    if (!want_knn) return;

    // LDS keys of all lanes, and (for the feature loads below) the avg/sd stores of all lanes
    // SPLIT: wave 0 builds the group of the current node, wave 1 the depot's group, side by side
    const bool do_last = !SPLIT || wave == 0, do_depot = !SPLIT || wave == 1;
    const int kd = min(depot_knn, N - t);                      // CVRP/CVRPModel.py:58,88
    const int kl = min(knn, kd);                               // :123
    // Long rows: every wave compacts the candidates of both groups over the range it scanned, into lists that
    // waves 0 and 1 then concatenate (short rows: those two compact alone and the others retire at once).
    const bool share = SPLIT && M >= 2048 && kd <= 64;
    const float *lists = nullptr;
    const int *counts = nullptr;
    float boundL = 0.0f, boundD = 0.0f;
    const float *wide_min = nullptr;
    float *wide_list = nullptr;
    if (SPLIT) {
        if (!share && wave >= 2) return;
        // LDS behind the row's arrays: scratch of wave 0 [128] | lane minima [2][64 WAVES] | scratch of wave 1 [128]
        //                              | candidate lists [2][WAVES][128] | counts [2][WAVES]
        const float *mins = scratch + 128;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            minL = fminf(minL, mins[64 * w + lane]);
            minD = fminf(minD, mins[WAVES * 64 + 64 * w + lane]);
        }
        if (share) {
            float *lw = scratch + 256 + 2 * WAVES * 64;
            int *cw = reinterpret_cast<int *>(lw + 2 * WAVES * 128);
            boundL = __uint_as_float(wave_candidate_bound(minL, kl, SQ::ULPS));
            boundD = __uint_as_float(wave_candidate_bound(minD, kd, SQ::ULPS));
            float *mine = lw + 128 * wave;
            const int cl = wave_compact(keysL, j_begin, j_end, boundL, mine, reinterpret_cast<int *>(mine) + 64);
            mine += 128 * WAVES;
            const int cd = wave_compact(keysD, j_begin, j_end, boundD, mine, reinterpret_cast<int *>(mine) + 64);
            if (lane == 0) { cw[wave] = cl; cw[WAVES + wave] = cd; }
            __syncthreads();                                   // `share` is uniform over the workgroup
            if (wave >= 2) return;
            lists = lw + 128 * WAVES * wave;
            counts = cw + WAVES * wave;
        }
        // the wide selection (more than 64 neighbours): this group's class minima and its half of the list area
        wide_min = mins + WAVES * 64 * wave;
        wide_list = scratch + 256 + 2 * WAVES * 64 + 128 * WAVES * wave;
        if (wave == 1) scratch += 128 + 2 * WAVES * 64;        // a search scratch of its own
    } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    const bool low = cap < 0.00001f;
    int32_t *il = idx_last_out + (size_t)r * knn;
    int32_t *id = idx_dep_out + (size_t)r * depot_knn;
    uint8_t *inf_row = infeasible_out + (size_t)r * knn;
    float *tileL = tile, *tileD = tile + (kl + 1) * DGL_TILE_STRIDE;

    // neighbourhood of the current node: feasibility flag and demand / capacity (CVRP/CVRPEnv.py:634,
    // CVRP/CVRPModel.py:172-174)
    auto finish_last = [&](float *f, int rank) {
        inf_row[rank] = cap_eps < f[2] ? 1 : 0;
        f[2] = low ? 1.1f : __fdiv_rn(f[2], cap);
    };
    float ckey;
    int cj;
    int c = -1;
    if (do_last) {
    c = share ? wave_candidates_from_lists(lists, counts, WAVES, keysL, M, kl, boundL, scratch, &ckey, &cj, SQ::ULPS)
              : wave_candidates(keysL, M, kl, minL, scratch, &ckey, &cj, SQ::ULPS);
    if (c >= 0) {
        // inputs of every candidate's row (and of the anchor, on lane 63 when it holds no candidate)
        // are requested before the ranking bisection
        const bool anchor_early = (c < 64) && (lane == 63);
        const int jl = anchor_early ? cur : cj;
        const float2 p = xy2[jl];
        const float dm = cx.demb[jl], av = avg[jl], sv = sd[jl];
        int rank;
        const bool pick = wave_pick_k(sq.dist_all(sq_lanes, ckey), c, kl, &rank);
        float fw[CVRP_F];
        if constexpr (SQ::WAVE_WIDE) cvrp_row<D3, true>(sq, cx, p, dm, av, sv, fw, sq_lanes);   // the rows' roots by every lane
        if (pick || anchor_early) {
            float *f = tileL + (pick ? rank : kl) * DGL_TILE_STRIDE;
            if constexpr (SQ::WAVE_WIDE) {
#pragma unroll
                for (int e = 0; e < CVRP_F; ++e) f[e] = fw[e];
            } else {
                cvrp_row<D3>(sq, cx, p, dm, av, sv, f);
            }
            if (pick) {
                il[rank] = cj;
                finish_last(f, rank);
            }
        }
        if (c == 64 && lane == 63) cvrp_features<D3>(sq, st, cx, cur, tileL + kl * DGL_TILE_STRIDE);
    } else {
        auto emit_last = [&](int j, int rank) {
            il[rank] = j;
            float *f = tileL + rank * DGL_TILE_STRIDE;
            cvrp_features<D3>(sq, st, cx, j, f);
            finish_last(f, rank);
        };
        if (!(SPLIT && kl > 64 && kl <= 256 && wave_select_wide<WAVES>(sq, keysL, M, kl, wide_min, wide_list, emit_last)))
        wave_select_general(sq, keysL, M, kl, scratch, [&](int j, int rank) {
            il[rank] = j;
            float *f = tileL + rank * DGL_TILE_STRIDE;
            cvrp_features<D3>(sq, st, cx, j, f);
            finish_last(f, rank);
        });
        if (lane == 63) cvrp_features<D3>(sq, st, cx, cur, tileL + kl * DGL_TILE_STRIDE);
    }
    for (int e = kl + lane; e < knn; e += 64) { il[e] = -1; inf_row[e] = 0; }
    wave_lds_sync();                                           // scratch is reused by the second search
    }
    if (do_depot) {
    // neighbourhood of the depot: column 3 holds distance_to_depot (CVRP/CVRPModel.py:180-183)
    c = share ? wave_candidates_from_lists(lists, counts, WAVES, keysD, M, kd, boundD, scratch, &ckey, &cj, SQ::ULPS)
              : wave_candidates(keysD, M, kd, minD, scratch, &ckey, &cj, SQ::ULPS);
    if (c >= 0) {
        const bool anchor_early = (c < 64) && (lane == 63);
        const int jl = anchor_early ? 0 : cj;
        const float2 p = xy2[jl];
        const float dm = cx.demb[jl], av = avg[jl], sv = sd[jl];
        int rank;
        const bool pick = wave_pick_k(sq.dist_all(sq_lanes, ckey), c, kd, &rank);
        float fw[CVRP_F], d0w = 0.0f;
        if constexpr (SQ::WAVE_WIDE) d0w = cvrp_row<D3, true>(sq, cx, p, dm, av, sv, fw, sq_lanes);   // the rows' roots by every lane
        if (pick || anchor_early) {
            float *f = tileD + (pick ? rank : kd) * DGL_TILE_STRIDE;
            float d0 = d0w;
            if constexpr (SQ::WAVE_WIDE) {
#pragma unroll
                for (int e = 0; e < CVRP_F; ++e) f[e] = fw[e];
            } else {
                d0 = cvrp_row<D3>(sq, cx, p, dm, av, sv, f);
            }
            if (pick) {
                id[rank] = cj;
                f[3] = d0;
            }
        }
        if (c == 64 && lane == 63) cvrp_features<D3>(sq, st, cx, 0, tileD + kd * DGL_TILE_STRIDE);
    } else {
        auto emit_depot = [&](int j, int rank) {
            id[rank] = j;
            float *f = tileD + rank * DGL_TILE_STRIDE;
            f[3] = cvrp_features<D3>(sq, st, cx, j, f);
        };
        if (!(SPLIT && kd > 64 && kd <= 256 && wave_select_wide<WAVES>(sq, keysD, M, kd, wide_min, wide_list, emit_depot)))
        wave_select_general(sq, keysD, M, kd, scratch, [&](int j, int rank) {
            id[rank] = j;
            float *f = tileD + rank * DGL_TILE_STRIDE;
            f[3] = cvrp_features<D3>(sq, st, cx, j, f);
        });
        if (lane == 63) cvrp_features<D3>(sq, st, cx, 0, tileD + kd * DGL_TILE_STRIDE);
    }
    for (int e = kd + lane; e < depot_knn; e += 64) id[e] = -1;
    wave_lds_sync();
    }

    float *tout = tile_out + (size_t)r * (knn + depot_knn + 2) * DGL_TILE_STRIDE;
    float *toutD = tout + (kl + 1) * DGL_TILE_STRIDE;
    // min-max per group (CVRP/CVRPModel.py:299,301); the column after the data is the group flag (:303-304)
    if (do_last) wave_minmax_norm_store(tileL, kl + 1, CVRP_F, 0.0f, tout, scratch);
    wave_lds_sync();                                           // scratch is reused
    if (do_depot) wave_minmax_norm_store(tileD, kd + 1, CVRP_F, 1.0f, toutD, scratch);
}

// ---------------------------------------------------------------- dense state (API parity)
template <bool D3, class SQ>
__global__ void cvrp_state_data_kernel(dgl_cvrp_state_t st, int t, float *__restrict__ out, SQ sq) {
    const int r = blockIdx.y, N = st.N, M = N + 1;
    const int b = r / st.rows_per_instance;
    CvrpRowCtx cx;
    cx.xyb = st.xy + (size_t)b * M * 2;
    cx.demb = st.demand + (size_t)b * M;
    cx.rowM = (size_t)r * M;
    const int cur = st.nodes[(size_t)r * N + (t - 1)];
    cx.xc = cx.xyb[2 * cur]; cx.yc = cx.xyb[2 * cur + 1]; cx.zc = cx.demb[cur]; cx.nc = cvrp_norm<D3, SQ::DIRECT>(cx.xc, cx.yc, cx.zc);
    cx.x0 = cx.xyb[0]; cx.y0 = cx.xyb[1]; cx.n0 = dgl_sqnorm_t<SQ::DIRECT>(cx.x0, cx.y0);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        float f[CVRP_F];
        cvrp_features<D3>(sq, st, cx, j, f);
        float *o = out + (cx.rowM + j) * CVRP_F;
#pragma unroll
        for (int c = 0; c < CVRP_F; ++c) o[c] = f[c];
    }
}

// -------------------------------------------------------------------- route length
// CVRP/CVRPEnv.py:951-993: position p contributes |cur - mid| + |prev - mid'| with mid = depot if
// the move goes via the depot else cur itself, mid' = mid except depot at p = 0; prev of p = 0 is
// the last customer (closing leg).  Segments sqrt(fl(dx^2)+fl(dy^2)), summed in binary64.
__device__ __forceinline__ float seg_len(const float *xy, int a, int c) {
    const float dx = __fsub_rn(xy[2 * a], xy[2 * c]), dy = __fsub_rn(xy[2 * a + 1], xy[2 * c + 1]);
    return sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}
__global__ __launch_bounds__(256) void cvrp_length_kernel(dgl_cvrp_state_t st, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= st.R) return;
    const int N = st.N, M = N + 1;
    const float *xyb = st.xy + (size_t)(r / st.rows_per_instance) * M * 2;
    const int32_t *nodes = st.nodes + (size_t)r * N;
    const int8_t *flags = st.flags + (size_t)r * N;
    double s = 0.0;
    for (int p = lane; p < N; p += 64) {
        const int cur = nodes[p], prev = nodes[p == 0 ? N - 1 : p - 1];
        const int mid = flags[p] ? 0 : cur;
        const int mid2 = p == 0 ? 0 : mid;
        s += (double)seg_len(xyb, cur, mid) + (double)seg_len(xyb, prev, mid2);
    }
    s = wave_sum_f64(s);
    if (lane == 0) out[r] = (float)s;
}

// The same length as the reference's torch build returns it (CVRP/CVRPEnv.py:991): both roots of a step through the
// reference-bits model, added, and the steps summed in torch.sum's order (dgl_torch_row_sum).  Half a wave per row.
__global__ __launch_bounds__(256) void cvrp_length_torch_kernel(dgl_cvrp_state_t st, float *__restrict__ out, DglSqrtMkl sq) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), r = min(row, st.R - 1);
    const int N = st.N, M = N + 1;
    const float *xyb = st.xy + (size_t)(r / st.rows_per_instance) * M * 2;
    const int32_t *nodes = st.nodes + (size_t)r * N;
    const int8_t *flags = st.flags + (size_t)r * N;
    const DglSqrtMkl::Lanes ln = sq.lanes();
    const float total = dgl_torch_row_sum(N, [&](int p, auto all_lanes) {
        const int cur = min(max(nodes[p], 0), N), prev = min(max(nodes[p == 0 ? N - 1 : p - 1], 0), N);
        const int mid = flags[p] ? 0 : cur;
        const int mid2 = p == 0 ? 0 : mid;
        auto sq2 = [&](int a, int c) {
            const float dx = __fsub_rn(xyb[2 * a], xyb[2 * c]), dy = __fsub_rn(xyb[2 * a + 1], xyb[2 * c + 1]);
            return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        };
        const float a2 = sq2(cur, mid), b2 = sq2(prev, mid2);
        return decltype(all_lanes)::value ? __fadd_rn(sq.all(ln, a2), sq.all(ln, b2)) : __fadd_rn(sq(a2), sq(b2));
    });
    if (row < st.R && (threadIdx.x & 31) == 0) out[row] = total;
}

// ------------------------------------------------------------------ beam permutation
__global__ __launch_bounds__(256) void cvrp_permute_kernel(dgl_cvrp_state_t src, dgl_cvrp_state_t dst,
                                                          const int32_t *__restrict__ prev_k, int beam, int t) {
    const int r = blockIdx.x, g = r / beam, sr = g * beam + min(max(prev_k[r], 0), beam - 1);
    const int N = src.N, M = N + 1, W = (M + 63) / 64;
    const size_t so = (size_t)sr * M, d0 = (size_t)r * M;
    for (int j = threadIdx.x; j < M; j += blockDim.x) {
        dst.avg[d0 + j] = src.avg[so + j];
        dst.sd[d0 + j] = src.sd[so + j];
    }
    for (int p = threadIdx.x; p < t; p += blockDim.x) {
        dst.nodes[(size_t)r * N + p] = src.nodes[(size_t)sr * N + p];
        dst.flags[(size_t)r * N + p] = src.flags[(size_t)sr * N + p];
    }
    for (int w = threadIdx.x; w < W; w += blockDim.x) dst.visited[(size_t)r * W + w] = src.visited[(size_t)sr * W + w];
    if (threadIdx.x == 0) dst.capacity[r] = src.capacity[sr];
}

// ------------------------------------------------------------------------ launchers
static int check_cvrp_state(const dgl_cvrp_state_t *st) {
    DGL_CHECK_ARG(st != nullptr, "state is NULL");
    DGL_CHECK_ARG(st->B > 0 && st->N > 1 && st->rows_per_instance > 0, "bad sizes B=%d N=%d rows_per_instance=%d",
                  st->B, st->N, st->rows_per_instance);
    DGL_CHECK_ARG(st->R == st->B * st->rows_per_instance, "R=%d != B*rows_per_instance", st->R);
    DGL_CHECK_ARG(st->xy && st->demand && st->avg && st->sd && st->capacity && st->visited && st->nodes && st->flags,
                  "NULL state buffer");
    return DGL_OK;
}

extern "C" int dgl_cvrp_reset(const dgl_cvrp_state_t *st, dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    const int M = st->N + 1;
    size_t lds = (size_t)M * (st->demand_in_distance ? 3 : 2) * sizeof(float);
    DGL_CHECK_ARG(lds <= 150 * 1024, "N=%d too large for the LDS-staged reset", st->N);
    if (dgl_sqrt_reference_bits()) {
        if (lds > 48 * 1024)
            dgl_max_dynamic_lds((const void *)cvrp_reset_torch_kernel, (int)lds);
        hipLaunchKernelGGL(cvrp_reset_torch_kernel, dim3((M + 7) / 8, st->B), dim3(256), lds, (hipStream_t)stream, *st,
                           DglSqrtMkl{});
        DGL_CHECK_LAUNCH("dgl_cvrp_reset");
        return DGL_OK;
    }
    if (lds > 48 * 1024)
        dgl_max_dynamic_lds((const void *)cvrp_reset_kernel, (int)lds);
    hipLaunchKernelGGL(cvrp_reset_kernel, dim3((M + 255) / 256, st->B), dim3(256), lds, (hipStream_t)stream, *st);
    DGL_CHECK_LAUNCH("dgl_cvrp_reset");
    return DGL_OK;
}

// Instances of at most 25 points, depot included (torch.cdist's direct evaluation, common.h): one wave per row.
template <bool UPDATE, bool D3, class SQ>
static int launch_cvrp_step_small(const dgl_cvrp_state_t *st, const int32_t *action, int32_t t, int32_t knn, int32_t depot_knn,
                                  int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                                  dgl_stream_t stream, const char *what, const CvrpBeamLink &link, SQ sq) {
    static_assert(SQ::DIRECT, "the small-instance shape is the direct evaluation's");
    const int M = st->N + 1;
    const int per_wave = 2 * M + (knn + depot_knn + 2) * DGL_TILE_STRIDE + 128;
    const size_t lds1 = (size_t)per_wave * sizeof(float);
    DGL_CHECK_ARG(lds1 <= 150 * 1024, "N=%d, knn=%d/%d need %zu bytes of LDS per row (max 153600)", st->N, knn,
                  depot_knn, lds1);
    if (lds1 > 48 * 1024)
        dgl_max_dynamic_lds((const void *)cvrp_step_kernel<SQ, 1, UPDATE, D3>, (int)lds1);
    hipLaunchKernelGGL((cvrp_step_kernel<SQ, 1, UPDATE, D3>), dim3(st->R), dim3(64), lds1, (hipStream_t)stream, *st, action,
                       t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out, tile_out, per_wave, link, sq);
    DGL_CHECK_LAUNCH(what);
    return DGL_OK;
}

template <bool UPDATE, bool D3, class SQ>
static int launch_cvrp_step_sq(const dgl_cvrp_state_t *st, const int32_t *action, int32_t t, int32_t knn, int32_t depot_knn,
                               int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                               dgl_stream_t stream, const char *what, const CvrpBeamLink &link, SQ sq) {
    const int M = st->N + 1;
    const int per_wave = 2 * M + (knn + depot_knn + 2) * DGL_TILE_STRIDE + 128;
    size_t lds4 = (size_t)per_wave * 4 * sizeof(float);
    // long rows, few of them: eight waves share a row (more loads in flight per CU), as in the TSP launcher
    static const int force_waves = getenv("DGL_STEP_WAVES") ? atoi(getenv("DGL_STEP_WAVES")) : 0;            // experiments
    const int waves = force_waves == 4 || force_waves == 8 ? force_waves : (st->N >= 4096 && st->R <= 1024 ? 8 : 4);
    // behind the row's arrays: lane minima [2][64 waves] | a second search scratch [128] | lists [2][waves][128] | counts
    const size_t lds_split = (size_t)(per_wave + 2 * 64 * waves + 128 + 2 * waves * 128 + 2 * waves) * sizeof(float);
    if ((st->R < 2048 || st->R >= 8192) && st->N >= 512 && lds_split <= 150 * 1024) {
        // few rows: four (or eight) waves share a row
        if (waves == 8) {
            if (lds_split > 48 * 1024)
                dgl_max_dynamic_lds((const void *)cvrp_step_kernel<SQ, 8, UPDATE, D3, true>, (int)lds_split);
            hipLaunchKernelGGL((cvrp_step_kernel<SQ, 8, UPDATE, D3, true>), dim3(st->R), dim3(512), lds_split, (hipStream_t)stream,
                               *st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out, tile_out,
                               per_wave, link, sq);
        } else {
            if (lds_split > 48 * 1024)
                dgl_max_dynamic_lds((const void *)cvrp_step_kernel<SQ, 4, UPDATE, D3, true>, (int)lds_split);
            hipLaunchKernelGGL((cvrp_step_kernel<SQ, 4, UPDATE, D3, true>), dim3(st->R), dim3(256), lds_split, (hipStream_t)stream,
                               *st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out, tile_out,
                               per_wave, link, sq);
        }
    } else if (lds4 <= 64 * 1024 && st->R >= 2048) {
        hipLaunchKernelGGL((cvrp_step_kernel<SQ, 4, UPDATE, D3>), dim3((st->R + 3) / 4), dim3(256), lds4, (hipStream_t)stream,
                           *st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out, tile_out,
                           per_wave, link, sq);
    } else {
        size_t lds1 = (size_t)per_wave * sizeof(float);
        DGL_CHECK_ARG(lds1 <= 150 * 1024, "N=%d, knn=%d/%d need %zu bytes of LDS per row (max 153600)", st->N, knn,
                      depot_knn, lds1);
        if (lds1 > 48 * 1024)
            dgl_max_dynamic_lds((const void *)cvrp_step_kernel<SQ, 1, UPDATE, D3>, (int)lds1);
        hipLaunchKernelGGL((cvrp_step_kernel<SQ, 1, UPDATE, D3>), dim3(st->R), dim3(64), lds1, (hipStream_t)stream, *st, action,
                           t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out, tile_out, per_wave, link, sq);
    }
    DGL_CHECK_LAUNCH(what);
    return DGL_OK;
}

// the square-root model in force on the current device picks the instantiation (dgl_set_sqrt_model)
template <bool UPDATE, bool D3>
static int launch_cvrp_step(const dgl_cvrp_state_t *st, const int32_t *action, int32_t t, int32_t knn, int32_t depot_knn,
                            int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                            dgl_stream_t stream, const char *what, const CvrpBeamLink &link = CvrpBeamLink{}) {
    if (st->N + 1 <= DGL_DIRECT_MAX_POINTS) {
        if (dgl_sqrt_reference_bits())
            return launch_cvrp_step_small<UPDATE, D3>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                                      tile_out, stream, what, link, DglSqrtMklT<true>{});
        return launch_cvrp_step_small<UPDATE, D3>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                                  tile_out, stream, what, link, DglSqrtIeeeT<true>{});
    }
    if (dgl_sqrt_reference_bits()) {
        if (st->R >= 2048)                                     // (common.h, GROUPED: registers count where the batch saturates the card)
            return launch_cvrp_step_sq<UPDATE, D3>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                                   tile_out, stream, what, link, DglSqrtMklT<false, false>{});
        return launch_cvrp_step_sq<UPDATE, D3>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                               tile_out, stream, what, link, DglSqrtMkl{});
    }
    return launch_cvrp_step_sq<UPDATE, D3>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                           tile_out, stream, what, link, DglSqrtIeee{});
}

extern "C" int dgl_cvrp_step(const dgl_cvrp_state_t *st, const int32_t *action, int32_t t, int32_t knn,
                             int32_t depot_knn, int32_t *idx_last_out, int32_t *idx_depot_out,
                             uint8_t *infeasible_out, float *tile_out, dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    DGL_CHECK_ARG(action != nullptr, "action is NULL");
    DGL_CHECK_ARG(t >= 1 && t <= st->N, "t=%d outside [1,N=%d]", t, st->N);
    const bool want = idx_last_out != nullptr;
    DGL_CHECK_ARG(want == (idx_depot_out != nullptr) && want == (infeasible_out != nullptr) && want == (tile_out != nullptr),
                  "idx_last_out, idx_depot_out, infeasible_out and tile_out must all be given or all be NULL");
    DGL_CHECK_ARG(!want || (knn >= 1 && depot_knn >= 1), "knn=%d depot_knn=%d must be >= 1", knn, depot_knn);
    if (!want) knn = depot_knn = 1;
    if (st->demand_in_distance)
        return launch_cvrp_step<true, true>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                            tile_out, stream, "dgl_cvrp_step");
    return launch_cvrp_step<true, false>(st, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                         tile_out, stream, "dgl_cvrp_step");
}

extern "C" int dgl_cvrp_step_beam(const dgl_cvrp_state_t *src, const dgl_cvrp_state_t *dst, const int32_t *prev_k,
                                  int32_t beam, const int32_t *action, int32_t t, int32_t knn, int32_t depot_knn,
                                  int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out,
                                  float *tile_out, int32_t *hist_action, int32_t *hist_parent, dgl_stream_t stream) {
    if (int e = check_cvrp_state(src)) return e;
    if (int e = check_cvrp_state(dst)) return e;
    DGL_CHECK_ARG(action != nullptr && prev_k != nullptr && hist_action != nullptr && hist_parent != nullptr, "NULL argument");
    DGL_CHECK_ARG(beam >= 1 && src->R == dst->R && src->N == dst->N && src->R % beam == 0 && src->xy == dst->xy &&
                      src->demand == dst->demand && src->demand_in_distance == dst->demand_in_distance,
                  "src/dst differ in shape or instance, or R %% beam != 0");
    DGL_CHECK_ARG(src->avg != dst->avg && src->sd != dst->sd && src->visited != dst->visited && src->capacity != dst->capacity &&
                      src->nodes != dst->nodes && src->flags != dst->flags, "src and dst must be different buffers");
    DGL_CHECK_ARG(t >= 2 && t <= dst->N, "t=%d outside [2,N=%d] (the first step has no back-pointer)", t, dst->N);
    const bool want = idx_last_out != nullptr;
    DGL_CHECK_ARG(want == (idx_depot_out != nullptr) && want == (infeasible_out != nullptr) && want == (tile_out != nullptr),
                  "idx_last_out, idx_depot_out, infeasible_out and tile_out must all be given or all be NULL");
    DGL_CHECK_ARG(!want || (knn >= 1 && depot_knn >= 1), "knn=%d depot_knn=%d must be >= 1", knn, depot_knn);
    if (!want) knn = depot_knn = 1;
    const CvrpBeamLink link{src->avg, src->sd, src->capacity, src->visited, src->nodes, src->flags, prev_k, beam,
                            hist_action, hist_parent};
    if (dst->demand_in_distance)
        return launch_cvrp_step<true, true>(dst, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                            tile_out, stream, "dgl_cvrp_step_beam", link);
    return launch_cvrp_step<true, false>(dst, action, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                         tile_out, stream, "dgl_cvrp_step_beam", link);
}

// nodes / flags of steps t_last .. t_first from the history of linked steps: one thread walks one row's back-pointers
__global__ void cvrp_backtrack_kernel(dgl_cvrp_state_t st, const int32_t *__restrict__ hist_action,
                                      const int32_t *__restrict__ hist_parent, int t_first, int t_last) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= st.R) return;
    const int N = st.N;
    int p = r;
    for (int t = t_last; t >= t_first; --t) {
        const size_t e = (size_t)(t - 1) * st.R + p;
        const int act = hist_action[e];
        const bool via = act > N;                               // CVRP/CVRPEnv.py:612-622
        st.nodes[(size_t)r * N + (t - 1)] = via ? act - (N + 1) : act;
        st.flags[(size_t)r * N + (t - 1)] = via ? 1 : 0;
        p = min(max(hist_parent[e], 0), st.R - 1);
    }
}

extern "C" int dgl_cvrp_beam_backtrack(const dgl_cvrp_state_t *st, const int32_t *hist_action, const int32_t *hist_parent,
                                       int32_t t_first, int32_t t_last, dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    DGL_CHECK_ARG(hist_action != nullptr && hist_parent != nullptr, "NULL history");
    DGL_CHECK_ARG(t_first >= 2 && t_last <= st->N && t_first <= t_last, "steps [%d,%d] outside [2,N=%d]", t_first, t_last, st->N);
    hipLaunchKernelGGL(cvrp_backtrack_kernel, dim3((st->R + 63) / 64), dim3(64), 0, (hipStream_t)stream, *st, hist_action,
                       hist_parent, t_first, t_last);
    DGL_CHECK_LAUNCH("dgl_cvrp_beam_backtrack");
    return DGL_OK;
}

extern "C" int dgl_cvrp_knn(const dgl_cvrp_state_t *st, int32_t t, int32_t knn, int32_t depot_knn,
                            int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                            dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    DGL_CHECK_ARG(t >= 1 && t < st->N, "t=%d outside [1,N-1=%d]", t, st->N - 1);
    DGL_CHECK_ARG(idx_last_out && idx_depot_out && infeasible_out && tile_out, "NULL output");
    DGL_CHECK_ARG(knn >= 1 && depot_knn >= 1, "knn=%d depot_knn=%d must be >= 1", knn, depot_knn);
    if (st->demand_in_distance)
        return launch_cvrp_step<false, true>(st, nullptr, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                             tile_out, stream, "dgl_cvrp_knn");
    return launch_cvrp_step<false, false>(st, nullptr, t, knn, depot_knn, idx_last_out, idx_depot_out, infeasible_out,
                                          tile_out, stream, "dgl_cvrp_knn");
}

extern "C" int dgl_cvrp_state_data(const dgl_cvrp_state_t *st, int32_t t, float *data_out, dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    DGL_CHECK_ARG(data_out != nullptr, "data_out is NULL");
    DGL_CHECK_ARG(t >= 1 && t <= st->N, "t=%d outside [1,N=%d]", t, st->N);
    dim3 grid(min((st->N + 256) / 256, 64), st->R);
    auto go = [&](auto sq) {
        using SQ = decltype(sq);
        if (st->demand_in_distance)
            hipLaunchKernelGGL((cvrp_state_data_kernel<true, SQ>), grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, sq);
        else
            hipLaunchKernelGGL((cvrp_state_data_kernel<false, SQ>), grid, dim3(256), 0, (hipStream_t)stream, *st, t, data_out, sq);
    };
    const bool tab = dgl_sqrt_reference_bits();
    if (st->N + 1 <= DGL_DIRECT_MAX_POINTS) {
        if (tab) go(DglSqrtMklT<true>{}); else go(DglSqrtIeeeT<true>{});
    } else {
        if (tab) go(DglSqrtMkl{}); else go(DglSqrtIeee{});
    }
    DGL_CHECK_LAUNCH("dgl_cvrp_state_data");
    return DGL_OK;
}

extern "C" int dgl_cvrp_route_length(const dgl_cvrp_state_t *st, float *length_out, dgl_stream_t stream) {
    if (int e = check_cvrp_state(st)) return e;
    DGL_CHECK_ARG(length_out != nullptr, "length_out is NULL");
    if (dgl_sqrt_reference_bits())
        hipLaunchKernelGGL(cvrp_length_torch_kernel, dim3((st->R + 7) / 8), dim3(256), 0, (hipStream_t)stream, *st, length_out, DglSqrtMkl{});
    else
        hipLaunchKernelGGL(cvrp_length_kernel, dim3((st->R + 3) / 4), dim3(256), 0, (hipStream_t)stream, *st, length_out);
    DGL_CHECK_LAUNCH("dgl_cvrp_route_length");
    return DGL_OK;
}

extern "C" int dgl_cvrp_permute(const dgl_cvrp_state_t *src, const dgl_cvrp_state_t *dst, const int32_t *prev_k,
                                int32_t beam, int32_t t, dgl_stream_t stream) {
    if (int e = check_cvrp_state(src)) return e;
    if (int e = check_cvrp_state(dst)) return e;
    DGL_CHECK_ARG(prev_k != nullptr && beam >= 1, "bad prev_k/beam");
    DGL_CHECK_ARG(src->R == dst->R && src->N == dst->N && src->R % beam == 0, "src/dst shapes differ or R %% beam != 0");
    DGL_CHECK_ARG(src->avg != dst->avg && src->sd != dst->sd && src->nodes != dst->nodes && src->visited != dst->visited &&
                      src->flags != dst->flags && src->capacity != dst->capacity,
                  "src and dst must be distinct buffers");
    hipLaunchKernelGGL(cvrp_permute_kernel, dim3(src->R), dim3(256), 0, (hipStream_t)stream, *src, *dst, prev_k, beam, t);
    DGL_CHECK_LAUNCH("dgl_cvrp_permute");
    return DGL_OK;
}
