// Wave-level exact k-smallest selection over keys held in LDS.
//
// One wave64 owns one row.  keys[0..n) are non-negative binary32 values; entries that
// must never be picked hold +inf.  The k smallest are chosen, the lowest index winning
// among equal keys, and are emitted in ascending index order, so the result needs no
// sort (the reference sorts its argsort prefix by index, TSP/TSPModel.py:89-91).
#pragma once
#include "common.h"

// ordering point between LDS writes of some lanes and reads by other lanes of the SAME wave
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// k-th smallest (1-based k, 1 <= k <= number of finite keys) of keys[0..n): bisection on the
// key's bit pattern with ballot/popcount counting (31 passes over the keys).
__device__ __forceinline__ float wave_kth_smallest(const float *keys, int n, int k, float *scratch) {
    const int lane = threadIdx.x & 63;
    unsigned res = 0;
    for (int bit = 30; bit >= 0; --bit) {
        unsigned cand = res | (1u << bit);
        if (cand > 0x7F800000u) continue;
        float cf = __uint_as_float(cand);
        int cnt = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            int j = j0 + lane;
            bool p = j < n && keys[j] < cf;
            cnt += __popcll(__ballot(p));
        }
        if (cnt < k) res = cand;
    }
    return __uint_as_float(res);
}

// Emit the k selected indices in ascending order: emit(j, rank) is called by the lane
// that owns j (j % 64 == lane), rank in [0,k).
template <class Emit>
__device__ __forceinline__ void wave_emit_selected(const float *keys, int n, int k, float T, Emit emit) {
    const int lane = threadIdx.x & 63;
    int less = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        int j = j0 + lane;
        less += __popcll(__ballot(j < n && keys[j] < T));
    }
    const int need_eq = k - less;
    int sel_run = 0, eq_run = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        int j = j0 + lane;
        float key = j < n ? keys[j] : INFINITY;
        bool lt = key < T;
        bool eq = (key == T) && (j < n);
        unsigned long long meq = __ballot(eq);
        bool pick = lt || (eq && (eq_run + lane_rank(meq) < need_eq));
        unsigned long long mp = __ballot(pick);
        if (pick) emit(j, sel_run + lane_rank(mp));
        sel_run += __popcll(mp);
        eq_run += __popcll(meq);
    }
}

// #{lanes : v <= x}, x given as the bit pattern of a non-negative float (wave-uniform)
__device__ __forceinline__ int wave_count_le(float v, unsigned x) { return __popcll(__ballot(v <= __uint_as_float(x))); }

// The k nearest of a row whose keys are SQUARED distances (the radicands of dgl_sqrt), ranked as
// the reference ranks them: by the rounded square root, lowest index first among equal roots
// (several neighbouring radicands share one root).  All searches are bisections on the bit
// pattern of non-negative floats, driven by ballot counts: wave-uniform control flow, one vector
// compare per probe, no cross-lane data movement.
//
// Step 1, wave_candidates: a bound B with at least k lane minima <= B (so at least k keys <= B),
// found by a bisection that stops as soon as the count is within [k, k+2] (with m of the 64 lane
// minima under B about -64 ln(1 - m/64) keys are: 40 for m = 30); every radicand whose root can
// tie with that of a key <= B lies within 2 ulps above B, so B is raised by 4 ulps; the keys <= B
// are compacted in index order, one per lane.  Returns their number c (lane < c holds key and
// index j).  More than 64 keys under the bound: one more pass with the tighter bound that the first 64
// candidates give (wave_candidates_again).  -1 when that fails too or k > 64 (then use
// wave_select_general).  scratch: 128 words.  The pieces (the bound, the compaction of a node range, the
// concatenation of per-wave lists) are separate so that the waves sharing a row can split the work.
//
// the bound of step 1 as a bit pattern (k <= 64)
__device__ __forceinline__ unsigned wave_candidate_bound(float lane_min, int k, unsigned ulps = 4u) {
    const unsigned FMAX = 0x7F7FFFFFu;
    unsigned hi = FMAX;                     // fewer than k finite lane minima: every finite key is a candidate
    if (wave_count_le(lane_min, FMAX) >= k) {
        unsigned lo = 0;
        if (wave_count_le(lane_min, 0u) >= k) hi = 0;
        while (hi - lo > 1) {               // invariant: count(lo) < k <= count(hi)
            const unsigned mid = lo + ((hi - lo) >> 1);
            const int cnt = wave_count_le(lane_min, mid);
            if (cnt >= k) {
                hi = mid;
                if (cnt <= k + 2) break;
            } else {
                lo = mid;
            }
        }
        hi = min(hi + ulps, FMAX);
    }
    return hi;
}

// keys[j] <= bound for j in [j_lo, j_hi) (j_lo a multiple of 64), compacted in index order: the first `cap` go to
// out_key / out_j; returns how many there are (possibly more than cap)
__device__ __forceinline__ int wave_compact(const float *keys, int j_lo, int j_hi, float bound, float *out_key, int *out_j,
                                            int cap = 64) {
    const int lane = threadIdx.x & 63;
    int c = 0;
    for (int j0 = j_lo; j0 < j_hi; j0 += 256) {
        float key[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // four LDS reads in flight
            const int j = j0 + 64 * q + lane;
            key[q] = j < j_hi ? keys[j] : INFINITY;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool p = key[q] <= bound;
            const unsigned long long m = __ballot(p);
            const int pos = c + lane_rank(m);
            if (p && pos < cap) {
                out_key[pos] = key[q];
                out_j[pos] = j0 + 64 * q + lane;
            }
            c += __popcll(m);
        }
    }
    return c;
}

// More than 64 keys lay under the first bound (about one row in some thousand: ~40 keys lie under the 30th smallest
// of 64 lane minima, with a long tail).  first_key: the first 64 of them, one per lane: k of the row's keys lie
// under their k-th smallest too, so it bounds the row's k-th smallest from above, and all but a few of the surplus
// keys lie above it.  One more pass over the row's keys with that bound.
__device__ __forceinline__ int wave_candidates_again(const float *keys, int n, int k, float first_key, float bound,
                                                     float *scratch, float *key_out, int *j_out, unsigned ulps = 4u) {
    const int lane = threadIdx.x & 63;
    bound = fminf(bound, __uint_as_float(wave_candidate_bound(first_key, k, ulps)));
    int *cand_j = reinterpret_cast<int *>(scratch) + 64;
    wave_lds_sync();                        // every lane has read its first_key (it may come from scratch)
    const int c = wave_compact(keys, 0, n, bound, scratch, cand_j);
    if (c > 64) return -1;
    wave_lds_sync();
    *key_out = lane < c ? scratch[lane] : INFINITY;
    *j_out = lane < c ? cand_j[lane] : 0;
    return c;
}

__device__ __forceinline__ int wave_candidates(const float *keys, int n, int k, float lane_min, float *scratch,
                                               float *key_out, int *j_out, unsigned ulps = 4u) {
    const int lane = threadIdx.x & 63;
    if (k > 64) return -1;
    const float bound = __uint_as_float(wave_candidate_bound(lane_min, k, ulps));
    int *cand_j = reinterpret_cast<int *>(scratch) + 64;
    const int c = wave_compact(keys, 0, n, bound, scratch, cand_j);
    wave_lds_sync();
    if (c > 64) return wave_candidates_again(keys, n, k, scratch[lane], bound, scratch, key_out, j_out, ulps);
    *key_out = lane < c ? scratch[lane] : INFINITY;
    *j_out = lane < c ? cand_j[lane] : 0;
    return c;
}

// The same with the compaction shared by the W waves that scanned a row: wave w has compacted its own node range
// into list w (wave_compact; 128 words per list: 64 keys, 64 indices) and left its count in counts[w]; the ranges
// ascend with w, so the lists concatenate in index order.  Called by one wave after the workgroup barrier.
__device__ __forceinline__ int wave_candidates_from_lists(const float *lists, const int *counts, int W, const float *keys,
                                                          int n, int k, float bound, float *scratch, float *key_out,
                                                          int *j_out, unsigned ulps = 4u) {
    const int lane = threadIdx.x & 63;
    int c = 0;
    float key = INFINITY;
    int j = 0;
    for (int w = 0; w < W; ++w) {
        const int cw = counts[w];
        if (lane >= c && lane < c + cw) {       // a list holds the first 64 of its wave's cw candidates
            key = lists[w * 128 + (lane - c)];
            j = reinterpret_cast<const int *>(lists)[w * 128 + 64 + (lane - c)];
        }
        c += cw;
    }
    if (c > 64) return wave_candidates_again(keys, n, k, key, bound, scratch, key_out, j_out, ulps);
    *key_out = key;
    *j_out = j;
    return c;
}

// Step 2, wave_pick_k: root = dgl_sqrt(key) of this lane's candidate (+inf for lane >= c).  Picks
// the k smallest roots, the lower lane (= lower index) first among equal ones; returns whether
// this lane's candidate is picked and its rank among the picks (ascending index).  The bisection
// ends early when a probe separates exactly k candidates.
__device__ __forceinline__ bool wave_pick_k(float root, int c, int k, int *rank_out) {
    const int lane = threadIdx.x & 63;
    bool pick = lane < c;
    if (c > k) {
        unsigned thi = 0x7F7FFFFFu, tlo = 0;                              // count(thi) = c > k
        bool split = false;                                               // a probe with exactly k roots below it
        if (wave_count_le(root, 0u) >= k) thi = 0;
        while (thi - tlo > 1) {                                           // invariant: count(tlo) < k <= count(thi)
            const unsigned mid = tlo + ((thi - tlo) >> 1);
            const int cnt = wave_count_le(root, mid);
            if (cnt >= k) thi = mid; else tlo = mid;
            if (cnt == k) { split = true; break; }
        }
        const float T = __uint_as_float(thi);
        if (split) {
            pick = root <= T;
        } else {                                                           // T = k-th smallest root, tied
            const bool lt = root < T, eq = root == T;
            const unsigned long long meq = __ballot(eq);
            const int need_eq = k - __popcll(__ballot(lt));
            pick = lt || (eq && lane_rank(meq) < need_eq);
        }
    }
    *rank_out = lane_rank(__ballot(pick));
    return pick;
}

// Wide neighbourhoods, 64 < k <= 256 (the reference trains with knn = depot_knn = 100, CVRP/train.py:130-131), for rows
// scanned by W waves that left the minima of their 64 W scan classes (class = (wave, lane) of the scanning thread) in
// LDS, class_min[64 w + lane].  Every class minimum is one of the row's keys, so the k-th smallest of them bounds the
// row's k-th smallest key from above, and about -C ln(1 - k/C) keys lie under it (C = 64 W: 111 for k = 100, C = 512).
// The keys under the bound (raised by 4 ulps, see wave_candidates) are compacted in index order into `list` (512 words:
// 256 keys, 256 indices), four per lane (slot 64 q + lane); their roots are ranked as in wave_pick_k.  emit(j, rank) is
// called for the k picks, rank ascending with the node index.  Returns false - nothing emitted - when the class minima
// cannot bound k keys or more than 256 keys lie under the bound: the caller then takes the general path.
template <int W, class SQ, class Emit>
__device__ __forceinline__ bool wave_select_wide(const SQ &sq, const float *keys, int n, int k, const float *class_min,
                                                 float *list, Emit emit) {
    const int lane = threadIdx.x & 63;
    const unsigned FMAX = 0x7F7FFFFFu;
    float m[W];
#pragma unroll
    for (int w = 0; w < W; ++w) m[w] = class_min[64 * w + lane];
    auto classes_le = [&](unsigned x) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) c += wave_count_le(m[w], x);
        return c;
    };
    if (classes_le(FMAX) < k) return false;
    unsigned hi = FMAX, lo = 0;
    if (classes_le(0u) >= k) hi = 0;
    while (hi - lo > 1) {                   // invariant: count(lo) < k <= count(hi)
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int cnt = classes_le(mid);
        if (cnt >= k) {
            hi = mid;
            if (cnt <= k + 4) break;
        } else {
            lo = mid;
        }
    }
    const float bound = __uint_as_float(min(hi + SQ::ULPS, FMAX));
    int *list_j = reinterpret_cast<int *>(list) + 256;
    const int c = wave_compact(keys, 0, n, bound, list, list_j, 256);
    if (c > 256) return false;
    wave_lds_sync();
    float root[4];
    int cj[4];
    bool pick[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = 64 * q + lane;
        root[q] = s < c ? sq.dist(list[s]) : INFINITY;
        cj[q] = s < c ? list_j[s] : 0;
        pick[q] = s < c;
    }
    if (c > k) {
        auto roots_le = [&](unsigned x) {
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt += wave_count_le(root[q], x);
            return cnt;
        };
        unsigned thi = FMAX, tlo = 0;                                       // count(thi) = c > k
        bool split = false;                                                 // a probe with exactly k roots at or below it
        if (roots_le(0u) >= k) thi = 0;
        while (thi - tlo > 1) {                                             // invariant: count(tlo) < k <= count(thi)
            const unsigned mid = tlo + ((thi - tlo) >> 1);
            const int cnt = roots_le(mid);
            if (cnt >= k) thi = mid; else tlo = mid;
            if (cnt == k) { split = true; break; }
        }
        const float T = __uint_as_float(thi);
        if (split) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pick[q] = root[q] <= T;
        } else {                                                            // T = the k-th smallest root, tied
            int less = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) less += __popcll(__ballot(root[q] < T));
            const int need_eq = k - less;
            int eq_run = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                   // slot order = index order
                const bool eq = root[q] == T;
                const unsigned long long meq = __ballot(eq);
                pick[q] = root[q] < T || (eq && eq_run + lane_rank(meq) < need_eq);
                eq_run += __popcll(meq);
            }
        }
    }
    int run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long mp = __ballot(pick[q]);
        if (pick[q]) emit(cj[q], run + lane_rank(mp));
        run += __popcll(mp);
    }
    return true;
}

// General path: all keys are rooted in place, then bisection over all keys + the two ballot scans
// of wave_emit_selected.  emit(j, rank) is called by the lane that owns j (j % 64 == lane).
template <class SQ, class Emit>
__device__ __forceinline__ void wave_select_general(const SQ &sq, float *keys, int n, int k, float *scratch, Emit emit) {
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < n; j += 64) keys[j] = sq.dist(keys[j]);    // each lane roots the keys it wrote
    wave_lds_sync();
    const float T = wave_kth_smallest(keys, n, k, scratch);
    wave_emit_selected(keys, n, k, T, emit);
}

// Per-column min-max normalisation of a token tile held in LDS ([n][DGL_TILE_STRIDE] floats, data
// in columns 0..ncol-1), written to global memory with the same stride:
//   out = range != 0 ? (v - min) / range : 0      (TSP/TSPModel.py:158-172)
// Column ncol of the output receives `pad_first`, the remaining padding columns 0, so the consumer
// can load whole strides.  Lane 12 g + c owns column c of the tokens g, g+5, g+10, ...: one store
// instruction covers 5 whole tokens (240 contiguous bytes) and min, range and 1/range are fixed
// per lane.  scratch: 128 words.
__device__ __forceinline__ void wave_minmax_norm_store(const float *tile, int n, int ncol, float pad_first,
                                                       float *out, float *scratch) {
    static_assert(DGL_TILE_STRIDE == 12, "lane mapping assumes 5 tokens x 12 columns per wave instruction");
    const int lane = threadIdx.x & 63;
    const int g = lane / DGL_TILE_STRIDE, c = lane - DGL_TILE_STRIDE * g;
    const bool data = (g < 5) && (c < ncol);
    float lo = INFINITY, hi = -INFINITY;
    if (data) {
        for (int tk = g; tk < n; tk += 5) {
            const float v = tile[tk * DGL_TILE_STRIDE + c];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    scratch[lane] = lo;
    scratch[64 + lane] = hi;
    wave_lds_sync();
    if (g < 5) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            lo = fminf(lo, scratch[DGL_TILE_STRIDE * q + c]);
            hi = fmaxf(hi, scratch[64 + DGL_TILE_STRIDE * q + c]);
        }
        const float rng = __fsub_rn(hi, lo);
        const DglConstDiv dv = dgl_const_div(rng);
        const float pad = (c == ncol) ? pad_first : 0.0f;
        for (int tk = g; tk < n; tk += 5) {
            float o = pad;
            if (data) o = (rng != 0.0f) ? dgl_div(__fsub_rn(tile[tk * DGL_TILE_STRIDE + c], lo), dv) : 0.0f;
            out[tk * DGL_TILE_STRIDE + c] = o;
        }
    }
}
