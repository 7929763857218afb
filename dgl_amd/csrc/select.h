// Wave-level exact k-smallest selection over keys held in LDS.
//
// One wave64 owns one row.  keys[0..n) are non-negative binary32 values; entries that
// must never be picked hold +inf.  The k smallest are chosen, the lowest index winning
// among equal keys, and are emitted in ascending index order, so the result needs no
// sort (the reference sorts its argsort prefix by index, TSP/TSPModel.py:89-91).
//
// Threshold search: T = k-th smallest key value.
//   fast path  - each lane tracks the minimum of the keys it wrote; the k-th smallest of
//                the 64 lane minima bounds T from above, the (few) keys below that bound
//                are compacted to one per lane and sorted with an in-register bitonic
//                network;
//   general    - bisection on the key's bit pattern with ballot/popcount counting
//                (31 passes over the keys), used when more than 64 keys survive the bound
//                or k > 64.
#pragma once
#include "common.h"

// ordering point between LDS writes of some lanes and reads by other lanes of the SAME wave
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ascending bitonic sort of one float per lane across the wave
__device__ __forceinline__ float wave_sort64(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            float o = __shfl_xor(v, j, 64);
            bool up = (lane & k) == 0;
            bool lower = (lane & j) == 0;
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return v;
}

// k-th smallest (1-based k, 1 <= k <= number of finite keys) of keys[0..n).
// lane_min: minimum over the keys this lane itself stored (keys j with j % 64 == lane).
// scratch: 64 floats of wave-private LDS.
__device__ __forceinline__ float wave_kth_smallest(const float *keys, int n, int k, float lane_min,
                                                   float *scratch) {
    const int lane = threadIdx.x & 63;
    if (k <= 64) {
        float s = wave_sort64(lane_min);
        float bound = __shfl(s, k - 1, 64);
        if (bound < INFINITY) {
            int c = 0;
            for (int j0 = 0; j0 < n; j0 += 64) {
                int j = j0 + lane;
                float key = j < n ? keys[j] : INFINITY;
                bool p = key <= bound;
                unsigned long long m = __ballot(p);
                int pos = c + lane_rank(m);
                if (p && pos < 64) scratch[pos] = key;
                c += __popcll(m);
            }
            if (c <= 64) {
                wave_lds_sync();
                float v = lane < c ? scratch[lane] : INFINITY;
                v = wave_sort64(v);
                return __shfl(v, k - 1, 64);
            }
        }
    }
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

// The k nearest of a row whose keys are SQUARED distances (the radicands of dgl_sqrt), ranked as
// the reference ranks them: by the rounded square root, lowest index first among equal roots
// (several neighbouring radicands share one root).  emit(j, rank) is called once per selected node,
// rank ascending with j, by an arbitrary lane.  keys[] may be overwritten.  scratch: 128 words.
//   fast path - the k-th smallest lane minimum bounds the k-th smallest radicand from above; every
//               radicand whose root can tie with it lies within a few ulps of that bound, so the
//               bound is raised by 4 ulps, the few keys below it are compacted (in index order)
//               to one per lane, rooted, and ranked with one in-register sort and two ballots;
//   general   - all keys are rooted in place, then bisection + the two ballot scans above.
template <class Emit>
__device__ __forceinline__ void wave_select_nearest(float *keys, int n, int k, float lane_min, float *scratch,
                                                    Emit emit) {
    const int lane = threadIdx.x & 63;
    if (k <= 64) {
        float bound = __shfl(wave_sort64(lane_min), k - 1, 64);
        bound = __uint_as_float(__float_as_uint(bound) + 4u);      // +inf turns into NaN: no fast path
        if (bound < INFINITY) {
            int *cand_j = reinterpret_cast<int *>(scratch) + 64;
            int c = 0;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                const float key = j < n ? keys[j] : INFINITY;
                const bool p = key <= bound;
                const unsigned long long m = __ballot(p);
                const int pos = c + lane_rank(m);
                if (p && pos < 64) {
                    scratch[pos] = key;
                    cand_j[pos] = j;
                }
                c += __popcll(m);
            }
            if (c <= 64) {
                wave_lds_sync();
                const float d = lane < c ? dgl_sqrt(scratch[lane]) : INFINITY;
                const int j = cand_j[lane];
                const float T = __shfl(wave_sort64(d), k - 1, 64);
                const bool lt = d < T, eq = d == T;
                const unsigned long long meq = __ballot(eq);
                const int need_eq = k - __popcll(__ballot(lt));
                const bool pick = lt || (eq && lane_rank(meq) < need_eq);
                const unsigned long long mp = __ballot(pick);
                if (pick) emit(j, lane_rank(mp));
                return;
            }
        }
    }
    for (int j = lane; j < n; j += 64) keys[j] = dgl_sqrt(keys[j]);    // each lane roots the keys it wrote
    wave_lds_sync();
    const float T = wave_kth_smallest(keys, n, k, INFINITY, scratch);
    wave_emit_selected(keys, n, k, T, emit);
}

// Per-column min-max normalisation of a token tile held in LDS ([n][stride] floats,
// columns 0..ncol-1), written to global memory with the same stride:
//   out = range != 0 ? (v - min) / range : 0      (TSP/TSPModel.py:158-172)
__device__ __forceinline__ void wave_minmax_norm_store(const float *tile, int n, int ncol, int stride,
                                                       float *out) {
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < ncol; ++c) {
        float lo = INFINITY, hi = -INFINITY;
        for (int tk = lane; tk < n; tk += 64) {
            float v = tile[tk * stride + c];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
        lo = wave_min_f32(lo);
        hi = wave_max_f32(hi);
        float rng = __fsub_rn(hi, lo);
        for (int tk = lane; tk < n; tk += 64) {
            float v = tile[tk * stride + c];
            out[tk * stride + c] = (rng != 0.0f) ? __fdiv_rn(__fsub_rn(v, lo), rng) : 0.0f;
        }
    }
}
