// Batched beam expansion (utils/beamsearch.py:52-131): per group of `beam` rows, add the
// cumulative scores to the log-probabilities, take the `beam` best of the beam*A candidates,
// split the flat id into back-pointer and action.
//
// Every candidate gets a unique 64-bit key  ord(value) << 32 | ~flat_index,  so "largest key"
// means highest value and, among equal values, lowest flat index; selection and ordering are
// then plain integer work with no tie handling.
//
// Two entry points: the dense form reads the [G][beam][A] log-probabilities the reference
// materialises; the compact form reads only the <= 32 per-row token probabilities the policy
// kernel produced and treats every other action as the implicit 1e-20 entry
// (TSP/TSPModel.py:337) - beam*32 explicit candidates instead of beam*A.
#include "common.h"

#define BEAM_THREADS 256
#define BEAM_MAX 256
#define DENSE_CAP 2048      // candidate keys kept in LDS by the dense kernel
#define COMPACT_CAP 8192    // explicit candidates the compact kernel sorts in LDS (more: two passes and a gather)

typedef unsigned long long u64;
#define BEAM_FAST 32        // beams up to this size are extracted by repeated wave-wide maxima instead of a sort

extern "C" __device__ unsigned __ockl_wfred_max_u32(unsigned);   // wave-wide maximum (device library), broadcast to all lanes
// the same for a 64-bit key as two 32-bit reductions (one DPP max per step each): high words, then the low words
// of the lanes that hold the largest high word
__device__ __forceinline__ u64 wave_max_key(u64 k) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned mh = __ockl_wfred_max_u32(hi);
    const unsigned ml = __ockl_wfred_max_u32(hi == mh ? lo : 0u);
    return ((u64)mh << 32) | ml;
}

__device__ __forceinline__ unsigned ord_f32(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone: larger float -> larger unsigned
}
__device__ __forceinline__ float unord_f32(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ u64 make_key(float v, unsigned flat) { return ((u64)ord_f32(v) << 32) | (u64)(0xFFFFFFFFu - flat); }
__device__ __forceinline__ float key_value(u64 k) { return unord_f32((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned key_flat(u64 k) { return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu); }

// descending bitonic sort of keys[0..P) in LDS, P a power of two >= BEAM_THREADS; whole block
__device__ void block_sort_desc(u64 *keys, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                int p = i ^ j;
                if (p > i) {
                    u64 a = keys[i], b = keys[p];
                    bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ float dense_value(const float *__restrict__ lp, const float *__restrict__ sc, int A,
                                             int step, unsigned i) {
    const int b = i / A;
    if (step >= 2) return __fadd_rn(lp[i], sc[b]);            // utils/beamsearch.py:66
    return b == 0 ? lp[i] : -1e20f;                            // :68-69
}

__device__ void write_result(const u64 *keys, int beam, int A, float *scores, int32_t *prev_k, int32_t *action) {
    for (int o = threadIdx.x; o < beam; o += blockDim.x) {
        const u64 k = keys[o];
        const unsigned flat = key_flat(k);
        scores[o] = key_value(k);
        prev_k[o] = (int)(flat / (unsigned)A);                 // :89-99
        action[o] = (int)(flat % (unsigned)A);
    }
}

__global__ __launch_bounds__(BEAM_THREADS) void beam_dense_kernel(const float *__restrict__ log_prob, float *scores,
                                                                 int beam, int A, int step, int32_t *prev_k_out,
                                                                 int32_t *action_out) {
    __shared__ u64 keys[DENSE_CAP];
    __shared__ float sc[BEAM_MAX];
    __shared__ int counter;
    __shared__ u64 bound;
    const int g = blockIdx.x, tid = threadIdx.x;
    const unsigned n = (unsigned)beam * (unsigned)A;
    const float *lp = log_prob + (size_t)g * n;
    for (int b = tid; b < beam; b += blockDim.x) sc[b] = scores[(size_t)g * beam + b];
    if (tid == 0) counter = 0;
    __syncthreads();

    // lower bound on the beam-th best key: the beam-th best of the per-thread maxima
    u64 best = 0;
    for (unsigned i = tid; i < n; i += BEAM_THREADS) {
        u64 k = make_key(dense_value(lp, sc, A, step, i), i);
        best = k > best ? k : best;
    }
    keys[tid] = best;
    __syncthreads();
    block_sort_desc(keys, BEAM_THREADS);
    if (tid == 0) bound = keys[min(beam, BEAM_THREADS) - 1];
    __syncthreads();
    u64 lb = bound;
    __syncthreads();

    // gather every key >= lb; normally a few hundred at most
    for (unsigned i = tid; i < n; i += BEAM_THREADS) {
        u64 k = make_key(dense_value(lp, sc, A, step, i), i);
        if (k >= lb) {
            int pos = atomicAdd(&counter, 1);
            if (pos < DENSE_CAP) keys[pos] = k;
        }
    }
    __syncthreads();
    int c = counter;
    __syncthreads();
    if (c > DENSE_CAP) {
        // pathological key distribution: find the beam-th largest key exactly by bisection on its
        // bits (64 counting passes), then gather the keys >= it (exactly `beam` of them)
        u64 res = 0;
        for (int bit = 63; bit >= 0; --bit) {
            u64 cand = res | (1ull << bit);
            if (tid == 0) counter = 0;
            __syncthreads();
            int local = 0;
            for (unsigned i = tid; i < n; i += BEAM_THREADS)
                local += make_key(dense_value(lp, sc, A, step, i), i) >= cand;
            atomicAdd(&counter, local);
            __syncthreads();
            if (counter >= beam) res = cand;
            __syncthreads();
        }
        if (tid == 0) counter = 0;
        __syncthreads();
        for (unsigned i = tid; i < n; i += BEAM_THREADS) {
            u64 k = make_key(dense_value(lp, sc, A, step, i), i);
            if (k >= res) keys[atomicAdd(&counter, 1)] = k;
        }
        __syncthreads();
        c = counter;
        __syncthreads();
    }
    int P = BEAM_THREADS;
    while (P < c) P <<= 1;
    for (int i = c + tid; i < P; i += BEAM_THREADS) keys[i] = 0;
    __syncthreads();
    block_sort_desc(keys, P);
    write_result(keys, beam, A, scores + (size_t)g * beam, prev_k_out + (size_t)g * beam, action_out + (size_t)g * beam);
}

// action id of token `tok` of a row, or -1 for an anchor token
__device__ __forceinline__ int token_action(const dgl_policy_io_t &io, int r, int tok) {
    if (tok == io.k_last || tok == io.n - 1) return -1;
    if (io.problem == DGL_PROBLEM_TSP || tok < io.k_last) return io.idx_last[(size_t)r * io.idx_stride + tok];
    return io.idx_depot[(size_t)r * io.idx_dep_stride + (tok - io.k_last - 1)] + io.N + 1;
}

// key of explicit candidate e = (beam b, token) of group g, 0 for an anchor token
__device__ __forceinline__ u64 compact_key(const dgl_policy_io_t &io, const float *__restrict__ prob, const float *sc, int g,
                                           int beam, int A, int step, int e) {
    const int n = io.n;
    const int b = e / n, tok = e % n, r = g * beam + b;
    // (requested before the action is known: two independent loads instead of two round trips)
    const float pv = prob[(size_t)r * (io.out_stride > 0 ? io.out_stride : DGL_MAX_TOKENS) + tok];
    const int act = token_action(io, r, tok);
    if (act < 0) return 0;
    const float lpv = logf(pv);
    const float v = step >= 2 ? __fadd_rn(lpv, sc[b]) : (b == 0 ? lpv : -1e20f);
    return make_key(v, (unsigned)b * (unsigned)A + (unsigned)act);
}

// P: LDS slots for explicit candidates (a power of two).  beam * n <= P: every explicit candidate is sorted in LDS.
// beam * n > P (wide beams over rows of many tokens, e.g. beam 100 x 202 tokens, CVRP/train.py:70,130-131): two
// passes over the candidates as in the dense kernel - the beam-th best of the per-thread maxima bounds the beam-th
// best key from below, the keys at or above it (a few hundred) are gathered and sorted.
__global__ __launch_bounds__(BEAM_THREADS) void beam_compact_kernel(dgl_policy_io_t io, const float *__restrict__ prob,
                                                                   float *scores, int beam, int A, int step,
                                                                   int32_t *prev_k_out, int32_t *action_out, int P) {
    extern __shared__ u64 ckeys[];          // [P] explicit candidates, then [beam] merged result
    __shared__ float sc[BEAM_MAX];
    __shared__ int counter;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int n = io.n;
    for (int b = tid; b < beam; b += blockDim.x) sc[b] = scores[(size_t)g * beam + b];
    if (tid == 0) counter = 0;
    __syncthreads();
    const int ncand = beam * n;
    int n_explicit = P;                     // sorted explicit keys available below
    if (ncand <= P) {
        for (int e = tid; e < P; e += BEAM_THREADS) ckeys[e] = e < ncand ? compact_key(io, prob, sc, g, beam, A, step, e) : 0;
        __syncthreads();
    } else {
        u64 best = 0;
        for (int e = tid; e < ncand; e += BEAM_THREADS) {
            const u64 k = compact_key(io, prob, sc, g, beam, A, step, e);
            best = k > best ? k : best;
        }
        ckeys[tid] = best;
        __syncthreads();
        block_sort_desc(ckeys, BEAM_THREADS);
        u64 lb = ckeys[min(beam, BEAM_THREADS) - 1];
        __syncthreads();
        for (int e = tid; e < ncand; e += BEAM_THREADS) {
            const u64 k = compact_key(io, prob, sc, g, beam, A, step, e);
            if (k >= lb && k != 0) {
                const int pos = atomicAdd(&counter, 1);
                if (pos < P) ckeys[pos] = k;
            }
        }
        __syncthreads();
        int c = counter;
        __syncthreads();
        if (c > P) {
            // more keys at or above the bound than slots (many equal values): the beam-th largest key exactly, by
            // bisection on its bits, then the keys at or above it (exactly `beam` of them: keys are unique)
            u64 res = 0;
            for (int bit = 63; bit >= 0; --bit) {
                const u64 cand = res | (1ull << bit);
                if (tid == 0) counter = 0;
                __syncthreads();
                int local = 0;
                for (int e = tid; e < ncand; e += BEAM_THREADS) local += compact_key(io, prob, sc, g, beam, A, step, e) >= cand;
                atomicAdd(&counter, local);
                __syncthreads();
                if (counter >= beam) res = cand;
                __syncthreads();
            }
            if (tid == 0) counter = 0;
            __syncthreads();
            for (int e = tid; e < ncand; e += BEAM_THREADS) {
                const u64 k = compact_key(io, prob, sc, g, beam, A, step, e);
                if (k >= res && k != 0) {
                    const int pos = atomicAdd(&counter, 1);
                    if (pos < P) ckeys[pos] = k;
                }
            }
            __syncthreads();
            c = min(counter, P);
            __syncthreads();
        }
        n_explicit = BEAM_THREADS;
        while (n_explicit < c) n_explicit <<= 1;
        for (int i = c + tid; i < n_explicit; i += BEAM_THREADS) ckeys[i] = 0;
        __syncthreads();
    }
    // The `beam` largest explicit keys in descending order.  Few beams over at most 512 candidates: one wave holds
    // all keys in registers and extracts the maximum `beam` times (keys are unique); otherwise a block-wide sort.
    __shared__ u64 top[BEAM_FAST];
    const bool fast = beam <= BEAM_FAST && n_explicit <= 512;
    if (fast) {
        if (tid < 64) {
            u64 k[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) k[q] = (tid + 64 * q < n_explicit) ? ckeys[tid + 64 * q] : 0;
            // each lane sorts its eight keys once (descending, 19 compare-exchanges); its largest is then k[0], and
            // the lane that owned an extracted maximum shifts its list up
#define BEAM_CE(i, j) { const u64 a_ = k[i], b_ = k[j]; const bool sw_ = a_ < b_; k[i] = sw_ ? b_ : a_; k[j] = sw_ ? a_ : b_; }
            BEAM_CE(0, 1) BEAM_CE(2, 3) BEAM_CE(4, 5) BEAM_CE(6, 7)
            BEAM_CE(0, 2) BEAM_CE(1, 3) BEAM_CE(4, 6) BEAM_CE(5, 7)
            BEAM_CE(1, 2) BEAM_CE(5, 6)
            BEAM_CE(0, 4) BEAM_CE(1, 5) BEAM_CE(2, 6) BEAM_CE(3, 7)
            BEAM_CE(2, 4) BEAM_CE(3, 5)
            BEAM_CE(1, 2) BEAM_CE(3, 4) BEAM_CE(5, 6)
#undef BEAM_CE
            for (int o = 0; o < beam; ++o) {
                const u64 m = wave_max_key(k[0]);
                if (tid == 0) top[o] = m;
                const bool mine = k[0] == m;            // keys are unique (0 = none left: shifting zeros is harmless)
#pragma unroll
                for (int q = 0; q < 7; ++q) k[q] = mine ? k[q + 1] : k[q];
                k[7] = mine ? 0 : k[7];
            }
        }
        __syncthreads();
    } else {
        block_sort_desc(ckeys, n_explicit);
    }
    const u64 *sorted = fast ? top : ckeys;
    const int n_sorted = fast ? beam : n_explicit;

    // implicit candidates: every action of beam b that is not one of its tokens has probability 1e-20
    // (value log(1e-20) + score_b, or -1e20 for b > 0 at step 1).  They only matter when fewer than
    // `beam` explicit candidates beat the best of them; one thread then merges the two sorted streams.
    // best implicit group: highest value, then lowest beam index (every thread computes the same pair)
    const float lfloor = logf(1e-20f);
    int bstar = 0;
    float fbest = -INFINITY;
    for (int b = 0; b < beam; ++b) {
        const float v = step >= 2 ? __fadd_rn(lfloor, sc[b]) : (b == 0 ? lfloor : -1e20f);
        if (v > fbest) { fbest = v; bstar = b; }
    }
    // the usual case: the beam-th explicit candidate has a higher value than any implicit one
    if (n_sorted >= beam && (unsigned)(sorted[beam - 1] >> 32) > ord_f32(fbest)) {
        write_result(sorted, beam, A, scores + (size_t)g * beam, prev_k_out + (size_t)g * beam, action_out + (size_t)g * beam);
        return;
    }
    u64 *result = ckeys + P;
    if (tid == 0) {
        const int rstar = g * beam + bstar;
        int a_next = 0, tok_scan = 0;        // walk beam bstar's actions in ascending order, skipping its tokens
        auto next_floor_key = [&]() -> u64 {
            for (;;) {
                if (a_next >= A) return 0;
                // token actions of a row ascend with the token index (anchors excluded)
                while (tok_scan < n && (token_action(io, rstar, tok_scan) < a_next)) ++tok_scan;
                if (tok_scan < n && token_action(io, rstar, tok_scan) == a_next) { ++a_next; continue; }
                return make_key(fbest, (unsigned)bstar * (unsigned)A + (unsigned)a_next);
            }
        };
        u64 fk = next_floor_key();
        int ci = 0;
        for (int o = 0; o < beam; ++o) {
            const u64 ck = ci < n_sorted ? sorted[ci] : 0;
            if (ck >= fk) { result[o] = ck; ++ci; }
            else { result[o] = fk; ++a_next; fk = next_floor_key(); }
        }
    }
    __syncthreads();
    write_result(result, beam, A, scores + (size_t)g * beam, prev_k_out + (size_t)g * beam, action_out + (size_t)g * beam);
}

extern "C" int dgl_beam_advance(const float *log_prob, float *scores, int32_t G, int32_t beam, int32_t A,
                                int32_t step, int32_t *prev_k_out, int32_t *action_out, dgl_stream_t stream) {
    DGL_CHECK_ARG(log_prob && scores && prev_k_out && action_out, "NULL argument");
    DGL_CHECK_ARG(G > 0 && beam >= 1 && A >= 1 && step >= 1, "bad sizes G=%d beam=%d A=%d step=%d", G, beam, A, step);
    DGL_CHECK_ARG(beam <= BEAM_MAX, "beam=%d exceeds the supported maximum %d", beam, BEAM_MAX);
    DGL_CHECK_ARG((long long)beam * A >= beam, "beam*A must be at least beam");
    DGL_CHECK_ARG((long long)beam * A < 0xFFFFFFFFll, "beam*A too large");
    hipLaunchKernelGGL(beam_dense_kernel, dim3(G), dim3(BEAM_THREADS), 0, (hipStream_t)stream, log_prob, scores, beam, A,
                       step, prev_k_out, action_out);
    DGL_CHECK_LAUNCH("dgl_beam_advance");
    return DGL_OK;
}

extern "C" int dgl_beam_advance_compact(const dgl_policy_io_t *io, const float *prob, float *scores, int32_t G,
                                        int32_t beam, int32_t A, int32_t step, int32_t *prev_k_out,
                                        int32_t *action_out, dgl_stream_t stream) {
    DGL_CHECK_ARG(io && prob && scores && prev_k_out && action_out, "NULL argument");
    DGL_CHECK_ARG(G > 0 && beam >= 1 && step >= 1, "bad sizes G=%d beam=%d step=%d", G, beam, step);
    DGL_CHECK_ARG(beam <= BEAM_MAX, "beam=%d exceeds the supported maximum %d", beam, BEAM_MAX);
    DGL_CHECK_ARG(io->R == G * beam, "io->R=%d != G*beam=%d", io->R, G * beam);
    DGL_CHECK_ARG(io->n >= 2 && io->n <= (io->out_stride > 0 ? io->out_stride : DGL_MAX_TOKENS) && io->idx_last != nullptr,
                  "bad policy io");
    const int need = io->problem == DGL_PROBLEM_TSP ? io->N : 2 * (io->N + 1);
    DGL_CHECK_ARG(A == need, "A=%d but the action space is %d", A, need);
    DGL_CHECK_ARG(A >= beam, "A=%d < beam=%d", A, beam);
    int P = BEAM_THREADS;
    while (P < beam * io->n && P < COMPACT_CAP) P <<= 1;       // more candidates than slots: the kernel's two-pass form
    size_t lds = (size_t)(P + beam) * sizeof(u64);
    if (lds > 48 * 1024)
        dgl_max_dynamic_lds((const void *)beam_compact_kernel, (int)lds);
    hipLaunchKernelGGL(beam_compact_kernel, dim3(G), dim3(BEAM_THREADS), lds, (hipStream_t)stream, *io, prob, scores,
                       beam, A, step, prev_k_out, action_out, P);
    DGL_CHECK_LAUNCH("dgl_beam_advance_compact");
    return DGL_OK;
}
