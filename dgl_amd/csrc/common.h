// Shared device/host helpers for the gfx950 kernels.  Compiled with -ffp-contract=off:
// every fused multiply-add is written explicitly, so float results do not depend on the
// compiler's contraction choices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dgl_amd.h"

#define DGL_WAVE 64

// host-side error plumbing (api.cpp)
int dgl_set_error(int code, const char *fmt, ...);
bool dgl_sqrt_reference_bits();                               // the current device's square-root model is DGL_SQRT_MKL_AVX512 (api.cpp)
void dgl_max_dynamic_lds(const void *kernel, int bytes);      // hipFuncAttributeMaxDynamicSharedMemorySize, once per device
#define DGL_CHECK_ARG(cond, ...)                                           \
    do {                                                                   \
        if (!(cond)) return dgl_set_error(DGL_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)
#define DGL_CHECK_LAUNCH(what)                                             \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

#ifdef __HIPCC__
// Correctly rounded a / b for a divisor shared by many dividends (b = a small positive integer
// here): y = RN(1/b), q = RN(a y), r = a - b q (exact, one FMA), result RN(q + r y).  This is the
// classical FMA division step; it returns RN(a/b) for every finite a when b's significand is not
// all ones and the quotient is a normal number (checked exhaustively for b = 1..20000 over random
// dividends by tools/check_math.hip).  For b = 0 (the terminal step, whose outputs nobody reads:
// the reference itself produces inf/nan there) the result is NaN.
struct DglConstDiv {
    float b, rb;
};
__device__ __forceinline__ DglConstDiv dgl_const_div(float b) {
    DglConstDiv c;
    c.b = b;
    c.rb = __fdiv_rn(1.0f, b);
    return c;
}
__device__ __forceinline__ float dgl_div(float a, const DglConstDiv &c) {
    const float q = __fmul_rn(a, c.rb);
    const float r = __fmaf_rn(-q, c.b, a);
    return __fmaf_rn(r, c.rb, q);
}

// Correctly rounded sqrt for x = 0, x >= 2^-104 and +inf (radicands on this path are sums of
// products of coordinates and running variances: 0 or far above 1e-31).  v_sqrt_f32 is within
// 1 ulp; the two FMA residuals pick the neighbour whose square brackets x.  For +-0, +inf,
// negative and NaN inputs the residual tests are false (NaN or -0 residuals) and the hardware's
// answer is kept (NaN for negative x, as torch.sqrt).  Same results as sqrtf() on that domain
// (all 2^32 inputs compared by tools/check_math.hip: the only differences are below 2^-104, where
// the residuals underflow) in half the instructions: no pre-scaling for tiny inputs.
__device__ __forceinline__ float dgl_sqrt(float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s0) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s0) + 1u);
    const float r_dn = __fmaf_rn(-s_dn, s0, x);
    const float r_up = __fmaf_rn(-s_up, s0, x);
    float s = (r_dn <= 0.0f) ? s_dn : s0;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}

// Two-wide float arithmetic: gfx950 issues v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 for these, one instruction for
// two independent correctly rounded results (the scan loops are bound by vector-ALU issue as much as by HBM).
typedef float dgl_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dgl_f2 dgl_fma2(dgl_f2 a, dgl_f2 b, dgl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dgl_f2 dgl_splat2(float v) { return dgl_f2{v, v}; }

// dgl_sqrt with both residuals in one packed FMA
__device__ __forceinline__ float dgl_sqrt_pk(float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s0) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s0) + 1u);
    const dgl_f2 res = dgl_fma2(-dgl_f2{s_dn, s_up}, dgl_splat2(s0), dgl_splat2(x));
    float s = (res.x <= 0.0f) ? s_dn : s0;
    s = (res.y > 0.0f) ? s_up : s;
    return s;
}

// ---------------------------------------------------------------------------------------------- square-root models
// Every square root on the decode path that the reference takes with torch.sqrt / torch.cdist (distance entries, the
// running standard deviations) goes through a model object `sq`:
//   DglSqrtIeee  the correctly rounded root;
//   DglSqrtMkl   what those torch calls return on a CPU build of torch linked against MKL that runs on an AVX-512 machine
//                (where the golden vectors were captured; the default, dgl_set_sqrt_model): MKL's vector sqrt refines
//                the instruction VRSQRT14's estimate y once, s = x y, result = fma(fma(-s, s, x), y / 2, s), which is
//                the correctly rounded root or - on 0.6 % of all inputs, whenever the exact root lies less than 0.06 ulp
//                above the midpoint of two floats - the float below it.  The estimate is evaluated in closed form
//                (DGL_RSQRT14_SEG below); it scales exactly with powers of 4.
// sq(x): one root; sq.pk(x): the same value (the IEEE model evaluates its two residuals with one packed FMA there);
// sq.all(ln, x) / sq.dist_all(ln, x): the same values again, for places where all 64 lanes of the wave are active (ln =
// sq.lanes(), taken where they are: see DglSqrtMklT).
// ULPS: how far above a bound on radicands a radicand can lie and still have a root that ties with the bound's.
// DIRECT (instances of at most 25 points, see dgl_direct_sq): distance roots are the correctly rounded ones under
// every model - torch.cdist takes them with the scalar sqrt there; sq.dist(x) / sq.dist_pk(x) are the roots of distance
// radicands, sq(x) / sq.pk(x) those of the running variances.
template <bool DIRECT_>
struct DglSqrtIeeeT {
    static constexpr unsigned ULPS = 4u;
    static constexpr bool DIRECT = DIRECT_;
    static constexpr bool WAVE_WIDE = false;    // no root needs the other lanes of its wave
    static constexpr bool GROUPED = false;
    struct Lanes {};
    __device__ __forceinline__ Lanes lanes() const { return Lanes{}; }
    __device__ __forceinline__ float all(const Lanes &, float x) const { return dgl_sqrt(x); }
    __device__ __forceinline__ float dist_all(const Lanes &, float x) const { return dgl_sqrt(x); }
    template <int K>
    __device__ __forceinline__ void all_n(const Lanes &, const float (&x)[K], float (&out)[K]) const {
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = dgl_sqrt(x[k]);
    }
    template <int K>
    __device__ __forceinline__ void dist_all_n(const Lanes &ln, const float (&x)[K], float (&out)[K]) const { all_n(ln, x, out); }
    __device__ __forceinline__ float operator()(float x) const { return dgl_sqrt(x); }
    __device__ __forceinline__ float pk(float x) const { return dgl_sqrt_pk(x); }
    __device__ __forceinline__ float dist(float x) const { return dgl_sqrt(x); }
    __device__ __forceinline__ float dist_pk(float x) const { return dgl_sqrt_pk(x); }
};
typedef DglSqrtIeeeT<false> DglSqrtIeee;
// VRSQRT14 in closed form.  Over [1, 4) the instruction's estimate is piecewise linear in the input's mantissa: 64
// segments (lowest bit of the biased exponent, upper 5 mantissa bits), and inside a segment
//   (estimate's bits - 0x3F000000) >> 7  =  (C - S m) >> 9,     m = mantissa bits 17..8,
// with one integer pair (C, S) per segment.  The pairs below were solved from the instruction's recorded values
// (tools/make_rsqrt14_table.py: for every segment exactly one pair reproduces its 1024 values; the estimate depends on no
// lower mantissa bit and scales exactly with powers of 4; tests/test_host_cpu.py re-checks all 2^16 recorded values
// against these constants).  {C, -S}, index = (bits >> 18) & 63.
static __constant__ int2 DGL_RSQRT14_SEG[64] = {
    {13896320, -707}, {13171840, -675}, {12480000, -647}, {11817472, -619},
    {11183616, -595}, {10574720, -571}, {9990272, -549}, {9428096, -527},
    {8887936, -509}, {8367488, -491}, {7864960, -473}, {7380608, -457},
    {6912640, -441}, {6460672, -427}, {6023296, -413}, {5600640, -401},
    {5190528, -389}, {4792704, -377}, {4407168, -365}, {4033664, -355},
    {3670400, -345}, {3317504, -335}, {2974208, -325}, {2640896, -317},
    {2316544, -309}, {2000512, -301}, {1692544, -293}, {1392384, -285},
    {1100416, -279}, {814720, -271}, {536576, -265}, {264960, -259},
    {33551488, -1001}, {32526464, -955}, {31548032, -915}, {30611712, -877},
    {29714176, -841}, {28853120, -807}, {28026496, -775}, {27232384, -747},
    {26467584, -719}, {25731200, -693}, {25021312, -669}, {24336896, -647},
    {23675136, -625}, {23035136, -603}, {22417280, -585}, {21818752, -567},
    {21238656, -549}, {20676992, -533}, {20131712, -517}, {19602432, -501},
    {19089024, -487}, {18590080, -473}, {18105344, -461}, {17633664, -449},
    {17174400, -437}, {16727424, -425}, {16292608, -415}, {15867648, -403},
    {15454080, -393}, {15051520, -385}, {14657408, -375}, {14273792, -367},
};
// GROUPED: the scan loops take the roots of a whole group of node chunks together (all_n: one branch, the chunks'
// dependent chains overlap) - 0.2-0.35 us less per launch where a launch is latency-bound, 1.5 % at a saturating batch
// for TSP; the CVRP kernel at a saturating batch loses 12 % that way (96 registers instead of 70), so its launcher picks
// the chunk-by-chunk form from 2048 rows on.
template <bool DIRECT_, bool GROUPED_ = true>
struct DglSqrtMklT {
    static constexpr unsigned ULPS = 8u;
    static constexpr bool DIRECT = DIRECT_;
    static constexpr bool GROUPED = GROUPED_;
    // Where all 64 lanes of a wave are active (the scan loops, the candidates' feature rows) a root is taken with
    // all(ln, x) / dist_all(ln, x): every lane goes through the closed form, the segment constants fetched from the
    // lanes' registers (Lanes: lane l holds segment l) by ds_bpermute - no memory access, no filter.  ds_bpermute
    // returns 0 for a lane that is switched off, hence the condition; kernels of this model (WAVE_WIDE) therefore take
    // those roots before their per-lane guards.  Every other place uses operator() / dist(), which look the constants
    // up in memory and work under any lane mask.
    static constexpr bool WAVE_WIDE = true;
    struct Lanes {
        int c, ns;
    };
    __device__ __forceinline__ Lanes lanes() const {
        const int2 e = DGL_RSQRT14_SEG[threadIdx.x & 63];
        return Lanes{e.x, e.y};
    }
    // MKL's refinement of the estimate whose segment constants are (c, ns): s = x y, result = fma(fma(-s, s, x), y / 2, s)
    // (x = +0 gives +0: the estimate's exponent stays finite)
    __device__ __forceinline__ static float refine(float x, int c, int ns) {
        const unsigned b = __float_as_uint(x);
        const unsigned frac = ((unsigned)(c + __mul24(ns, (int)((b >> 8) & 1023u))) >> 9) << 7;
        // exponent: y = 2^-q (0.5 + frac), q = floor((biased exponent - 127) / 2)
        const unsigned yb = 0x5F000000u - (((b + 0x00800000u) >> 1) & 0x7F800000u) + frac;
        const float y = __uint_as_float(yb);
        const float s = __fmul_rn(x, y), h = __fmul_rn(0.5f, y);
        return __fmaf_rn(__fmaf_rn(-s, s, x), h, s);
    }
    // +0 and positive normal numbers go through the closed form; everything else (negative, NaN, +inf, denormal, -0)
    // takes the hardware's root as under the other model
    __device__ __forceinline__ static bool ordinary(float x) { return __builtin_amdgcn_classf(x, 0x140); }
    // the model's root of x; all 64 lanes active
    __device__ __forceinline__ float all(const Lanes &ln, float x) const {
        const int addr = (int)((__float_as_uint(x) >> 16) & 0xFCu);
        const int c = __builtin_amdgcn_ds_bpermute(addr, ln.c), ns = __builtin_amdgcn_ds_bpermute(addr, ln.ns);
        float out = refine(x, c, ns);
        const bool odd = !ordinary(x);
        if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {
            if (odd) out = dgl_sqrt(x);
        }
        return out;
    }
    // (distance roots under DIRECT: the correctly rounded ones)
    __device__ __forceinline__ float dist_all(const Lanes &ln, float x) const { return DIRECT ? dgl_sqrt(x) : all(ln, x); }
    // K independent roots behind one branch for the unusual inputs: their dependent chains (look-up, refinement) overlap
    template <int K>
    __device__ __forceinline__ void all_n(const Lanes &ln, const float (&x)[K], float (&out)[K]) const {
        bool odd = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int addr = (int)((__float_as_uint(x[k]) >> 16) & 0xFCu);
            const int c = __builtin_amdgcn_ds_bpermute(addr, ln.c), ns = __builtin_amdgcn_ds_bpermute(addr, ln.ns);
            out[k] = refine(x[k], c, ns);
            odd = odd || !ordinary(x[k]);
        }
        if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (!ordinary(x[k])) out[k] = dgl_sqrt(x[k]);
        }
    }
    template <int K>
    __device__ __forceinline__ void dist_all_n(const Lanes &ln, const float (&x)[K], float (&out)[K]) const {
        if (DIRECT) {
#pragma unroll
            for (int k = 0; k < K; ++k) out[k] = dgl_sqrt(x[k]);
        } else {
            all_n(ln, x, out);
        }
    }
    __device__ __forceinline__ float dist(float x) const { return DIRECT ? dgl_sqrt(x) : (*this)(x); }
    __device__ __forceinline__ float dist_pk(float x) const { return dist(x); }
    // one root under any lane mask: the correctly rounded root, replaced where MKL's result can differ from it.  Over all
    // 2^24 inputs of [1, 4) MKL's result is the float below the correctly rounded root on 98 955 of them, the closest
    // one 0.43991 ulp below it (its deficit is about 3/8 of the squared relative error of the 14-bit estimate): only
    // roots that were rounded up by more than DGL_MKL_FILTER / 2 = 0.4375 ulp are evaluated again.
#ifndef DGL_MKL_FILTER
#define DGL_MKL_FILTER 0.875f
#endif
    __device__ __forceinline__ float operator()(float x) const {
        const float s0 = dgl_sqrt(x);
        const float ulp = __fsub_rn(s0, __uint_as_float(__float_as_uint(s0) - 1u));
        const float r = __fmaf_rn(-s0, s0, x);                  // x - s0^2: root - s0 ~ r / (2 s0)
        float res = s0;
        if (r < -DGL_MKL_FILTER * __fmul_rn(s0, ulp) && ordinary(x)) {
            const int2 e = DGL_RSQRT14_SEG[(__float_as_uint(x) >> 18) & 63u];
            res = refine(x, e.x, e.y);
        }
        return res;
    }
    __device__ __forceinline__ float pk(float x) const { return (*this)(x); }
};
typedef DglSqrtMklT<false> DglSqrtMkl;

// One node's running mean / standard deviation of its distances to the unvisited nodes when the node at distance
// dcol leaves that set (TSP/TSPEnv.py:386,408-409; CVRP/CVRPEnv.py:711,733-738): up1 = u + 1, du = division by u,
//   avg' = (avg up1 - dcol) / u,   sd' = sqrt((sd^2 up1 - (dcol - avg)^2) / u)   (CLAMP: radicand < 0 -> 0).
// The same roundings as the scalar formulas, with the two numerators and the two divisions in packed instructions.
// the same without the final root: *item = the radicand of sd'
template <bool CLAMP>
__device__ __forceinline__ void dgl_update_moments_item(float dcol, float a0, float s0, float up1, const DglConstDiv &du,
                                                        float *a1, float *item_out) {
    const float e = __fsub_rn(dcol, a0);
    const dgl_f2 num = dgl_f2{a0, __fmul_rn(s0, s0)} * dgl_splat2(up1) - dgl_f2{dcol, __fmul_rn(e, e)};
    const dgl_f2 q = num * dgl_splat2(du.rb);
    const dgl_f2 rem = dgl_fma2(-q, dgl_splat2(du.b), num);
    const dgl_f2 res = dgl_fma2(rem, dgl_splat2(du.rb), q);
    float item = res.y;
    if (CLAMP && item < 0.0f) item = 0.0f;
    *a1 = res.x;
    *item_out = item;
}
template <bool CLAMP, class SQ = DglSqrtIeee>
__device__ __forceinline__ void dgl_update_moments(float dcol, float a0, float s0, float up1, const DglConstDiv &du,
                                                   float *a1, float *s1, const SQ &sq = SQ{}) {
    const float e = __fsub_rn(dcol, a0);
    const dgl_f2 num = dgl_f2{a0, __fmul_rn(s0, s0)} * dgl_splat2(up1) - dgl_f2{dcol, __fmul_rn(e, e)};
    const dgl_f2 q = num * dgl_splat2(du.rb);
    const dgl_f2 rem = dgl_fma2(-q, dgl_splat2(du.b), num);
    const dgl_f2 res = dgl_fma2(rem, dgl_splat2(du.rb), q);
    float item = res.y;
    if (CLAMP && item < 0.0f) item = 0.0f;
    *a1 = res.x;
    *s1 = sq.pk(item);
}

// |p|^2 as the reference's x.pow(2).sum(-1): two rounded products, one rounded add.
__device__ __forceinline__ float dgl_sqnorm(float x, float y) { return __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)); }

// d(i,j) of torch.cdist's matrix-product evaluation (row i = query, column j):
//   [-2x_i, -2y_i, n_i, 1] . [x_j, y_j, 1, n_j] accumulated left to right with FMAs.
// The first two terms are symmetric in (i,j) (the factor -2 is exact), so d(i,j) and d(j,i) share
// them and differ only in the order of the two norm additions.
__device__ __forceinline__ float dgl_chain_cross(float xi, float yi, float xj, float yj) {
    return __fmaf_rn(-2.0f * yi, yj, __fmul_rn(-2.0f * xi, xj));
}
// squared distance before the square root, clamped at 0 (torch.cdist: clamp_min(0).sqrt())
__device__ __forceinline__ float dgl_chain_sq(float cross, float ni, float nj) {
    return fmaxf(__fadd_rn(__fadd_rn(cross, ni), nj), 0.0f);
}
__device__ __forceinline__ float dgl_chain_dist(float xi, float yi, float ni, float xj, float yj, float nj) {
    return dgl_sqrt(dgl_chain_sq(dgl_chain_cross(xi, yi, xj, yj), ni, nj));
}
// Instances of at most 25 points: torch.cdist does not go through a matrix product there but evaluates
//   d(i,j) = sqrt(fma(dy, dy, dx * dx)),  dx = x_i - x_j, dy = y_i - y_j   (a third coordinate: one more FMA)
// per pair, with the correctly rounded scalar sqrt (checked bit for bit against torch.cdist for 20, 21 and 25 points):
// symmetric, zero diagonal.  The kernels keep their structure (norms, cross term, sum, clamp, root) and switch the
// pieces: with DIRECT the "norms" are 0 and the "cross term" is the whole squared distance, so dgl_chain_sq returns it
// unchanged.
#define DGL_DIRECT_MAX_POINTS 25
__device__ __forceinline__ float dgl_direct_sq(float xi, float yi, float xj, float yj) {
    const float dx = __fsub_rn(xi, xj), dy = __fsub_rn(yi, yj);
    return __fmaf_rn(dy, dy, __fmul_rn(dx, dx));
}
template <bool DIRECT>
__device__ __forceinline__ float dgl_sqnorm_t(float x, float y) { return DIRECT ? 0.0f : dgl_sqnorm(x, y); }
template <bool DIRECT>
__device__ __forceinline__ float dgl_cross_t(float xi, float yi, float xj, float yj) {
    return DIRECT ? dgl_direct_sq(xi, yi, xj, yj) : dgl_chain_cross(xi, yi, xj, yj);
}
__device__ __forceinline__ float dgl_pair_dist(bool direct, float xi, float yi, float xj, float yj) {
    return direct ? dgl_sqrt(dgl_direct_sq(xi, yi, xj, yj))
                  : dgl_chain_dist(xi, yi, dgl_sqnorm(xi, yi), xj, yj, dgl_sqnorm(xj, yj));
}
template <class SQ>
__device__ __forceinline__ float dgl_chain_dist(const SQ &sq, float xi, float yi, float ni, float xj, float yj, float nj) {
    return sq.dist(dgl_chain_sq(dgl_cross_t<SQ::DIRECT>(xi, yi, xj, yj), ni, nj));
}

// v for the lanes whose bit is clear in the wave-uniform mask, +inf for the others: the mask
// serves directly as the condition register of one v_cndmask.
__device__ __forceinline__ float dgl_inf_where(unsigned long long mask, float v) {
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? INFINITY : v;
}
// 64-bit word held by lane `src` (wave-uniform index) of a per-lane value
__device__ __forceinline__ unsigned long long dgl_readlane_u64(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
#endif
