// The split-f16 policy network for rows of 33..128 tokens in ONE launch: neighbourhoods of 50 + 50 nodes (the CVRPLIB
// setting, CVRP/test_cvrplib.py:326-327) or 100 nodes (TSP), which the reference evaluates with the same
// TSPModel / CVRPModel forward (TSP/TSPModel.py:238-457, CVRP/CVRPModel.py:378-642).
//
// One workgroup of four waves owns one row = NT tiles of 32 tokens (NT = 2 or 4; rows of 65..96 tokens run as four
// tiles, the last one padding).  As in the 32-token kernel
// (policy16_body.h) wave w owns output features [32w, 32w + 32) = heads 2w, 2w + 1 and keeps its slice of the residual
// stream of EVERY tile in registers.  What changes with more than one tile:
//   * the weight stream is read once per row for Q/K/V, the combination and the anchors (all NT tiles share each
//     fragment: 9 NT matrix instructions per six fragments in the Q/K/V phase), and once per PAIR of tiles for the
//     feed-forward layers, whose hidden-layer images need four LDS images per tile (the ring is pointed back at the
//     feed-forward section of the block for the second pair);
//   * attention runs over all NT key tiles: K and V of the wave's two heads stay in registers in split form (K: 8 NT
//     registers per head, V: 16 NT for both heads), scores of one query tile against all keys are 16 NT registers,
//     softmax over them, then P V accumulates over the key tiles;
//   * LDS: split images of X for all tiles (2 NT; during the attention they hold Q, then its outputs), a work area of
//     8 images (the parked residual stream while Q/K/V and the attention need the registers, then the hidden-layer
//     images B, C of a pair), logits, the staged vectors: 162 KB for NT = 4;
//   * the query-tile loop of the attention and the pair loop of the feed-forward layers are loops in the instruction
//     stream: one block of the network has to fit the 64 KB instruction cache two CUs share.
// Arithmetic, operand roles, range guard: exactly those of the 32-token kernel.
#pragma once
#include "policy16_body.h"

template <int NT>
constexpr size_t policy16_wide_lds_bytes() {
    return (size_t)(2 * NT + 8) * ARR * sizeof(_Float16) + (size_t)(5 * NT * 32 + NHEAD) * sizeof(float);
}

// split images of registers r0..r0+7 of a transposed-form tile (the outputs of one head): dst = [token q][32w + 4h]
__device__ __forceinline__ void store_split_half(const f32x16 &x, int r0, _Float16 *H, _Float16 *L, float &mx) {
#pragma unroll
    for (int g = r0 / 4; g < r0 / 4 + 2; ++g) {
        u32x2 hi, lo;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned a, b;
            split2(x[4 * g + 2 * p], x[4 * g + 2 * p + 1], a, b);
            track_range(x[4 * g + 2 * p], x[4 * g + 2 * p + 1], mx);
            hi[p] = a;
            lo[p] = b;
        }
        *reinterpret_cast<u32x2 *>(H + 8 * g) = hi;
        *reinterpret_cast<u32x2 *>(L + 8 * g) = lo;
    }
}

// Feed-forward layers + residual for ROWS (1 or 2) tiles: the software pipeline of the 32-token kernel (hidden chunks
// of 128 alternate between the images B and C; the epilogue of a chunk's first product rides in the shadow of
// neighbouring products' matrix instructions).  X0: XH of the first tile (XL one image further, tiles 2 images apart);
// W0: the work area (per tile BH BL CH CL).  Ends with the new residual in xres and, if store_x, its images in X.
template <int ROWS>
__device__ __forceinline__ void wide_ff(WeightRing<PF16> &ring, f32x16 (&xres)[ROWS], _Float16 *X0, _Float16 *W0,
                                        const float *vl, int PV, int rd, int wr, bool store_x, float &range_mx) {
    constexpr int XS = 2 * ARR, HS = 4 * ARR;
    _Float16 *XH = X0, *XL = X0 + ARR, *BH = W0, *BL = W0 + ARR, *CH = W0 + 2 * ARR, *CL = W0 + 3 * ARR;
    auto vec4 = [&](int off, int g) { return *reinterpret_cast<const f32x4 *>(vl + off + 8 * g); };
    f32x16 af[ROWS], ahA[ROWS], ahB[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) { af[rr] = zero16(); ahA[rr] = zero16(); ahB[rr] = zero16(); }
    float v0[ROWS], v1[ROWS], t0[ROWS], t1[ROWS];
    unsigned hp[ROWS], lp[ROWS], hq[ROWS], lq[ROWS], lw0[ROWS], lw1[ROWS];
    f32x4 bb[4];
    auto load_bias = [&](int hc) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bb[g] = vec4(PV + PKV_B1 + 128 * hc, g);
    };
    auto piece = [&](auto stc, auto rc, auto pp, const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
        constexpr int st = decltype(stc)::value, rr = decltype(rc)::value, part = decltype(pp)::value;
        constexpr int g = st / 2, e = 2 * (st & 1);
        if constexpr (part == 0) {
            v0[rr] = fmaxf(ah[rr][2 * st] + bb[g][e], 0.0f);
            v1[rr] = fmaxf(ah[rr][2 * st + 1] + bb[g][e + 1], 0.0f);
        } else if constexpr (part == 1) {
            hq[rr] = pack_rtz(v0[rr], v1[rr]);
            track_range(v0[rr], v1[rr], range_mx);
            t0[rr] = minus_half_lo(v0[rr], hq[rr]);
            t1[rr] = minus_half_hi(v1[rr], hq[rr]);
        } else {
            lq[rr] = pack_rtz(t0[rr], t1[rr]);
            if constexpr (st & 1) {
                *reinterpret_cast<u32x2 *>(HH + rr * HS + wr + 8 * g) = u32x2{hp[rr], hq[rr]};
                if constexpr (st == 7) {
                    *reinterpret_cast<u32x2 *>(HL + rr * HS + wr + 8 * g) = u32x2{lp[rr], lq[rr]};
                } else {
                    lw0[rr] = lp[rr];
                    lw1[rr] = lq[rr];
                }
            } else {
                if constexpr (st >= 2) *reinterpret_cast<u32x2 *>(HL + rr * HS + wr + 8 * (g - 1)) = u32x2{lw0[rr], lw1[rr]};
                hp[rr] = hq[rr];
                lp[rr] = lq[rr];
            }
        }
    };
    auto whole = [&](const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
        return [&, HH, HL](auto ss, auto rc, auto pp) { piece(ss, rc, pp, ah, HH, HL); };
    };
    auto half = [&](auto first, const f32x16 (&ah)[ROWS], _Float16 *HH, _Float16 *HL) {
        return [&, HH, HL](auto ss, auto rc, auto pp) {
            constexpr int s = decltype(ss)::value, st0 = decltype(first)::value;
            if constexpr ((s & 1) == 0) piece(std::integral_constant<int, st0 + s / 2>{}, rc, pp, ah, HH, HL);
        };
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    gemm16_t<ROWS, XS, 64, 3, true>(ring, XH + rd, XL + rd, ahA);                                     // G1(0)
    load_bias(0);
    gemm16_t_with<ROWS, XS, 80, 3, true>(ring, XH + rd, XL + rd, ahB, whole(ahA, BH, BL));            // G1(1) || E(0)
    __syncthreads();
    load_bias(1);
    gemm16_t_with<ROWS, HS, 96, 3, true>(ring, BH + rd, BL + rd, af, half(I0{}, ahB, CH, CL));        // G2(0) || E(1)a
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) ahA[rr] = zero16();
    gemm16_t_with<ROWS, XS, 112, 3, true>(ring, XH + rd, XL + rd, ahA, half(I4{}, ahB, CH, CL));      // G1(2) || E(1)b
    __syncthreads();
    load_bias(2);
    gemm16_t_with<ROWS, HS, 128, 3, true>(ring, CH + rd, CL + rd, af, half(I0{}, ahA, BH, BL));       // G2(1) || E(2)a
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) ahB[rr] = zero16();
    gemm16_t_with<ROWS, XS, 144, 3, true>(ring, XH + rd, XL + rd, ahB, half(I4{}, ahA, BH, BL));      // G1(3) || E(2)b
    __syncthreads();
    load_bias(3);
    gemm16_t_with<ROWS, HS, 160, 3, true>(ring, BH + rd, BL + rd, af, whole(ahB, CH, CL));            // G2(2) || E(3)
    __syncthreads();
    // the last barrier came after every wave's last read of these tiles' X (W1 of chunk 3): X may be replaced
    gemm16_t<ROWS, HS, 176, 3, true>(ring, CH + rd, CL + rd, af);                                     // G2(3)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b2 = vec4(PV + PKV_B2, g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) xres[rr][4 * g + e] = xres[rr][4 * g + e] + (af[rr][4 * g + e] + b2[e]);
    }
    if (store_x) {
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) store_split(xres[rr], XH + rr * XS + wr, XL + rr * XS + wr, range_mx);
    }
}

// masking, softmax, probability floor, greedy pick for one row of up to 128 tokens: one wave, two tokens per lane
// (the output stage of TSP/TSPModel.py:312-337, CVRP/CVRPModel.py:462-516); lg: the row's logits in LDS
__device__ __forceinline__ void wide_row_output(const dgl_policy_io_t &io, int r, const float *lg) {
    const int lane = threadIdx.x & 63, n = io.n;
    const int stride = io.out_stride > 0 ? io.out_stride : DGL_MAX_TOKENS;
    const int anchor0 = io.k_last, anchor1 = n - 1;
    const bool cvrp = io.problem == DGL_PROBLEM_CVRP;
    bool cand[2], open[2];
    int act[2];
    float v[2], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        cand[u] = t < n && t != anchor0 && t != anchor1;
        open[u] = cand[u];
        if (cand[u] && cvrp && io.infeasible != nullptr && t < io.k_last)
            open[u] = io.infeasible[(size_t)r * io.idx_stride + t] == 0;
        act[u] = 0x7FFFFFFF;
        if (cand[u])
            act[u] = (!cvrp || t < io.k_last) ? io.idx_last[(size_t)r * io.idx_stride + t]
                                              : io.idx_depot[(size_t)r * io.idx_dep_stride + (t - io.k_last - 1)] + io.N + 1;
        v[u] = t < n ? lg[t] : 0.0f;
        if (t < n && io.logits_out) io.logits_out[(size_t)r * stride + t] = v[u];
        if (open[u]) mx = fmaxf(mx, v[u]);
    }
    mx = wave_max_f32(mx);
    float sum = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        v[u] = open[u] ? expf(v[u] - mx) : 0.0f;
        sum += v[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float best = -1.0f;
    int best_act = 0x7FFFFFFF;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        float pf = v[u] / sum;
        if (pf <= 1e-5f) pf = pf + 1e-7f;                // TSP/TSPModel.py:323-325
        if (!cand[u]) pf = 0.0f;
        if (t < n && io.prob_out) io.prob_out[(size_t)r * stride + t] = pf;
        if (cand[u] && (pf > best || (pf == best && act[u] < best_act))) { best = pf; best_act = act[u]; }
    }
    if (io.selected_out) {
        // highest probability, lowest action id among equals (the arg-max over the reference's dense tensor)
        const float wbest = wave_max_f32(best);
        int cand_act = (best == wbest) ? best_act : 0x7FFFFFFF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand_act = min(cand_act, __shfl_xor(cand_act, o, 64));
        if (lane == 0) io.selected_out[r] = cand_act;
    }
}

// WIDE_STAMPS (profiling builds, tools/bench_wide_stamps.py): thread 0 stores the cycle counter at every phase boundary
// into the row's prob_out slots instead of the probabilities.
#ifdef WIDE_STAMPS
#define WSTAMP() do { if (tid == 0 && io.prob_out && nst < io.out_stride) \
        io.prob_out[(size_t)row * io.out_stride + nst++] = (float)(__builtin_amdgcn_s_memtime() - t_start); } while (0)
#else
#define WSTAMP() do { } while (0)
#endif

// The network for row `row`; called by the 256 threads of a workgroup (it contains workgroup barriers).
template <int NT>
__device__ __forceinline__ void policy16_wide(const float *__restrict__ P, const dgl_policy_io_t &io, float *lds, int row) {
    static_assert(NT == 2 || NT == 4, "2 or 4 tiles of 32 tokens (3 tiles run as 4 with a padded one)");
    constexpr int XS = 2 * ARR;                                      // halfs between the images of consecutive tiles
    _Float16 *XA = reinterpret_cast<_Float16 *>(lds);                // [NT][XH XL]
    _Float16 *WA = XA + NT * XS;                                     // work area: 8 images
    float *part = lds + (2 * NT + 8) * ARR / 2;                      // [4][NT][32] partial logits, then [NT][32] logits
    float *lg = part + 4 * NT * 32;
    float *headv = lg + NT * 32;                                     // [NHEAD] copy of P[0 .. PK_STREAM0)
    float *vecs = headv + PK_VEC0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane & 31, h = lane >> 5;
    const int n = io.n;
    const float *vl = vecs + 32 * w + 4 * h;
    auto vec4 = [&](int off, int g) { return *reinterpret_cast<const f32x4 *>(vl + off + 8 * g); };
    const int rd = q * LDH + 8 * h;                // operand read offset: [token q][8h]
    const int wr = q * LDH + 32 * w + 4 * h;       // split store offset: [token q][32w + 4h]
    const f32x4 *wave_stream = reinterpret_cast<const f32x4 *>(P + PK_STREAM16 + (size_t)w * WAVE_FRAGS * FRAG) + lane;

    float range_mx = 0.0f;
    WeightRing<PF16> ring;
    f32x16 xres[NT];
#ifdef WIDE_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    int nst = 0;
#endif
    WSTAMP();
    {
        constexpr int PIECES = NHEAD / 4, PER = (PIECES + 255) / 256;
        f32x4 hv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) hv[e] = reinterpret_cast<const f32x4 *>(P)[min(tid + e * 256, PIECES - 1)];
        f32x4 t[NT][3];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int tok = 32 * tt + q;
            const bool live = tok < n;
            const f32x4 *tp = reinterpret_cast<const f32x4 *>(io.tile + (size_t)row * io.tile_row_stride +
                                                             (live ? tok : 0) * DGL_TILE_STRIDE);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t[tt][i] = tp[i];
                if (!live) t[tt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        ring.fill(wave_stream);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = tid + e * 256;
            if (i < PIECES) reinterpret_cast<f32x4 *>(headv)[i] = hv[e];
        }
        __syncthreads();
        // embedding, directly in the residual layout (feature f(r) = 32w + (r&3) + 8(r>>2) + 4h of token 32 tt + q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(headv + PK_EMB_W + c * EMB_IN);
            const f32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
            const float be = headv[PK_EMB_B + c];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                float acc = 0.0f;
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[tt][0][f], w0[f], acc);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[tt][1][f], w1[f], acc);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc = fmaf(t[tt][2][f], w2[f], acc);
                xres[tt][r] = (32 * tt + q < n) ? acc + be : 0.0f;
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) store_split(xres[tt], XA + tt * XS + wr, XA + tt * XS + ARR + wr, range_mx);
    }
    __syncthreads();
    WSTAMP();                                          // embedding done

    const int anchor0 = io.k_last, anchor1 = n - 1;    // TSP: k_last == n-1
    for (int layer = 0; layer < 4; ++layer) {
        const int PV = layer * PKV_SIZE;
        const int blk = layer == 0 ? 0 : BLK_FRAGS + ANCHOR_FRAGS + (layer - 1) * BLK_FRAGS;   // the block's first fragment

        if (layer == 1) {
            // decoder entry: re-embed the anchor token(s) (TSP/TSPModel.py:304, CVRP/CVRPModel.py:445-446)
            f32x16 a0[NT], a1[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) { a0[tt] = zero16(); a1[tt] = zero16(); }
            gemm16_t<NT, XS, 0, 3, true>(ring, XA + rd, XA + ARR + rd, a0);
            gemm16_t<NT, XS, 16, 3, true>(ring, XA + rd, XA + ARR + rd, a1);
            __syncthreads();                           // every wave has finished reading X
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int tok = (a == 0) ? anchor0 : anchor1;
                if (a == 0 || io.problem == DGL_PROBLEM_CVRP) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        if (tok == 32 * tt + q) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 bb = vec4(PK_ANCHOR_B - PK_VEC0 + a * D_MODEL, g);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    xres[tt][4 * g + e] = (a == 0 ? a0[tt][4 * g + e] : a1[tt][4 * g + e]) + bb[e];
                            }
                            store_split(xres[tt], XA + tt * XS + wr, XA + tt * XS + ARR + wr, range_mx);
                        }
                    }
                }
            }
            __syncthreads();
        }

        WSTAMP();                                      // block start (after the anchors)
        // The residual stream is not needed until the combination: it waits in the (idle) work area, one float per
        // (tile, register, thread), so that Q/K/V and the attention have the registers to themselves.
        float *parked = reinterpret_cast<float *>(WA);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) parked[(tt * 16 + r) * 256 + tid] = xres[tt][r];

        // ---- Q^T, K^T (transposed form) and V (normal form) of heads 2w, 2w+1 for all tiles ----
        f32x16 aq[NT], ak[NT], av[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) { aq[tt] = zero16(); ak[tt] = zero16(); av[tt] = zero16(); }
        {
            f16x8 nh[NT], nl[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                nh[tt] = *reinterpret_cast<const f16x8 *>(XA + tt * XS + rd);
                nl[tt] = *reinterpret_cast<const f16x8 *>(XA + tt * XS + ARR + rd);
            }
            Unroll<0, 8>::run([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                asm volatile("" ::: "memory");           // loads stay where they are written (see gemm16_t, FENCE)
                const f16x8 qh = as_h8(ring.template take<(6 * s) % PF16>());
                const f16x8 ql = as_h8(ring.template take<(6 * s + 1) % PF16>());
                const f16x8 kh = as_h8(ring.template take<(6 * s + 2) % PF16>());
                const f16x8 kl = as_h8(ring.template take<(6 * s + 3) % PF16>());
                const f16x8 vh = as_h8(ring.template take<(6 * s + 4) % PF16>());
                const f16x8 vl_ = as_h8(ring.template take<(6 * s + 5) % PF16>());
                f16x8 xh[NT], xl[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) { xh[tt] = nh[tt]; xl[tt] = nl[tt]; }
                if constexpr (s + 1 < 8) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        nh[tt] = *reinterpret_cast<const f16x8 *>(XA + tt * XS + rd + 16 * (s + 1));
                        nl[tt] = *reinterpret_cast<const f16x8 *>(XA + tt * XS + ARR + rd + 16 * (s + 1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    aq[tt] = MFMA16(ql, xh[tt], aq[tt]);
                    aq[tt] = MFMA16(qh, xl[tt], aq[tt]);
                    aq[tt] = MFMA16(qh, xh[tt], aq[tt]);
                    ak[tt] = MFMA16(kl, xh[tt], ak[tt]);
                    ak[tt] = MFMA16(kh, xl[tt], ak[tt]);
                    ak[tt] = MFMA16(kh, xh[tt], ak[tt]);
                    av[tt] = MFMA16(xh[tt], vl_, av[tt]);
                    av[tt] = MFMA16(xl[tt], vh, av[tt]);
                    av[tt] = MFMA16(xh[tt], vh, av[tt]);
                    pin_acc(aq[tt]); pin_acc(ak[tt]); pin_acc(av[tt]);
                }
            });
        }

        WSTAMP();                                      // Q/K/V done
        __syncthreads();                               // every wave has read X: the attention outputs take its place
        WSTAMP();

        // ---- attention of the two local heads over all key tiles, straight from the accumulators ----
        {
            // V[key][feature] of key tile kt: keys summed, two K-steps per tile; K of head hd: its 16 dims are one K-step
            f16x8 vh[NT][2], vlo[NT][2], kh[2][NT], klo[2][NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                split8(av[kt], 0, vh[kt][0], vlo[kt][0], range_mx);
                split8(av[kt], 8, vh[kt][1], vlo[kt][1], range_mx);
                split8(ak[kt], 0, kh[0][kt], klo[0][kt], range_mx);
                split8(ak[kt], 8, kh[1][kt], klo[1][kt], range_mx);
            }
            float pmx = 0.0f;                                    // (attention weights lie in [0, 1]: their range is not tracked)
            // Q of every tile waits in split form where the tile's attention output will go: this lane's own 64 bytes of
            // the tile's X images (written and read by this lane only), so that only one tile's Q occupies registers
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                f16x8 qh[2], ql[2];
                split8(aq[qt], 0, qh[0], ql[0], range_mx);
                split8(aq[qt], 8, qh[1], ql[1], range_mx);
                u32x4 *dh = reinterpret_cast<u32x4 *>(XA + qt * XS + q * LDH + 32 * w + 16 * h);
                u32x4 *dl = reinterpret_cast<u32x4 *>(XA + qt * XS + ARR + q * LDH + 32 * w + 16 * h);
                dh[0] = __builtin_bit_cast(u32x4, qh[0]); dh[1] = __builtin_bit_cast(u32x4, ql[0]);
                dl[0] = __builtin_bit_cast(u32x4, qh[1]); dl[1] = __builtin_bit_cast(u32x4, ql[1]);
            }
            // (a loop in the instruction stream, not unrolled: the kernel's code has to stay within the instruction cache -
            // unrolled, one block of the network was 100 KB of code and every pass through it was fetched from memory)
#pragma unroll 1
            for (int qt = 0; qt < NT; ++qt) {
                f32x16 o_sel = zero16();
                const u32x4 *dh = reinterpret_cast<const u32x4 *>(XA + qt * XS + q * LDH + 32 * w + 16 * h);
                const u32x4 *dl = reinterpret_cast<const u32x4 *>(XA + qt * XS + ARR + q * LDH + 32 * w + 16 * h);
                const u32x4 qv[4] = {dh[0], dh[1], dl[0], dl[1]};
#pragma unroll
                for (int hd = 0; hd < 2; ++hd) {
                    const f16x8 qh = __builtin_bit_cast(f16x8, qv[2 * hd]), ql = __builtin_bit_cast(f16x8, qv[2 * hd + 1]);
                    // S^T[key j][query i] for the keys of every tile; lane (i, h) holds keys 32 kt + (m&3) + 8(m>>2) + 4h
                    f32x16 s[NT];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        s[kt] = zero16();
                        s[kt] = MFMA16(klo[hd][kt], qh, s[kt]);
                        s[kt] = MFMA16(kh[hd][kt], ql, s[kt]);
                        s[kt] = MFMA16(kh[hd][kt], qh, s[kt]);
                    }
                    // padded keys: rows of this kernel have more than 16 NT tokens, so only the upper half of the tiles
                    // can hold any (a real branch per tile: the statement in it keeps the compiler from turning it
                    // into sixteen selects that every row would execute)
#pragma unroll
                    for (int kt = NT / 2; kt < NT; ++kt) {
                        if (n < 32 * (kt + 1)) {                 // wave-uniform
                            asm volatile("");
#pragma unroll
                            for (int m = 0; m < 16; ++m) {
                                const int j = 32 * kt + (m & 3) + 8 * (m >> 2) + 4 * h;
                                s[kt][m] = (j < n) ? s[kt][m] : -INFINITY;
                            }
                        }
                    }
                    float mx = s[0][0];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int m = 0; m < 16; ++m) mx = fmaxf(mx, s[kt][m]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    // 1/4 = 1/sqrt(16) folded into the exponent: exp((s - max)/4) = 2^((s - max) log2(e)/4)
                    const float off = -mx * 0.36067376022224085f;
                    float sum = 0.0f;
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int m = 0; m < 16; ++m) {
                            s[kt][m] = __builtin_amdgcn_exp2f(fmaf(s[kt][m], 0.36067376022224085f, off));
                            sum += s[kt][m];
                        }
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.0f / sum;
                    // O^T[d][i] = sum_j V[j][d] P[i][j]; the weights enter unnormalised (largest = 1)
                    f32x16 o = zero16();
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            f16x8 ph, pl;
                            split8(s[kt], 8 * t, ph, pl, pmx);
                            o = MFMA16(vlo[kt][t], ph, o);
                            o = MFMA16(vh[kt][t], pl, o);
                            o = MFMA16(vh[kt][t], ph, o);
                        }
#pragma unroll
                    for (int m = 0; m < 8; ++m) o_sel[8 * hd + m] = o[8 * hd + m] * inv;
                    __builtin_amdgcn_sched_barrier(0);       // one (tile, head) at a time: the scores of two would not fit
                }
                // (both heads' Q have been read: the tile's output may replace them)
                store_split(o_sel, XA + qt * XS + wr, XA + qt * XS + ARR + wr, range_mx);
            }
        }
        WSTAMP();                                      // attention done
        __syncthreads();
        WSTAMP();

        // ---- multi_head_combine + residual for all tiles ----
        {
            f32x16 acc[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[tt] = zero16();
            gemm16_t<NT, XS, 48, 3, true>(ring, XA + rd, XA + ARR + rd, acc);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[tt][r] = parked[(tt * 16 + r) * 256 + tid];
            __syncthreads();                           // every wave has read the attention outputs: O1's images replace them
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = vec4(PV + PKV_BC, g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) xres[tt][4 * g + e] = xres[tt][4 * g + e] + (acc[tt][4 * g + e] + bb[e]);
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) store_split(xres[tt], XA + tt * XS + wr, XA + tt * XS + ARR + wr, range_mx);
        }
        __syncthreads();
        WSTAMP();                                      // combination done

        // ---- feed forward, a pair of tiles at a time (the work area holds the hidden images of one pair) ----
        // One copy of the code: the second pair's residual registers trade places with the first pair's around its pass.
        // The ring is pointed back at the block's feed-forward fragments for the second pass (it was a block ahead: those
        // loads are dropped); a wave that starts the second pass early writes image B only, which every wave has finished
        // reading.
#pragma unroll 1
        for (int pass = 0; pass < NT / 2; ++pass) {
            if (pass == 1) ring.fill(wave_stream + (size_t)(blk + 64) * 64);
            if (pass == 1) {                           // the second pair's residual registers trade places with the first's
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = xres[0][r], b = xres[1][r];
                    xres[0][r] = xres[NT - 2][r]; xres[1][r] = xres[NT - 1][r];
                    xres[NT - 2][r] = a; xres[NT - 1][r] = b;
                }
            }
            wide_ff<2>(ring, *reinterpret_cast<f32x16(*)[2]>(&xres[0]), XA + pass * 2 * XS, WA, vl, PV, rd, wr, layer < 3, range_mx);
            if (pass == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = xres[0][r], b = xres[1][r];
                    xres[0][r] = xres[NT - 2][r]; xres[1][r] = xres[NT - 1][r];
                    xres[NT - 2][r] = a; xres[NT - 1][r] = b;
                }
            }
            WSTAMP();                                  // a pair's feed forward done
        }
        __syncthreads();
        WSTAMP();
    }
    if (io.overflow_out != nullptr && (!(range_mx < 65504.0f) || headv[PK_WEIGHTS_UNFIT] != 0.0f)) *io.overflow_out = 1;

    // ---- Linear_final: logit[token] = bf + sum_c X[token][c] wf[c] ----
    {
        float wf[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = vec4(PK_FINAL_W - PK_VEC0, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) wf[4 * g + e] = v[e];
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = fmaf(xres[tt][r], wf[r], acc);
            acc += __shfl_xor(acc, 32, 64);
            if (h == 0) part[(w * NT + tt) * 32 + q] = acc;
        }
    }
    __syncthreads();
    if (tid < NT * 32) {
        float logit = P[PK_FINAL_B];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) logit += part[ww * NT * 32 + tid];
        if (io.overflow_out != nullptr && tid < n && logit != logit) *io.overflow_out = 1;
        lg[tid] = logit;
    }
    __syncthreads();
#ifdef WIDE_STAMPS
    WSTAMP();
    if (w == 0) { dgl_policy_io_t io2 = io; io2.prob_out = nullptr; wide_row_output(io2, row, lg); }
#else
    if (w == 0) wide_row_output(io, row, lg);
#endif
}
