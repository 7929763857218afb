// Fused greedy decode: the whole loop of TSPTester.beamsearch_tour_nodes_shortest for beam 1
// (TSP/TSPTester.py:163-180: model(...) -> arg-max -> env.step(...), N - 1 times) in ONE kernel launch.
//
// Rows are independent on this path, so nothing has to cross workgroups between two decode steps: a workgroup
// (4 waves) owns two rows from the first decision to the last and alternates between
//   policy phase : policy16_rows<2, 1> (policy16_body.h) on the two rows' token tiles -> selected[r]
//   step phase   : tsp_step_row<2, ...> (tsp_step_body.h), two waves per row, both rows side by side:
//                  state update, next neighbourhood, next token tile
// with workgroup barriers in between.  Against one launch per phase this removes two kernel boundaries per decode
// step (about 2 us of drain + launch each, which a captured graph does not remove either) and the start-up ramp of
// 2 x 999 kernels; the arithmetic is the very same device code, so the results are bit-identical to the step-by-step
// path.  The hand-offs (selected node, index set, token tile) stay in global memory: a workgroup's waves share
// their CU's write-through L1, so a workgroup barrier orders them.  The step phase's LDS aliases the policy's.
#include "policy16_body.h"
#include "tsp_step_body.h"

// The two phases are compiled as functions of their own (not inlined into the step loop): inlined, the register
// allocator treats the loop as one body, hoists loop-invariant loads of one phase across the other and spills inside
// the matrix loops (policy phase 187k instead of 142k cycles per decision).
__device__ __attribute__((noinline)) void decode_policy_phase(const float *P, const dgl_policy_io_t &io, float *lds, int wg) {
    policy16_rows<2, 1>(P, io, lds, wg);
}
__device__ __attribute__((noinline)) void decode_step_phase(const dgl_tsp_state_t &st, const int32_t *selected, int t, int knn,
                                                            int32_t *idx_out, float *tile_out, int per_wave, int mom_off,
                                                            float *step_lds, int r, int gw, bool spare) {
    // (`spare` is uniform over a wave pair; both variants meet the same workgroup barriers)
    if (spare)
        tsp_step_row<2, true, true, false, true>(st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, TspBeamLink{}, step_lds, r, gw);
    else
        tsp_step_row<2, true, true, false>(st, selected, t, knn, idx_out, tile_out, per_wave, mom_off, TspBeamLink{}, step_lds, r, gw);
}

__global__ __launch_bounds__(256) void tsp_decode_kernel(dgl_tsp_state_t st, const float *__restrict__ P, dgl_policy_io_t io,
                                                         int t_done, int t_end, int knn, int step_floats, int per_wave,
                                                         int mom_off) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 1, gw = wave & 1;
    // an odd row count leaves the last workgroup's second pair of waves without a row: they go through the first
    // pair's row without storing anything (the step updates a row's moments in place: a second update of the same row
    // would read values the first has already replaced), so that every wave meets every barrier
    const bool spare = (int)blockIdx.x * 2 + grp >= st.R;
    const int r = min((int)blockIdx.x * 2 + grp, st.R - 1);
    float *step_lds = lds + (size_t)grp * step_floats;
    const int N = st.N;
#ifdef FUSED_STAMPS
    unsigned long long cyc_policy = 0, cyc_step = 0, cyc_bar = 0;
    float *stamp_out = io.logits_out;
    io.logits_out = nullptr;
#endif
    for (int t = t_done; t < t_end; ++t) {
#ifdef FUSED_STAMPS
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
        const int k = min(knn, N - t);
        io.n = k + 1;
        io.k_last = k;
        decode_policy_phase(P, io, lds, blockIdx.x);
#ifdef FUSED_STAMPS
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();                                   // selected[r] written, the policy's LDS is free
#ifdef FUSED_STAMPS
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
        decode_step_phase(st, io.selected_out, t + 1, knn, const_cast<int32_t *>(io.idx_last), const_cast<float *>(io.tile),
                          per_wave, mom_off, step_lds, r, gw, spare);
#ifdef FUSED_STAMPS
        const unsigned long long c3 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();                                   // index sets and tiles written, the step's LDS is free
#ifdef FUSED_STAMPS
        const unsigned long long c4 = __builtin_amdgcn_s_memtime();
        cyc_policy += c1 - c0; cyc_step += c3 - c2; cyc_bar += (c2 - c1) + (c4 - c3);
#endif
    }
#ifdef FUSED_STAMPS
    if (stamp_out && (threadIdx.x & 63) == 0) {
        float *o = stamp_out + ((size_t)blockIdx.x * 4 + wave) * 4;
        o[0] = (float)cyc_policy; o[1] = (float)cyc_step; o[2] = (float)cyc_bar; o[3] = (float)(t_end - t_done);
    }
#endif
}

extern "C" int dgl_tsp_greedy_decode(const dgl_tsp_state_t *st, const float *packed_weights, const dgl_policy_io_t *io,
                                     int32_t t_done, int32_t t_end, int32_t knn, dgl_stream_t stream) {
    DGL_CHECK_ARG(st != nullptr && packed_weights != nullptr && io != nullptr, "NULL argument");
    DGL_CHECK_ARG(st->B > 0 && st->N > 1 && st->rows_per_instance > 0 && st->R == st->B * st->rows_per_instance,
                  "bad sizes B=%d N=%d rows_per_instance=%d R=%d", st->B, st->N, st->rows_per_instance, st->R);
    // the fused kernel's step phase implements the matrix-product distances of more than 25 points only (common.h,
    // dgl_direct_sq): smaller instances take dgl_tsp_step + dgl_policy_forward
    if (st->N <= DGL_DIRECT_MAX_POINTS)
        return dgl_set_error(DGL_ERR_UNSUPPORTED, "N=%d: the fused decode takes instances of more than 25 points", st->N);
    // ... and correctly rounded roots only
    if (dgl_sqrt_reference_bits())
        return dgl_set_error(DGL_ERR_UNSUPPORTED, "the fused decode takes correctly rounded square roots: dgl_set_sqrt_model(DGL_SQRT_IEEE) first");
    DGL_CHECK_ARG(st->xy && st->avg && st->sd && st->visited && st->tour, "NULL state buffer");
    DGL_CHECK_ARG(t_done >= 1 && t_done <= t_end && t_end <= st->N, "steps [%d, %d) outside [1, N=%d]", t_done, t_end, st->N);
    DGL_CHECK_ARG(io->problem == DGL_PROBLEM_TSP && io->R == st->R && io->N == st->N, "io does not describe this state");
    DGL_CHECK_ARG(knn >= 1 && knn + 1 <= DGL_MAX_TOKENS, "knn=%d: the fused kernels take rows of at most %d tokens", knn,
                  DGL_MAX_TOKENS);
    DGL_CHECK_ARG(io->tile != nullptr && io->idx_last != nullptr && io->selected_out != nullptr,
                  "tile / idx_last / selected_out missing");
    DGL_CHECK_ARG(io->idx_stride == knn && io->tile_row_stride == (knn + 1) * DGL_TILE_STRIDE,
                  "idx_stride / tile_row_stride must be knn / (knn + 1) * %d (the step kernel's layout)", DGL_TILE_STRIDE);
    DGL_CHECK_ARG(io->arithmetic == DGL_ARITH_SPLIT_F16, "the fused decode runs the split-f16 arithmetic only");
    if (t_done == t_end) return DGL_OK;
    // step area of one row (two waves): keys | tile | scratch | lane minima | candidate lists | counts | (avg, sd) pairs
    const int per_wave = st->N + (knn + 1) * DGL_TILE_STRIDE + 128;
    const int split_floats = per_wave + 2 * (64 + 128) + 4;
    const int mom_off = (split_floats + 1) & ~1;
    const int step_floats = (mom_off + 2 * st->N + 3) & ~3;
    const size_t lds = policy16_lds_bytes<2, 1>() > (size_t)2 * step_floats * sizeof(float) ? policy16_lds_bytes<2, 1>()
                                                                                          : (size_t)2 * step_floats * sizeof(float);
    if (lds > 150 * 1024)
        return dgl_set_error(DGL_ERR_UNSUPPORTED, "N=%d: two rows' step arrays (%zu bytes) do not fit the workgroup's LDS", st->N, lds);
    dgl_policy_io_t k = *io;
#ifndef FUSED_STAMPS
    k.logits_out = nullptr;
#endif
#if !(defined(FUSED_STAMPS) && defined(PROFILE_STAMPS))
    k.prob_out = nullptr;
#endif
    dgl_max_dynamic_lds((const void *)tsp_decode_kernel, (int)lds);
    hipLaunchKernelGGL(tsp_decode_kernel, dim3((st->R + 1) / 2), dim3(256), lds, (hipStream_t)stream, *st, packed_weights, k,
                       t_done, t_end, knn, step_floats, per_wave, mom_off);
    DGL_CHECK_LAUNCH("dgl_tsp_greedy_decode");
    return DGL_OK;
}
