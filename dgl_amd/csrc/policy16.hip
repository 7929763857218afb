// Policy network of the DGL decode step on gfx950, dense contractions on the binary16 matrix
// cores with split operands: encoder block + 3 decoder blocks, d = 128, 8 heads x 16, FF 512,
// no LayerNorm (TSP/TSPModel.py:238-457, CVRP/CVRPModel.py:378-642).
//
// Arithmetic: every weight x activation product is evaluated as
//     (Wh + Wl)(Xh + Xl) ~= Wl*Xh + Wh*Xl + Wh*Xh,   h = binary16(v), l = binary16(v - h),
// three v_mfma_f32_32x32x16_f16 with binary32 accumulation.  Each operand keeps 22 significant
// bits and the products are exact in the accumulator, so the result is as close to the real dot
// product as a binary32 FMA chain (tools/check_mfma16.hip measures both), at 16/3 of the f32
// matrix rate.  Weights are split once by dgl_policy_pack, activations when they are written to
// LDS, Q/K/V and the attention probabilities in registers.  Softmax, residuals, biases and the
// output stage are binary32 VALU.
//
// Execution shape
//   * one workgroup = 4 waves handles ROWS rows (partial tours); a row is one 32-token tile;
//   * wave w owns output features [32w, 32w+32) of every 128-wide projection, i.e. heads 2w and
//     2w+1: Q, K, V and the attention of those heads never leave its registers, and neither does
//     its slice of the residual stream (lane = token, registers = features);
//   * what the other waves need of an activation is its split image in LDS: XH/XL[token][feature]
//     binary16, row stride 136 halfs, so the 16-byte operand read of a lane (8 consecutive k of
//     one token) is bank-conflict free;
//   * weights stream from L2 in consumption order through a register ring per wave; bias vectors
//     are staged in LDS once, so nothing else uses vector loads inside the layer loop and no wait
//     on a small load ever drains the ring.
//
// Operand roles: for D[32x32] += A[32x16] * B[16x32] lane l supplies A[l&31][8(l>>5)+j] and
// B[8(l>>5)+j][l&31], j = 0..7; register r of lane l is D[(r&3)+8(r>>2)+4(l>>5)][l&31].  A lane
// (q, h) therefore always fetches X[token q][16s+8h..] and W[out q][16s+8h..]; passing (W, X)
// gives Y^T (lane = token, registers = features), passing (X, W) gives Y (lane = feature,
// registers = tokens).
#include <cstdlib>

#include "policy16_body.h"
#include "policy16_wide.h"

template <int ROWS, int GROUPS = 1>
__global__ __launch_bounds__(256 * GROUPS) void policy_kernel_f16(const float *__restrict__ P, dgl_policy_io_t io) {
    extern __shared__ float lds[];
    policy16_rows<ROWS, GROUPS>(P, io, lds, blockIdx.x);
}

template <int ROWS, int GROUPS>
static int launch16(const float *P, const dgl_policy_io_t *io, hipStream_t s) {
    const size_t lds = policy16_lds_bytes<ROWS, GROUPS>();
    // (set on every launch: the attribute belongs to the current device's copy of the kernel)
    if (lds > 48 * 1024)
        dgl_max_dynamic_lds((const void *)policy_kernel_f16<ROWS, GROUPS>, (int)lds);
    const int per_wg = ROWS * GROUPS;
    hipLaunchKernelGGL((policy_kernel_f16<ROWS, GROUPS>), dim3((io->R + per_wg - 1) / per_wg), dim3(256 * GROUPS), lds, s, P, *io);
    DGL_CHECK_LAUNCH("dgl_policy_forward");
    return DGL_OK;
}

// rows of 33..128 tokens: one workgroup per row (policy16_wide.h)
template <int NT>
__global__ __launch_bounds__(256) void policy_wide_kernel_f16(const float *__restrict__ P, dgl_policy_io_t io) {
    extern __shared__ float lds[];
    policy16_wide<NT>(P, io, lds, blockIdx.x);
}

template <int NT>
static int launch_wide(const float *P, const dgl_policy_io_t *io, hipStream_t s) {
    const size_t lds = policy16_wide_lds_bytes<NT>();
    static_assert(policy16_wide_lds_bytes<NT>() <= 160 * 1024, "a row's images must fit a workgroup's LDS");
    dgl_max_dynamic_lds((const void *)policy_wide_kernel_f16<NT>, (int)lds);
    hipLaunchKernelGGL((policy_wide_kernel_f16<NT>), dim3(io->R), dim3(256), lds, s, P, *io);
    DGL_CHECK_LAUNCH("dgl_policy_forward");
    return DGL_OK;
}

int dgl_launch_policy16_wide(const float *packed, const dgl_policy_io_t *io, hipStream_t stream) {
    const int nt = (io->n + 31) / 32;
    switch (nt) {
    case 2: return launch_wide<2>(packed, io, stream);
    case 3:
    case 4: return launch_wide<4>(packed, io, stream);
    default: return dgl_set_error(DGL_ERR_UNSUPPORTED, "n=%d tokens per row: the one-launch kernels take 2..%d", io->n, DGL_MAX_WIDE_TOKENS);
    }
}

int dgl_launch_policy16(const float *packed, const dgl_policy_io_t *io, hipStream_t stream) {
    // shape = 10 * groups + rows per wave.  Two rows per workgroup share every weight fragment; a single row per
    // workgroup keeps all 256 CUs busy when there are few rows.
    static const int forced = getenv("DGL_POLICY_SHAPE") ? atoi(getenv("DGL_POLICY_SHAPE")) : 0;
    const int shape = forced ? forced : (io->R >= 512 ? 12 : 11);
    switch (shape) {
    case 11: return launch16<1, 1>(packed, io, stream);
    case 12: return launch16<2, 1>(packed, io, stream);
    // (a shape of two wave groups per workgroup, 21, was measured slower and is no longer built: within 256 registers
    // its matrix results live in VGPRs, where the operand split's inline assembly is not safe - see split_f16.h)
    default: return dgl_set_error(DGL_ERR_INVALID_ARGUMENT, "DGL_POLICY_SHAPE=%d (11 or 12)", shape);
    }
}
