// Instance augmentation of the reference's aug_test / training reset: mix_aug = rotation (even
// instances) or reflection (odd instances) by a random angle, shift by +10, rescale to the unit
// square (utils/utils.py:790-878: Scale, Rotate_aug, Reflect_aug, mix_aug).  The angles are drawn
// by the host (the reference's torch.rand stream); the kernel reproduces the float arithmetic
// operation by operation: rounded products, one rounded add/sub, +10, min/max, IEEE division.
#include "common.h"

// kind[b]: 0 = copy, 1 = rotate (cs = cos t, sin t), 2 = reflect (cs = cos 2t, sin 2t).  One workgroup per instance.
__global__ __launch_bounds__(256) void augment_kernel(const float *__restrict__ xy, int M, const float *__restrict__ cs,
                                                      const int32_t *__restrict__ kind, float *__restrict__ out) {
    __shared__ float red[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float2 *in = reinterpret_cast<const float2 *>(xy) + (size_t)b * M;
    float2 *o = reinterpret_cast<float2 *>(out) + (size_t)b * M;
    const int kd = kind[b];
    if (kd == 0) {
        for (int j = tid; j < M; j += 256) o[j] = in[j];
        return;
    }
    const float c = cs[2 * b], s = cs[2 * b + 1];
    auto moved = [&](float2 p) {
        float2 q;
        if (kd == 1) {
            q.x = __fsub_rn(__fmul_rn(p.x, c), __fmul_rn(p.y, s));
            q.y = __fadd_rn(__fmul_rn(p.x, s), __fmul_rn(p.y, c));
        } else {
            q.x = __fadd_rn(__fmul_rn(p.x, c), __fmul_rn(p.y, s));
            q.y = __fsub_rn(__fmul_rn(p.x, s), __fmul_rn(p.y, c));
        }
        q.x = __fadd_rn(q.x, 10.0f);
        q.y = __fadd_rn(q.y, 10.0f);
        return q;
    };
    float lox = INFINITY, loy = INFINITY, hix = -INFINITY, hiy = -INFINITY;
    for (int j = tid; j < M; j += 256) {
        const float2 q = moved(in[j]);
        lox = fminf(lox, q.x); loy = fminf(loy, q.y);
        hix = fmaxf(hix, q.x); hiy = fmaxf(hiy, q.y);
    }
    lox = wave_min_f32(lox); loy = wave_min_f32(loy); hix = wave_max_f32(hix); hiy = wave_max_f32(hiy);
    if (lane == 0) { red[w][0] = lox; red[w][1] = loy; red[w][2] = hix; red[w][3] = hiy; }
    __syncthreads();
    lox = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    loy = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
    hix = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
    hiy = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    // Scale: translate by the minimum, divide by the larger extent (extent = max - min of the translated values)
    const float rx = __fsub_rn(__fsub_rn(hix, lox), __fsub_rn(lox, lox));
    const float ry = __fsub_rn(__fsub_rn(hiy, loy), __fsub_rn(loy, loy));
    float ratio = fmaxf(rx, ry);
    if (ratio == 0.0f) ratio = 1.0f;
    for (int j = tid; j < M; j += 256) {
        const float2 q = moved(in[j]);
        float2 r;
        r.x = __fdiv_rn(__fsub_rn(q.x, lox), ratio);
        r.y = __fdiv_rn(__fsub_rn(q.y, loy), ratio);
        o[j] = r;
    }
}

extern "C" int dgl_augment(const float *xy, int32_t B, int32_t M, const float *cos_sin, const int32_t *kind, float *xy_out,
                           dgl_stream_t stream) {
    DGL_CHECK_ARG(xy && cos_sin && kind && xy_out && B > 0 && M > 0, "bad argument");
    hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, xy, M, cos_sin, kind, xy_out);
    DGL_CHECK_LAUNCH("dgl_augment");
    return DGL_OK;
}
