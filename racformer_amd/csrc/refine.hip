// refine.hip -- box refinement tail of a decoder layer as one kernel (gfx950).
//
// Replaces ~27 elementwise launches per layer: refine_bbox (models/racformer_transformer.py:230-236:
// theta += (2*sigmoid(d0)-1)/num_ray; (d,z) = sigmoid(delta + inverse_sigmoid(d,z)); dims 3..9 from the
// regression output), the velocity scaling by time_diff[:,1] (:265-269, values < 1e-5 replaced by 1) and
// the polar -> normalised-xy conversion of the emitted boxes (theta_d2xy_coods, models/bbox/utils.py:82-90;
// :134).  One thread per query; pure elementwise fp32.
#include "rac_common.h"

#define REF_TWO_PI 6.283185307179586f

__device__ __forceinline__ float ref_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float ref_inverse_sigmoid(float x)
{
    // models/utils.py:86-101, eps = 1e-5
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
    return logf(x1 / x2);
}

__global__ __launch_bounds__(256) void refine_kernel(const float *__restrict__ prop, const float *__restrict__ delta,
                                                     const float *__restrict__ td_safe, float *__restrict__ pred,
                                                     float *__restrict__ xy, int n, int Q, int T, float num_ray)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float *p = prop + (size_t)i * 10, *d = delta + (size_t)i * 10;
    float o[10];
    o[0] = p[0] + (ref_sigmoid(d[0]) * 2.f - 1.f) / num_ray;
    o[1] = ref_sigmoid(d[1] + ref_inverse_sigmoid(p[1]));
    o[2] = ref_sigmoid(d[2] + ref_inverse_sigmoid(p[2]));
#pragma unroll
    for (int k = 3; k < 10; ++k)
        o[k] = d[k];
    if (T > 1) {
        const float td = td_safe[(i / Q) * T + 1];
        o[8] = o[8] / td;
        o[9] = o[9] / td;
    }
    float *pp = pred + (size_t)i * 10, *px = xy + (size_t)i * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k)
        pp[k] = o[k];
    const float ang = o[0] * REF_TWO_PI, rad = o[1] * 65.0f;
    px[0] = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    px[1] = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
#pragma unroll
    for (int k = 2; k < 10; ++k)
        px[k] = o[k];
}

extern "C" int rac_refine_fwd(const float *proposal, const float *delta, const float *time_diff_safe, float *bbox_pred,
                              float *bbox_xy, int B, int Q, int T, float num_ray, void *stream)
{
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && num_ray > 0.f, "rac_refine_fwd: bad sizes B=%d Q=%d T=%d", B, Q, T);
    if (B * Q == 0)
        return 0;
    RAC_CHECK_ARG(proposal && delta && time_diff_safe && bbox_pred && bbox_xy, "rac_refine_fwd: null pointer");
    const int n = B * Q;
    hipLaunchKernelGGL(refine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, proposal, delta,
                       time_diff_safe, bbox_pred, bbox_xy, n, Q, T, num_ray);
    return rac_launch_status("rac_refine_fwd");
}

// ------------------------------------------------------------------------------------------------ head outputs
// What RaCFormerTransformer.forward and RaCFormer_head.forward do to the stacked decoder outputs, one launch instead of five:
//   cls  <- nan_to_num(cls)                                      (racformer_transformer.py:58)
//   box  <- nan_to_num(xy), then centre * pc_range span + origin and the (x, y, w, l, z, h, ...) column order
//           (racformer_head.py:124-131).  nan_to_num first, as the reference: a NaN centre becomes the range origin.
__device__ __forceinline__ float hf_nan_to_num(float v)
{
    if (v != v)
        return 0.f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);   // +-inf -> +-FLT_MAX (torch.nan_to_num defaults)
}

__global__ __launch_bounds__(256) void head_finish_kernel(float *__restrict__ cls, long n_cls, const float *__restrict__ xy,
                                                          float *__restrict__ box, long rows, float x0, float y0, float z0,
                                                          float sx, float sy, float sz)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_cls)
        cls[i] = hf_nan_to_num(cls[i]);
    if (i < rows) {
#pragma clang fp contract(off)   /* multiply and add rounded separately, as the reference's two tensor operations */
        const float *s = xy + i * 10;
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; ++k)
            v[k] = hf_nan_to_num(s[k]);
        float *d = box + i * 10;
        d[0] = v[0] * sx + x0; d[1] = v[1] * sy + y0; d[2] = v[3]; d[3] = v[4]; d[4] = v[2] * sz + z0;
        d[5] = v[5]; d[6] = v[6]; d[7] = v[7]; d[8] = v[8]; d[9] = v[9];
    }
}

extern "C" int rac_head_finish_fwd(float *cls, int64_t n_cls, const float *xy, float *box, int64_t rows, int code_size,
                                   const float *pc_range, void *stream)
{
    RAC_CHECK_ARG(n_cls >= 0 && rows >= 0 && code_size == 10, "rac_head_finish_fwd: n_cls=%lld rows=%lld code_size=%d (10)",
                  (long long)n_cls, (long long)rows, code_size);
    const long n = n_cls > rows ? n_cls : rows;
    if (n == 0)
        return 0;
    RAC_CHECK_ARG((cls || n_cls == 0) && (rows == 0 || (xy && box)) && pc_range && xy != box, "rac_head_finish_fwd: null pointer or xy aliases box");
    hipLaunchKernelGGL(head_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cls, (long)n_cls, xy, box,
                       (long)rows, pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1],
                       pc_range[5] - pc_range[2]);
    return rac_launch_status("rac_head_finish_fwd");
}

// ------------------------------------------------------------------------------------------------ layer boundary
// One launch at the boundary between two decoder layers: refine_bbox of the finished layer (as refine_kernel) and, for
// the boxes it produces, what the next layer computes first: the per-query box table (box_prep_kernel) and the head of the
// position encoder relu(LN(W x + b)) on (theta, d, z) (pe_head_kernel).  One wave per query: every lane evaluates the
// (cheap, identical) box arithmetic, lane 0 stores it, then the wave does the 256-wide LayerNorm.
__device__ __forceinline__ float lb_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void layer_boundary_kernel(const float *__restrict__ prop, const float *__restrict__ delta,
                                                             const float *__restrict__ td_safe, float *__restrict__ pred,
                                                             float *__restrict__ xy, float *__restrict__ table,
                                                             const float *__restrict__ W, const float *__restrict__ bias,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ h_out, int n, int Q, int T, float num_ray,
                                                             float eps, float p0, float p1, float p2, float sx, float sy, float sz)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n)
        return;
    const float *p = prop + (size_t)i * 10, *d = delta + (size_t)i * 10;
    float o[10];
    o[0] = p[0] + (ref_sigmoid(d[0]) * 2.f - 1.f) / num_ray;
    o[1] = ref_sigmoid(d[1] + ref_inverse_sigmoid(p[1]));
    o[2] = ref_sigmoid(d[2] + ref_inverse_sigmoid(p[2]));
#pragma unroll
    for (int k = 3; k < 10; ++k)
        o[k] = d[k];
    if (T > 1) {
        const float td = td_safe[(i / Q) * T + 1];
        o[8] = o[8] / td;
        o[9] = o[9] / td;
    }
    const float ang = o[0] * REF_TWO_PI, rad = o[1] * 65.0f;
    const float xn = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    const float yn = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
    if (lane == 0) {
        float *pp = pred + (size_t)i * 10, *px = xy + (size_t)i * 10, *t = table + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < 10; ++k)
            pp[k] = o[k];
        px[0] = xn;
        px[1] = yn;
#pragma unroll
        for (int k = 2; k < 10; ++k)
            px[k] = o[k];
        const float yaw = atan2f(o[6], o[7]);
        t[0] = xn * sx + p0;
        t[1] = yn * sy + p1;
        t[2] = o[2] * sz + p2;
        t[3] = expf(o[3]);
        t[4] = expf(o[4]);
        t[5] = expf(o[5]);
        t[6] = cosf(yaw);
        t[7] = sinf(yaw);
    }
    // position-encoder head on (theta, d, z) of the refined box (as pe_head_kernel)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j;
        v[j] = W[c * 3] * o[0] + W[c * 3 + 1] * o[1] + W[c * 3 + 2] * o[2] + bias[c];
    }
    const float mean = lb_wave_sum((v[0] + v[1]) + (v[2] + v[3])) / 256.f;
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    const float rstd = 1.f / sqrtf(lb_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 256.f + eps);
    const rac_f4 g = rac_ld4(gamma + lane * 4), b = rac_ld4(beta + lane * 4);
    rac_f4 y = {fmaxf(d0 * rstd * g.x + b.x, 0.f), fmaxf(d1 * rstd * g.y + b.y, 0.f), fmaxf(d2 * rstd * g.z + b.z, 0.f),
                fmaxf(d3 * rstd * g.w + b.w, 0.f)};
    *reinterpret_cast<rac_f4 *>(h_out + (size_t)i * 256 + lane * 4) = y;
}

extern "C" int rac_layer_boundary_fwd(const float *proposal, const float *delta, const float *time_diff_safe, float *bbox_pred,
                                      float *bbox_xy, float *box_table, const float *pc_range, const float *pe_weight,
                                      const float *pe_bias, const float *pe_gamma, const float *pe_beta, float *pe_out, int B,
                                      int Q, int T, int dim, float num_ray, float eps, void *stream)
{
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && T >= 1 && num_ray > 0.f && dim == 256, "rac_layer_boundary_fwd: bad sizes B=%d Q=%d T=%d dim=%d", B, Q, T, dim);
    if (B * Q == 0)
        return 0;
    RAC_CHECK_ARG(proposal && delta && time_diff_safe && bbox_pred && bbox_xy && box_table && pc_range && pe_weight && pe_bias &&
                      pe_gamma && pe_beta && pe_out,
                  "rac_layer_boundary_fwd: null pointer");
    const int n = B * Q;
    hipLaunchKernelGGL(layer_boundary_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, proposal, delta,
                       time_diff_safe, bbox_pred, bbox_xy, box_table, pe_weight, pe_bias, pe_gamma, pe_beta, pe_out, n, Q, T,
                       num_ray, eps, pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1],
                       pc_range[5] - pc_range[2]);
    return rac_launch_status("rac_layer_boundary_fwd");
}
