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
