// box_prep.hip -- per-query box constants shared by the fused sampling kernels (gfx950).
//
// decode_bbox(theta_d2xy_coods(query_bbox)) (models/bbox/utils.py:66-90) evaluated once per query
// and layer instead of once per keypoint: table[b,q] = (cx, cy, cz [m], w, l, h [m], cos yaw, sin yaw).
// 900 threads of trigonometry per layer; rac_sampling4d_fwd / rac_bev_sampling_fwd read the table.
#include "rac_common.h"

#define BOX_TWO_PI 6.283185307179586f

__global__ __launch_bounds__(256) void box_prep_kernel(const float *__restrict__ qbox, float *__restrict__ table,
                                                       int n, float p0, float p1, float p2, float sx, float sy, float sz)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float *qb = qbox + (size_t)i * 10;
    const float ang = qb[0] * BOX_TWO_PI, rad = qb[1] * 65.0f;
    const float xn = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
    const float yn = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
    const float yaw = atan2f(qb[6], qb[7]);
    float *t = table + (size_t)i * 8;
    t[0] = xn * sx + p0;
    t[1] = yn * sy + p1;
    t[2] = qb[2] * sz + p2;
    t[3] = expf(qb[3]);
    t[4] = expf(qb[4]);
    t[5] = expf(qb[5]);
    t[6] = cosf(yaw);
    t[7] = sinf(yaw);
}

extern "C" int rac_box_prep_fwd(const float *query_bbox, float *table, int num_boxes, const float *pc_range, void *stream)
{
    RAC_CHECK_ARG(num_boxes >= 0, "rac_box_prep_fwd: num_boxes=%d", num_boxes);
    if (num_boxes == 0)
        return 0;
    RAC_CHECK_ARG(query_bbox && table && pc_range, "rac_box_prep_fwd: null pointer");
    hipLaunchKernelGGL(box_prep_kernel, dim3((num_boxes + 255) / 256), dim3(256), 0, (hipStream_t)stream, query_bbox, table,
                       num_boxes, pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1],
                       pc_range[5] - pc_range[2]);
    return rac_launch_status("rac_box_prep_fwd");
}
