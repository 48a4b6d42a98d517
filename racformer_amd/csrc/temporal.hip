// temporal.hip -- the element-wise pieces of RadarBEVTemporalEncoder (models/racformer_transformer.py:618-720) that
// torch runs as a dozen tiny launches per ConvGRU step plus a slow generic resize (gfx950).  The convolutions of the
// encoder stay library (MIOpen) / rac_conv3x3_fwd calls.
#include "rac_common.h"

// ConvGRUCell update (models/racformer_transformer.py:705-720) after the gates convolution:
//   z = sigmoid(g[0:C]), r = sigmoid(g[C:2C]), cand = tanh(g[2C:3C] + r * h_prev), h = (1 - z) * h_prev + z * cand
// gates [B][3C][HW], h_prev [B][C][HW] (batch stride hp_bstride), h_out [B][C][HW] (batch stride ho_bstride: the step's
// slot of the [B,T,C,H,W] output, which is also the next step's h_prev).
// Optional: bias_map [3C][HW] added to the gates first (the gates convolution then runs without bias; the map also carries
// the composed matching-layer term, see ConvGRU.fused_pack), h_out2 = a second destination of h (the hidden half of the
// next step's convolution input), so that no concatenation / copy is launched per step.
__global__ __launch_bounds__(256) void gru_gate_kernel(const float *__restrict__ gates, const float *__restrict__ h_prev,
                                                       long hp_bstride, float *__restrict__ h_out, long ho_bstride,
                                                       const float *__restrict__ bias_map, float *__restrict__ h_out2,
                                                       long ho2_bstride, int B, long chw)
{
    const long n4 = (long)B * (chw >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long b = i / (chw >> 2), e = (i - b * (chw >> 2)) * 4;
        const float *g = gates + b * 3 * chw + e;
        rac_f4 zg = rac_ld4(g), rg = rac_ld4(g + chw), cg = rac_ld4(g + 2 * chw);
        if (bias_map) {
            const rac_f4 bz = rac_ld4(bias_map + e), br = rac_ld4(bias_map + chw + e), bc = rac_ld4(bias_map + 2 * chw + e);
            zg.x += bz.x; zg.y += bz.y; zg.z += bz.z; zg.w += bz.w;
            rg.x += br.x; rg.y += br.y; rg.z += br.z; rg.w += br.w;
            cg.x += bc.x; cg.y += bc.y; cg.z += bc.z; cg.w += bc.w;
        }
        const rac_f4 hp = rac_ld4(h_prev + b * hp_bstride + e);
        rac_f4 o;
#define GRU1(c)                                                   \
        {                                                         \
            const float z = 1.f / (1.f + expf(-zg.c));            \
            const float r = 1.f / (1.f + expf(-rg.c));            \
            const float cand = tanhf(cg.c + r * hp.c);            \
            o.c = (1.f - z) * hp.c + z * cand;                    \
        }
        GRU1(x) GRU1(y) GRU1(z) GRU1(w)
#undef GRU1
        *reinterpret_cast<rac_f4 *>(h_out + b * ho_bstride + e) = o;
        if (h_out2)
            *reinterpret_cast<rac_f4 *>(h_out2 + b * ho2_bstride + e) = o;
    }
}

// nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) on [N][C][h][w] -> [N][C][2h][2w]
// (models/racformer_transformer.py:633-636; torch's area_pixel_compute_source_index with align_corners:
// src = dst * (in - 1) / (out - 1)).  One thread per 4 output pixels of a row.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float *__restrict__ src, float *__restrict__ dst, long planes,
                                                         int h, int w)
{
    const int oh = 2 * h, ow = 2 * w, ow4 = ow >> 2;
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const long n = planes * oh * ow4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x4 = (int)(i % ow4);
        const long t = i / ow4;
        const int y = (int)(t % oh);
        const long pl = t / oh;
        const float fy = sy * (float)y;
        const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
        const float ly = fy - (float)y0, hy = 1.f - ly;
        const float *r0 = src + (pl * h + y0) * w, *r1 = src + (pl * h + y1) * w;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float fx = sx * (float)(x4 * 4 + j);
            const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            o[j] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        }
        *reinterpret_cast<rac_f4 *>(dst + (pl * oh + y) * ow + x4 * 4) = (rac_f4){o[0], o[1], o[2], o[3]};
    }
}

extern "C" int rac_gru_gate_fwd(const float *gates, const float *h_prev, int64_t h_prev_bstride, float *h_out,
                                int64_t h_out_bstride, const float *bias_map, float *h_out2, int64_t h_out2_bstride, int B,
                                int C, int HW, void *stream)
{
    const long chw = (long)C * HW;
    RAC_CHECK_ARG(B >= 0 && C > 0 && HW > 0 && chw % 4 == 0 && h_prev_bstride % 4 == 0 && h_out_bstride % 4 == 0 && h_out2_bstride % 4 == 0,
                  "rac_gru_gate_fwd: C*H*W=%ld and the batch strides must be multiples of 4", chw);
    if (B == 0)
        return 0;
    RAC_CHECK_ARG(gates && h_prev && h_out, "rac_gru_gate_fwd: null pointer");
    long blocks = ((long)B * (chw / 4) + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(gru_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gates, h_prev,
                       (long)h_prev_bstride, h_out, (long)h_out_bstride, bias_map, h_out2, (long)h_out2_bstride, B, chw);
    return rac_launch_status("rac_gru_gate_fwd");
}

extern "C" int rac_upsample2x_fwd(const float *src, float *dst, int64_t planes, int h, int w, void *stream)
{
    RAC_CHECK_ARG(planes >= 0 && h > 0 && w > 0 && w % 2 == 0, "rac_upsample2x_fwd: h=%d w=%d (w even)", h, w);
    if (planes == 0)
        return 0;
    RAC_CHECK_ARG(src && dst, "rac_upsample2x_fwd: null pointer");
    long blocks = ((long)planes * 2 * h * (2 * w / 4) + 255) / 256;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long)planes, h, w);
    return rac_launch_status("rac_upsample2x_fwd");
}
