// msda_bwd.hip -- multi-scale deformable attention, backward, for gfx950.
//
// Replaces mmcv-full 1.6.0's `_ext.ms_deform_attn_backward` at the call site
// models/multi_scale_deformable_attn_function.py:148-158 (Deformable-DETR col2im semantics, the same
// formulas the in-tree msmv backward derives from, msmv_sampling_backward.cu:29-105, with
// align_corners=False: h_im = y*H - 0.5, w_im = x*W - 0.5):
//   grad_value[tap]   += w_tap * grad_out * attn          (scatter, float atomics)
//   grad_attn[p]       = sum_c grad_out[c] * bilinear[c]
//   grad_loc[p] (x,y)  = (W | H) * attn * sum_c grad_out[c] * d bilinear[c] / d(w|h)
// A 16-lane group owns one (batch, query, head, level, point) sample (dim=64: 4 channels per lane); the
// channel sums are a butterfly inside the group with one writer per element (deterministic); only the
// value scatter uses atomics.  Round 4: lane c of a group owns channels c + 16 j (not 4c .. 4c+3), so that every atomic
// instruction adds whole 64-byte segments (see msmv_bwd.hip).
// The caller zero-fills grad_value; grad_loc / grad_attn are fully overwritten.
#include "rac_common.h"

struct MsdaBwdArgs {
    const float *grad_out;  // [bs,Q,heads*dim]
    const float *value;     // [bs,keys,heads,dim]
    const float *loc;       // [bs,Q,heads,L,P,2]
    const float *attn;      // [bs,Q,heads,L,P]
    float *gvalue, *gloc, *gattn;
    int H[RAC_MAX_LEVELS], W[RAC_MAX_LEVELS];
    long start[RAC_MAX_LEVELS];
    int bs, keys, heads, dim, Q, L, P;
};

__device__ __forceinline__ float db_group_sum16(float v)
{
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 16);
    return v;
}

__global__ __launch_bounds__(256) void msda_bwd_d64_kernel(const MsdaBwdArgs a)
{
    const int lane16 = threadIdx.x & 15;
    const long smp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;  // ((b*Q+q)*heads+h)*L*P + l*P + p
    const long total = (long)a.bs * a.Q * a.heads * a.L * a.P;
    const bool act = smp < total;
    const long sc = act ? smp : 0;
    const int lp_ = (int)(sc % ((long)a.L * a.P));
    const int l = lp_ / a.P;
    const long item = sc / ((long)a.L * a.P);  // (b*Q+q)*heads + h
    const int h = (int)(item % a.heads);
    const int b = (int)(item / ((long)a.heads * a.Q));
    const float x = a.loc[sc * 2], y = a.loc[sc * 2 + 1];
    const float at = a.attn[sc];
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        g[j] = act ? a.grad_out[item * 64 + 16 * j + lane16] : 0.f;
    const int H = a.H[l], W = a.W[l];
    const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
    const bool in = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int stride = a.heads * 64;
    const size_t map = (((size_t)b * a.keys + a.start[l]) * a.heads + h) * 64 + lane16;
    const float *base = a.value + map;
    float *gbase = a.gvalue + map;
    const bool ok[4] = {in && h_low >= 0 && w_low >= 0, in && h_low >= 0 && w_high <= W - 1,
                        in && h_high <= H - 1 && w_low >= 0, in && h_high <= H - 1 && w_high <= W - 1};
    const size_t o[4] = {((size_t)h_low * W + w_low) * stride, ((size_t)h_low * W + w_high) * stride,
                         ((size_t)h_high * W + w_low) * stride, ((size_t)h_high * W + w_high) * stride};
    const float tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
    const float dh[4] = {-hw, -lw, hw, lw}, dw[4] = {-hh, hh, -lh, lh};
    float sv = 0.f, sh = 0.f, sw_ = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = ok[t] ? base[o[t] + 16 * j] : 0.f;
        if (ok[t]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                atomicAdd(gbase + o[t] + 16 * j, tw[t] * (g[j] * at));
        }
        const float dot = (v[0] * g[0] + v[1] * g[1]) + (v[2] * g[2] + v[3] * g[3]);
        sv += tw[t] * dot;
        sh += dh[t] * dot;
        sw_ += dw[t] * dot;
    }
    sv = db_group_sum16(sv);
    sh = db_group_sum16(sh);
    sw_ = db_group_sum16(sw_);
    if (act && lane16 == 0) {
        a.gattn[sc] = sv;
        a.gloc[sc * 2] = (float)W * sw_ * at;
        a.gloc[sc * 2 + 1] = (float)H * sh * at;
    }
}

__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(const MsdaBwdArgs a)
{
    const long total = (long)a.bs * a.Q * a.heads * a.L * a.P;
    for (long sc = (long)blockIdx.x * blockDim.x + threadIdx.x; sc < total; sc += (long)gridDim.x * blockDim.x) {
        const int lp_ = (int)(sc % ((long)a.L * a.P));
        const int l = lp_ / a.P;
        const long item = sc / ((long)a.L * a.P);
        const int h = (int)(item % a.heads);
        const int b = (int)(item / ((long)a.heads * a.Q));
        const float x = a.loc[sc * 2], y = a.loc[sc * 2 + 1], at = a.attn[sc];
        const int H = a.H[l], W = a.W[l];
        const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
        float sv = 0.f, sh = 0.f, sw_ = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            const int stride = a.heads * a.dim;
            const size_t map = (((size_t)b * a.keys + a.start[l]) * a.heads + h) * a.dim;
            const float *base = a.value + map;
            float *gbase = a.gvalue + map;
            const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
            const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
            const size_t o1 = ((size_t)h_low * W + w_low) * stride, o2 = ((size_t)h_low * W + w_high) * stride;
            const size_t o3 = ((size_t)h_high * W + w_low) * stride, o4 = ((size_t)h_high * W + w_high) * stride;
            for (int c = 0; c < a.dim; ++c) {
                const float g = a.grad_out[item * a.dim + c];
                const float tg = g * at;
                const float v1 = ok1 ? base[o1 + c] : 0.f, v2 = ok2 ? base[o2 + c] : 0.f;
                const float v3 = ok3 ? base[o3 + c] : 0.f, v4 = ok4 ? base[o4 + c] : 0.f;
                if (ok1) atomicAdd(gbase + o1 + c, w1 * tg);
                if (ok2) atomicAdd(gbase + o2 + c, w2 * tg);
                if (ok3) atomicAdd(gbase + o3 + c, w3 * tg);
                if (ok4) atomicAdd(gbase + o4 + c, w4 * tg);
                sv += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * g;
                sh += (-hw * v1 - lw * v2 + hw * v3 + lw * v4) * g;
                sw_ += (-hh * v1 + hh * v2 - lh * v3 + lh * v4) * g;
            }
        }
        a.gattn[sc] = sv;
        a.gloc[sc * 2] = (float)W * sw_ * at;
        a.gloc[sc * 2 + 1] = (float)H * sh * at;
    }
}

extern "C" int rac_msda_bwd(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *starts,
                            const float *loc, const float *attn, float *grad_value, float *grad_loc, float *grad_attn,
                            int bs, int keys, int heads, int dim, int Q, int L, int P, void *stream)
{
    RAC_CHECK_ARG(L >= 1 && L <= RAC_MAX_LEVELS, "rac_msda_bwd: L=%d out of [1,%d]", L, RAC_MAX_LEVELS);
    RAC_CHECK_ARG(bs >= 0 && Q >= 0 && heads >= 1 && dim >= 1 && keys >= 0 && P >= 0, "rac_msda_bwd: bad sizes");
    if (bs == 0 || Q == 0 || P == 0)
        return 0;
    RAC_CHECK_ARG(grad_out && value && shapes && starts && loc && attn && grad_value && grad_loc && grad_attn,
                  "rac_msda_bwd: null pointer");
    MsdaBwdArgs a;
    for (int l = 0; l < RAC_MAX_LEVELS; ++l) {
        a.H[l] = a.W[l] = 1;
        a.start[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = starts[l];
        RAC_CHECK_ARG(h >= 1 && w >= 1 && st >= 0 && st + h * w <= keys, "rac_msda_bwd: level %d exceeds keys=%d", l, keys);
        a.H[l] = (int)h;
        a.W[l] = (int)w;
        a.start[l] = (long)st;
    }
    a.grad_out = grad_out; a.value = value; a.loc = loc; a.attn = attn;
    a.gvalue = grad_value; a.gloc = grad_loc; a.gattn = grad_attn;
    a.bs = bs; a.keys = keys; a.heads = heads; a.dim = dim; a.Q = Q; a.L = L; a.P = P;
    const long total = (long)bs * Q * heads * L * P;
    hipStream_t st = (hipStream_t)stream;
    if (dim == 64) {
        hipLaunchKernelGGL(msda_bwd_d64_kernel, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0, st, a);
    } else {
        const unsigned nb = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(msda_bwd_generic_kernel, dim3(nb), dim3(256), 0, st, a);
    }
    return rac_launch_status("rac_msda_bwd");
}
