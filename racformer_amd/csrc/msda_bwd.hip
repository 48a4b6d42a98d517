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
// value scatter uses atomics (one contiguous 256-byte head row per group-instruction).
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
    const rac_f4 z = {0.f, 0.f, 0.f, 0.f};
    const rac_f4 g = act ? rac_ld4(a.grad_out + item * 64 + lane16 * 4) : z;
    const int H = a.H[l], W = a.W[l];
    const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
    const bool in = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int stride = a.heads * 64;
    const size_t map = (((size_t)b * a.keys + a.start[l]) * a.heads + h) * 64 + lane16 * 4;
    const float *base = a.value + map;
    float *gbase = a.gvalue + map;
    const bool ok1 = in && h_low >= 0 && w_low >= 0, ok2 = in && h_low >= 0 && w_high <= W - 1;
    const bool ok3 = in && h_high <= H - 1 && w_low >= 0, ok4 = in && h_high <= H - 1 && w_high <= W - 1;
    const size_t o1 = ((size_t)h_low * W + w_low) * stride, o2 = ((size_t)h_low * W + w_high) * stride;
    const size_t o3 = ((size_t)h_high * W + w_low) * stride, o4 = ((size_t)h_high * W + w_high) * stride;
    const rac_f4 v1 = ok1 ? rac_ld4(base + o1) : z, v2 = ok2 ? rac_ld4(base + o2) : z;
    const rac_f4 v3 = ok3 ? rac_ld4(base + o3) : z, v4 = ok4 ? rac_ld4(base + o4) : z;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    const rac_f4 tg = {g.x * at, g.y * at, g.z * at, g.w * at};
#define DB_ADD4(ptr, wt)                   \
    do {                                   \
        atomicAdd((ptr), (wt) * tg.x);     \
        atomicAdd((ptr) + 1, (wt) * tg.y); \
        atomicAdd((ptr) + 2, (wt) * tg.z); \
        atomicAdd((ptr) + 3, (wt) * tg.w); \
    } while (0)
    if (ok1) DB_ADD4(gbase + o1, w1);
    if (ok2) DB_ADD4(gbase + o2, w2);
    if (ok3) DB_ADD4(gbase + o3, w3);
    if (ok4) DB_ADD4(gbase + o4, w4);
#undef DB_ADD4
#define DB_DOT(fx)                                                                                   \
    ((fx(v1.x, v2.x, v3.x, v4.x)) * g.x + (fx(v1.y, v2.y, v3.y, v4.y)) * g.y + (fx(v1.z, v2.z, v3.z, v4.z)) * g.z + \
     (fx(v1.w, v2.w, v3.w, v4.w)) * g.w)
#define DB_VAL(a1, a2, a3, a4) (w1 * (a1) + w2 * (a2) + w3 * (a3) + w4 * (a4))
#define DB_DH(a1, a2, a3, a4) (-hw * (a1) - lw * (a2) + hw * (a3) + lw * (a4))
#define DB_DW(a1, a2, a3, a4) (-hh * (a1) + hh * (a2) - lh * (a3) + lh * (a4))
    const float sv = db_group_sum16(DB_DOT(DB_VAL));
    const float sh = db_group_sum16(DB_DOT(DB_DH));
    const float sw_ = db_group_sum16(DB_DOT(DB_DW));
#undef DB_DOT
#undef DB_VAL
#undef DB_DH
#undef DB_DW
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
