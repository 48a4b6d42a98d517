// msda_fwd.hip -- multi-scale deformable attention, forward, for gfx950 (MI355X).
//
// Replaces mmcv-full 1.6.0's `_ext.ms_deform_attn_forward` at the call site
// models/multi_scale_deformable_attn_function.py:118-124 (used by BEVSelfAttention,
// models/bev_self_attention.py:199-201).  Deformable-DETR semantics: h_im = y*H - 0.5,
// w_im = x*W - 0.5 (align_corners=False), point skipped unless h_im>-1 && w_im>-1 && h_im<H &&
// w_im<W, 4 bounds-checked taps, zero padding; out[b,q,h*D+c] = sum_{l,p} attn * bilinear.
//
// CDNA4 mapping: dim=64 fast path -- a 16-lane group owns one (batch, query, head) item, 4
// channels per lane (16-byte loads; one wave-instruction = four 256-byte head rows).  Points
// are unrolled by 4 so 16 tap loads are in flight per lane.  The four items of a wave are
// consecutive heads of one query, so the wave stores one contiguous 1 KiB output row.
// Locations/weights are staged through LDS once per workgroup; blocks are mapped so that all
// workgroups of one BEV frame (one `bs` index, 16.8 MB of value) run on one XCD.
#include "rac_common.h"

struct MsdaArgs {
    const void *value;
    const float *loc;
    const float *attn;
    float *out;
    int H[RAC_MAX_LEVELS], W[RAC_MAX_LEVELS];
    long start[RAC_MAX_LEVELS];
    int bs, keys, heads, dim, Q, L, P;
    int blocks_per_b;
};

#define MSDA_ITEMS 16 /* items per 256-thread workgroup: 4 waves x 4 sixteen-lane groups */

template <typename FT>
__device__ __forceinline__ rac_f4 msda_tap(const FT *base, long pix, int stride, bool ok)
{
    rac_f4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        v = rac_ld4(base + pix * stride);
    return v;
}

template <typename FT>
__global__ __launch_bounds__(256) void msda_fwd_d64_kernel(const MsdaArgs a)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, c4 = lane & 15;
    const int grp = tid >> 4;  // 0..15: item within the workgroup
    const int LP = a.L * a.P;

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int b = xcd + 8 * (j / a.blocks_per_b);
    if (b >= a.bs)
        return;
    const int per_b = a.Q * a.heads;
    const int i0 = (j % a.blocks_per_b) * MSDA_ITEMS;
    const int nitems = min(MSDA_ITEMS, per_b - i0);
    const size_t item0 = (size_t)b * per_b + i0;

    float *sloc = smem;                          // [items][L*P][2]
    float *sattn = smem + MSDA_ITEMS * LP * 2;   // [items][L*P]
    {
        const float *gl = a.loc + item0 * LP * 2;
        const float *ga = a.attn + item0 * LP;
        for (int i = tid; i < nitems * LP * 2; i += 256)
            sloc[i] = gl[i];
        for (int i = tid; i < nitems * LP; i += 256)
            sattn[i] = ga[i];
    }
    __syncthreads();
    if (grp >= nitems)
        return;
    const int h = (i0 + grp) % a.heads;
    const int stride = a.heads * 64;  // elements between neighbouring keys
    const float *lp = sloc + grp * LP * 2;
    const float *ap = sattn + grp * LP;

    rac_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < a.L; ++l) {
        const int H = a.H[l], W = a.W[l];
        const FT *base = (const FT *)a.value + (((size_t)b * a.keys + a.start[l]) * a.heads + h) * 64 + c4 * 4;
        for (int p0 = 0; p0 < a.P; p0 += 4) {
            rac_f4 v[4][4];
            float tw[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = p0 + k;
                const bool act = p < a.P;
                const int pp = act ? p : a.P - 1;
                const float x = lp[(l * a.P + pp) * 2], y = lp[(l * a.P + pp) * 2 + 1];
                const float at = act ? ap[l * a.P + pp] : 0.f;
                const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
                const bool in = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                const bool t_ok = in && h_low >= 0, b_ok = in && h_high <= H - 1;
                const bool l_ok = w_low >= 0, r_ok = w_high <= W - 1;
                v[k][0] = msda_tap(base, (long)h_low * W + w_low, stride, t_ok && l_ok);
                v[k][1] = msda_tap(base, (long)h_low * W + w_high, stride, t_ok && r_ok);
                v[k][2] = msda_tap(base, (long)h_high * W + w_low, stride, b_ok && l_ok);
                v[k][3] = msda_tap(base, (long)h_high * W + w_high, stride, b_ok && r_ok);
                tw[k][0] = hh * hw * at;
                tw[k][1] = hh * lw * at;
                tw[k][2] = lh * hw * at;
                tw[k][3] = lh * lw * at;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x += tw[k][0] * v[k][0].x + tw[k][1] * v[k][1].x + tw[k][2] * v[k][2].x + tw[k][3] * v[k][3].x;
                acc.y += tw[k][0] * v[k][0].y + tw[k][1] * v[k][1].y + tw[k][2] * v[k][2].y + tw[k][3] * v[k][3].y;
                acc.z += tw[k][0] * v[k][0].z + tw[k][1] * v[k][1].z + tw[k][2] * v[k][2].z + tw[k][3] * v[k][3].z;
                acc.w += tw[k][0] * v[k][0].w + tw[k][1] * v[k][1].w + tw[k][2] * v[k][2].w + tw[k][3] * v[k][3].w;
            }
        }
    }
    *reinterpret_cast<rac_f4 *>(a.out + (item0 + grp) * 64 + c4 * 4) = acc;
}

template <typename FT>
__device__ __forceinline__ float msda_ld1(const FT *p);
template <>
__device__ __forceinline__ float msda_ld1<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float msda_ld1<unsigned short>(const unsigned short *p) { return rac_bf16_to_f32(*p); }

template <typename FT>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(const MsdaArgs a)
{
    const long total = (long)a.bs * a.Q * a.heads * a.dim;
    const int LP = a.L * a.P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % a.dim);
        const long item = idx / a.dim;
        const int h = (int)(item % a.heads);
        const int b = (int)(item / ((long)a.heads * a.Q));
        const float *lp = a.loc + item * LP * 2;
        const float *ap = a.attn + item * LP;
        const int stride = a.heads * a.dim;
        float acc = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const int H = a.H[l], W = a.W[l];
            const FT *base = (const FT *)a.value + (((size_t)b * a.keys + a.start[l]) * a.heads + h) * a.dim + c;
            for (int p = 0; p < a.P; ++p) {
                const float x = lp[(l * a.P + p) * 2], y = lp[(l * a.P + p) * 2 + 1];
                const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W))
                    continue;
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                if (h_low >= 0 && w_low >= 0) v1 = msda_ld1(base + ((long)h_low * W + w_low) * stride);
                if (h_low >= 0 && w_high <= W - 1) v2 = msda_ld1(base + ((long)h_low * W + w_high) * stride);
                if (h_high <= H - 1 && w_low >= 0) v3 = msda_ld1(base + ((long)h_high * W + w_low) * stride);
                if (h_high <= H - 1 && w_high <= W - 1) v4 = msda_ld1(base + ((long)h_high * W + w_high) * stride);
                acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * ap[l * a.P + p];
            }
        }
        a.out[idx] = acc;
    }
}

extern "C" int rac_msda_fwd(const void *value, const int64_t *shapes, const int64_t *starts,
                            const float *loc, const float *attn, float *out, int bs, int keys,
                            int heads, int dim, int Q, int L, int P, int dtype, void *stream)
{
    RAC_CHECK_ARG(L >= 1 && L <= RAC_MAX_LEVELS, "rac_msda_fwd: L=%d out of [1,%d]", L, RAC_MAX_LEVELS);
    RAC_CHECK_ARG(bs >= 0 && Q >= 0 && heads >= 1 && dim >= 1 && keys >= 0 && P >= 0,
                  "rac_msda_fwd: bad sizes bs=%d keys=%d heads=%d dim=%d Q=%d P=%d", bs, keys, heads, dim, Q, P);
    RAC_CHECK_ARG(dtype == RAC_F32 || dtype == RAC_BF16, "rac_msda_fwd: dtype %d", dtype);
    if (bs == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(value && shapes && starts && loc && attn && out, "rac_msda_fwd: null pointer");
    MsdaArgs a;
    for (int l = 0; l < RAC_MAX_LEVELS; ++l) {
        a.H[l] = a.W[l] = 1;
        a.start[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = starts[l];
        RAC_CHECK_ARG(h >= 1 && w >= 1 && st >= 0 && st + h * w <= keys,
                      "rac_msda_fwd: level %d (h=%ld,w=%ld,start=%ld) exceeds keys=%d", l, (long)h, (long)w, (long)st, keys);
        a.H[l] = (int)h;
        a.W[l] = (int)w;
        a.start[l] = (long)st;
    }
    a.value = value; a.loc = loc; a.attn = attn; a.out = out;
    a.bs = bs; a.keys = keys; a.heads = heads; a.dim = dim; a.Q = Q; a.L = L; a.P = P;
    a.blocks_per_b = (Q * heads + MSDA_ITEMS - 1) / MSDA_ITEMS;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)MSDA_ITEMS * L * P * 3 * sizeof(float);
    if (dim == 64 && P >= 1 && lds <= 48 * 1024) {
        const int nb = 8 * ((bs + 7) / 8) * a.blocks_per_b;
        if (dtype == RAC_F32)
            hipLaunchKernelGGL(msda_fwd_d64_kernel<float>, dim3(nb), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL(msda_fwd_d64_kernel<unsigned short>, dim3(nb), dim3(256), lds, st, a);
    } else {
        const long total = (long)bs * Q * heads * dim;
        const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        if (dtype == RAC_F32)
            hipLaunchKernelGGL(msda_fwd_generic_kernel<float>, dim3(nb), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL(msda_fwd_generic_kernel<unsigned short>, dim3(nb), dim3(256), 0, st, a);
    }
    return rac_launch_status("rac_msda_fwd");
}
