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
// The bilinear footprint of a point is computed once (one thread per point, into an LDS tap table of four byte
// offsets + four weights), the taps are buffer-descriptor loads whose range check zero-fills taps outside the map,
// the accumulation is packed FMAs; blocks are mapped so that all workgroups of one BEV frame (one `bs` index,
// 16.8 MB of value) run on one XCD.
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
    unsigned value_bytes;   // size of the value buffer (the buffer descriptor's range; dim = 64 path)
};

#define MSDA_ITEMS 16 /* items per 256-thread workgroup: 4 waves x 4 sixteen-lane groups */

#define MSDA_TAP_OUTSIDE 0x80000000u   /* tap offset past the end of the value buffer: the buffer load returns zeros */
typedef float msda_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int msda_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int msda_u4 __attribute__((ext_vector_type(4)));

// Four channels of one tap through a buffer descriptor: its range check stands in for the four branches of the bilinear
// footprint (a tap outside the map carries the offset MSDA_TAP_OUTSIDE and reads as zero).
template <typename FT>
__device__ __forceinline__ rac_f4 msda_tap(__amdgpu_buffer_rsrc_t rsrc, unsigned off);
template <>
__device__ __forceinline__ rac_f4 msda_tap<float>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    return __builtin_bit_cast(rac_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
}
template <>
__device__ __forceinline__ rac_f4 msda_tap<unsigned short>(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    const msda_u2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);    // 4 x bf16
    return (rac_f4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                    __uint_as_float(r.y & 0xffff0000u)};
}

template <typename FT>
__global__ __launch_bounds__(256) void msda_fwd_d64_kernel(const MsdaArgs a)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, c4 = lane & 15;
    const int grp = tid >> 4;  // 0..15: item within the workgroup
    const int LP = a.L * a.P;
    const int LPp = (LP + 3) & ~3;   // an item's tap list, padded to whole batches of 4 with zero-weight outside taps

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int b = xcd + 8 * (j / a.blocks_per_b);
    if (b >= a.bs)
        return;
    const int per_b = a.Q * a.heads;
    const int i0 = (j % a.blocks_per_b) * MSDA_ITEMS;
    const int nitems = min(MSDA_ITEMS, per_b - i0);
    const size_t item0 = (size_t)b * per_b + i0;

    // tap table [items][LPp][8]: per point 4 tap byte offsets into the value buffer (MSDA_TAP_OUTSIDE = outside the map) and 4
    // bilinear weights with the attention weight folded in -- one thread per point builds it from the op's location / weight
    // rows, instead of each of the 16 lanes that gather the point
    float *stab = smem;
    const unsigned key_bytes = (unsigned)(a.heads * 64 * sizeof(FT));       // one key: heads x 64 channels
    for (int i = tid; i < nitems * LPp; i += 256) {
        const int it = i / LPp, lp = i - it * LPp;
        msda_u4 off = {MSDA_TAP_OUTSIDE, MSDA_TAP_OUTSIDE, MSDA_TAP_OUTSIDE, MSDA_TAP_OUTSIDE};
        rac_f4 w4 = {0.f, 0.f, 0.f, 0.f};
        if (lp < LP) {
            const int l = lp / a.P;
            const int H = a.H[l], W = a.W[l];
            const float *gl = a.loc + ((item0 + it) * LP + lp) * 2;
            const float x = gl[0], y = gl[1];
            const float at = a.attn[(item0 + it) * LP + lp];
            const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = in && h_low >= 0, b_ok = in && h_high <= H - 1;
            const bool l_ok = w_low >= 0, r_ok = w_high <= W - 1;
            const int h = (i0 + it) % a.heads;
            const unsigned base = ((unsigned)((long)b * a.keys + a.start[l]) * (unsigned)a.heads + (unsigned)h) * (unsigned)(64 * sizeof(FT));
            off.x = t_ok && l_ok ? base + (unsigned)(h_low * W + w_low) * key_bytes : MSDA_TAP_OUTSIDE;
            off.y = t_ok && r_ok ? base + (unsigned)(h_low * W + w_high) * key_bytes : MSDA_TAP_OUTSIDE;
            off.z = b_ok && l_ok ? base + (unsigned)(h_high * W + w_low) * key_bytes : MSDA_TAP_OUTSIDE;
            off.w = b_ok && r_ok ? base + (unsigned)(h_high * W + w_high) * key_bytes : MSDA_TAP_OUTSIDE;
            w4 = (rac_f4){hh * hw * at, hh * lw * at, lh * hw * at, lh * lw * at};
        }
        *reinterpret_cast<msda_u4 *>(stab + i * 8) = off;
        *reinterpret_cast<rac_f4 *>(stab + i * 8 + 4) = w4;
    }
    __syncthreads();
    if (grp >= nitems)
        return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0, a.value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(c4 * 4 * sizeof(FT));
    const float *e = stab + grp * LPp * 8;      // same address for the 16 lanes of the group
    rac_acc4 acc4 = rac_acc4_zero();
    for (int p0 = 0; p0 < LPp; p0 += 4) {       // 4 points = 16 taps in flight per lane; per tap one add, one load, two packed FMAs
        rac_f4 v[4][4], tw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const msda_u4 o = *reinterpret_cast<const msda_u4 *>(e + (p0 + k) * 8);
            tw[k] = *reinterpret_cast<const rac_f4 *>(e + (p0 + k) * 8 + 4);
            v[k][0] = msda_tap<FT>(rsrc, o.x + lane_off);
            v[k][1] = msda_tap<FT>(rsrc, o.y + lane_off);
            v[k][2] = msda_tap<FT>(rsrc, o.z + lane_off);
            v[k][3] = msda_tap<FT>(rsrc, o.w + lane_off);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w4[4] = {tw[k].x, tw[k].y, tw[k].z, tw[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                rac_tap_fma(acc4, v[k][c].x, v[k][c].y, v[k][c].z, v[k][c].w, w4[c]);
            }
        }
    }
    rac_f4 r;
    rac_acc4_get(acc4, r.x, r.y, r.z, r.w);
    *reinterpret_cast<rac_f4 *>(a.out + (item0 + grp) * 64 + c4 * 4) = r;
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
    const size_t lds = (size_t)MSDA_ITEMS * ((L * P + 3) & ~3) * 8 * sizeof(float);
    const size_t value_bytes = (size_t)bs * keys * heads * dim * (dtype == RAC_F32 ? 4 : 2);
    a.value_bytes = (unsigned)value_bytes;
    if (dim == 64 && P >= 1 && lds <= 64 * 1024 && value_bytes < (size_t)MSDA_TAP_OUTSIDE) {   // (larger values: 31-bit tap offsets do not reach)
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
