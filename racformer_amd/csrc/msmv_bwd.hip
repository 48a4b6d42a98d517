// msmv_bwd.hip -- multi-scale multi-view sampling, backward, for gfx950.
//
// Replaces ms_deformable_col2im_gpu_kernel_gm_{c45,c2345,c23456}
// (models/csrc/msmv_sampling/msmv_sampling_backward.cu:29-440) behind rac_msmv_bwd.  Same math:
//   grad_feat[l][tap]  += w_tap * grad_out * weight_l                       (scatter, float atomics)
//   grad_weight[p,l]    = sum_c grad_out[c] * bilinear_l[c]
//   grad_loc[p] (u, v)  = sum_l (W_l-1 | H_l-1) * weight_l * sum_c grad_out[c] * d bilinear_l[c] / d(w|h)
//   grad_loc[p] view    = 0   (the kernel treats the view as an index, :139)
// The reference uses one thread per (row, channel, point) and atomics for ALL THREE gradients; here a
// 16-lane group owns a point (4 channels per lane), so the channel sums of grad_weight / grad_loc are a
// butterfly inside the group with a single writer per element (deterministic).  Only the feature
// scatter needs atomics (several points may hit one pixel); each group-instruction adds one contiguous
// 256-byte pixel row, the shape the memory-side atomic units run at full rate on (MI355X_MICROARCH.md).
// The caller zero-fills grad_feat; grad_loc / grad_weight are fully overwritten.
#include "rac_common.h"

struct MsmvBwdArgs {
    const void *feat[RAC_MAX_LEVELS];
    float *gfeat[RAC_MAX_LEVELS];
    int H[RAC_MAX_LEVELS];
    int W[RAC_MAX_LEVELS];
    const float *grad_out;  // [S,Q,C,P]
    const float *loc;       // [S,Q,P,3]
    const float *w;         // [S,Q,P,L]
    float *gloc;            // [S,Q,P,3]
    float *gw;              // [S,Q,P,L]
    int L, S, N, Q, P, C;
};

__device__ __forceinline__ float mb_group_sum16(float v)
{
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 16);
    return v;
}

__device__ __forceinline__ void mb_atomic_add4(float *p, rac_f4 v)
{
    atomicAdd(p, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
    atomicAdd(p + 3, v.w);
}

// C = 64, fp32 features: one 16-lane group per (row, point)
template <int L>
__global__ __launch_bounds__(256) void msmv_bwd_c64_kernel(const MsmvBwdArgs a)
{
    const int lane16 = threadIdx.x & 15;
    const long pt = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;  // (s*Q+q)*P + p
    const long npts = (long)a.S * a.Q * a.P;
    const bool act = pt < npts;
    const long ptc = act ? pt : 0;
    const int p = (int)(ptc % a.P);
    const long row = ptc / a.P;
    const int s = (int)(row / a.Q);
    const float *lp = a.loc + ptc * 3;
    const float *wp = a.w + ptc * L;
    const float lu = lp[0], lv = lp[1];
    int view = (int)roundf(lp[2] * (float)(a.N - 1));
    view = min(max(view, 0), a.N - 1);
    // grad_out[s,q,c,p] for this lane's 4 channels
    rac_f4 g;
    {
        const float *go = a.grad_out + (row * 64 + lane16 * 4) * a.P + p;
        g.x = act ? go[0] : 0.f;
        g.y = act ? go[a.P] : 0.f;
        g.z = act ? go[2 * (size_t)a.P] : 0.f;
        g.w = act ? go[3 * (size_t)a.P] : 0.f;
    }
    float gu = 0.f, gv = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int H = a.H[l], W = a.W[l];
        const float h_im = lv * (float)(H - 1), w_im = lu * (float)(W - 1);
        const bool in = act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const size_t map = ((size_t)s * a.N + view) * H * W * 64 + lane16 * 4;
        const float *base = (const float *)a.feat[l] + map;
        float *gbase = a.gfeat[l] + map;
        const float wl = wp[l];
        const bool ok1 = in && h_low >= 0 && w_low >= 0, ok2 = in && h_low >= 0 && w_high <= W - 1;
        const bool ok3 = in && h_high <= H - 1 && w_low >= 0, ok4 = in && h_high <= H - 1 && w_high <= W - 1;
        const size_t o1 = ((size_t)h_low * W + w_low) * 64, o2 = ((size_t)h_low * W + w_high) * 64;
        const size_t o3 = ((size_t)h_high * W + w_low) * 64, o4 = ((size_t)h_high * W + w_high) * 64;
        const rac_f4 z = {0.f, 0.f, 0.f, 0.f};
        const rac_f4 v1 = ok1 ? rac_ld4(base + o1) : z, v2 = ok2 ? rac_ld4(base + o2) : z;
        const rac_f4 v3 = ok3 ? rac_ld4(base + o3) : z, v4 = ok4 ? rac_ld4(base + o4) : z;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        // feature scatter
        const rac_f4 tg = {g.x * wl, g.y * wl, g.z * wl, g.w * wl};
        if (ok1) mb_atomic_add4(gbase + o1, (rac_f4){w1 * tg.x, w1 * tg.y, w1 * tg.z, w1 * tg.w});
        if (ok2) mb_atomic_add4(gbase + o2, (rac_f4){w2 * tg.x, w2 * tg.y, w2 * tg.z, w2 * tg.w});
        if (ok3) mb_atomic_add4(gbase + o3, (rac_f4){w3 * tg.x, w3 * tg.y, w3 * tg.z, w3 * tg.w});
        if (ok4) mb_atomic_add4(gbase + o4, (rac_f4){w4 * tg.x, w4 * tg.y, w4 * tg.z, w4 * tg.w});
        // per-channel bilinear value and its derivatives w.r.t. (h_im, w_im); invalid taps hold zeros
#define MB_DOT(fx)                                                                                   \
    ((fx(v1.x, v2.x, v3.x, v4.x)) * g.x + (fx(v1.y, v2.y, v3.y, v4.y)) * g.y + (fx(v1.z, v2.z, v3.z, v4.z)) * g.z + \
     (fx(v1.w, v2.w, v3.w, v4.w)) * g.w)
#define MB_VAL(a1, a2, a3, a4) (w1 * (a1) + w2 * (a2) + w3 * (a3) + w4 * (a4))
#define MB_DH(a1, a2, a3, a4) (-hw * (a1) - lw * (a2) + hw * (a3) + lw * (a4))
#define MB_DW(a1, a2, a3, a4) (-hh * (a1) + hh * (a2) - lh * (a3) + lh * (a4))
        const float sv = mb_group_sum16(MB_DOT(MB_VAL));
        const float sh = mb_group_sum16(MB_DOT(MB_DH));
        const float sw_ = mb_group_sum16(MB_DOT(MB_DW));
#undef MB_DOT
#undef MB_VAL
#undef MB_DH
#undef MB_DW
        if (act && lane16 == 0)
            a.gw[ptc * L + l] = in ? sv : 0.f;
        gu += (float)(W - 1) * sw_ * wl;
        gv += (float)(H - 1) * sh * wl;
    }
    if (act && lane16 == 0) {
        a.gloc[ptc * 3] = gu;
        a.gloc[ptc * 3 + 1] = gv;
        a.gloc[ptc * 3 + 2] = 0.f;
    }
}

// generic path: any C, any L <= 8.  One thread per (row, point); serial over channels (no atomics for
// grad_loc / grad_weight either, atomics for the feature scatter).
__global__ __launch_bounds__(256) void msmv_bwd_generic_kernel(const MsmvBwdArgs a)
{
    const long npts = (long)a.S * a.Q * a.P;
    for (long pt = (long)blockIdx.x * blockDim.x + threadIdx.x; pt < npts; pt += (long)gridDim.x * blockDim.x) {
        const int p = (int)(pt % a.P);
        const long row = pt / a.P;
        const int s = (int)(row / a.Q);
        const float *lp = a.loc + pt * 3;
        const float *wp = a.w + pt * a.L;
        const float lu = lp[0], lv = lp[1];
        int view = (int)roundf(lp[2] * (float)(a.N - 1));
        view = min(max(view, 0), a.N - 1);
        float gu = 0.f, gv = 0.f;
        for (int l = 0; l < a.L; ++l) {
            const int H = a.H[l], W = a.W[l];
            const float h_im = lv * (float)(H - 1), w_im = lu * (float)(W - 1);
            float sv = 0.f, sh = 0.f, sw_ = 0.f;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const size_t map = ((size_t)s * a.N + view) * H * W * a.C;
                const float *base = (const float *)a.feat[l] + map;
                float *gbase = a.gfeat[l] + map;
                const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
                const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
                const size_t o1 = ((size_t)h_low * W + w_low) * a.C, o2 = ((size_t)h_low * W + w_high) * a.C;
                const size_t o3 = ((size_t)h_high * W + w_low) * a.C, o4 = ((size_t)h_high * W + w_high) * a.C;
                for (int c = 0; c < a.C; ++c) {
                    const float g = a.grad_out[(row * a.C + c) * a.P + p];
                    const float tg = g * wp[l];
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
            a.gw[pt * a.L + l] = sv;
            gu += (float)(W - 1) * sw_ * wp[l];
            gv += (float)(H - 1) * sh * wp[l];
        }
        a.gloc[pt * 3] = gu;
        a.gloc[pt * 3 + 1] = gv;
        a.gloc[pt * 3 + 2] = 0.f;
    }
}

extern "C" int rac_msmv_bwd(const float *grad_out, const void *const *feats, const int32_t *hw, int L, const float *loc,
                            const float *w, void *const *grad_feats, float *grad_loc, float *grad_w, int S, int N, int Q,
                            int P, int C, void *stream)
{
    RAC_CHECK_ARG(L >= 1 && L <= RAC_MAX_LEVELS, "rac_msmv_bwd: L=%d out of [1,%d]", L, RAC_MAX_LEVELS);
    RAC_CHECK_ARG(S >= 0 && Q >= 0 && N >= 1 && C >= 1, "rac_msmv_bwd: bad sizes S=%d N=%d Q=%d C=%d", S, N, Q, C);
    RAC_CHECK_ARG(P >= 0 && P <= RAC_MAX_POINTS, "rac_msmv_bwd: num_point exceed limits (P=%d > %d)", P, RAC_MAX_POINTS);
    if (S == 0 || Q == 0 || P == 0)
        return 0;
    RAC_CHECK_ARG(grad_out && feats && hw && loc && w && grad_feats && grad_loc && grad_w, "rac_msmv_bwd: null pointer");
    MsmvBwdArgs a;
    for (int l = 0; l < RAC_MAX_LEVELS; ++l) {
        a.feat[l] = nullptr;
        a.gfeat[l] = nullptr;
        a.H[l] = a.W[l] = 1;
    }
    for (int l = 0; l < L; ++l) {
        RAC_CHECK_ARG(feats[l] && grad_feats[l] && hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "rac_msmv_bwd: level %d", l);
        a.feat[l] = feats[l];
        a.gfeat[l] = (float *)grad_feats[l];
        a.H[l] = hw[2 * l];
        a.W[l] = hw[2 * l + 1];
    }
    a.grad_out = grad_out; a.loc = loc; a.w = w; a.gloc = grad_loc; a.gw = grad_w;
    a.L = L; a.S = S; a.N = N; a.Q = Q; a.P = P; a.C = C;
    hipStream_t st = (hipStream_t)stream;
    const long npts = (long)S * Q * P;
    if (C == 64 && (L == 2 || L == 4 || L == 5)) {
        const unsigned nb = (unsigned)((npts * 16 + 255) / 256);
        if (L == 2) hipLaunchKernelGGL(msmv_bwd_c64_kernel<2>, dim3(nb), dim3(256), 0, st, a);
        else if (L == 4) hipLaunchKernelGGL(msmv_bwd_c64_kernel<4>, dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(msmv_bwd_c64_kernel<5>, dim3(nb), dim3(256), 0, st, a);
    } else {
        const unsigned nb = (unsigned)((npts + 255) / 256 > 4096 ? 4096 : (npts + 255) / 256);
        hipLaunchKernelGGL(msmv_bwd_generic_kernel, dim3(nb), dim3(256), 0, st, a);
    }
    return rac_launch_status("rac_msmv_bwd");
}
