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
// scatter needs atomics (several points may hit one pixel).  Round 4: a lane's four channels are c, 16 + c, 32 + c, 48 + c
// (c = lane of the group), NOT 4c .. 4c+3: a float atomic is one dword per lane, so with consecutive channels per lane
// every atomic instruction touched all sixteen 64-byte lines of its four pixel rows at a quarter of their width; now an
// instruction adds four whole 64-byte segments -- the request shape the memory-side atomic units take at full rate
// (MI355X_MICROARCH.md: a 256-byte wave-instruction leaves L2 as four 64-byte atomic requests).
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

// C = 64, fp32 features: one 16-lane group per (row, point); lane c of the group owns channels c + 16 j, j = 0..3
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
    float g[4];
    {
        const float *go = a.grad_out + (row * 64 + lane16) * a.P + p;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            g[j] = act ? go[(size_t)(16 * j) * a.P] : 0.f;
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
        const size_t map = ((size_t)s * a.N + view) * H * W * 64 + lane16;
        const float *base = (const float *)a.feat[l] + map;
        float *gbase = a.gfeat[l] + map;
        const float wl = wp[l];
        const bool ok[4] = {in && h_low >= 0 && w_low >= 0, in && h_low >= 0 && w_high <= W - 1,
                            in && h_high <= H - 1 && w_low >= 0, in && h_high <= H - 1 && w_high <= W - 1};
        const size_t o[4] = {((size_t)h_low * W + w_low) * 64, ((size_t)h_low * W + w_high) * 64,
                             ((size_t)h_high * W + w_low) * 64, ((size_t)h_high * W + w_high) * 64};
        const float tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        const float dh[4] = {-hw, -lw, hw, lw}, dw[4] = {-hh, hh, -lh, lh};
        float sv = 0.f, sh = 0.f, sw_ = 0.f;      // this lane's share of sum_c grad_out[c] * {value, d/dh, d/dw}[c]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = ok[t] ? base[o[t] + 16 * j] : 0.f;
            // feature scatter: per instruction the group adds 16 consecutive floats of the tap's pixel row
            if (ok[t]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    atomicAdd(gbase + o[t] + 16 * j, tw[t] * (g[j] * wl));
            }
            const float dot = (v[0] * g[0] + v[1] * g[1]) + (v[2] * g[2] + v[3] * g[3]);
            sv += tw[t] * dot;
            sh += dh[t] * dot;
            sw_ += dw[t] * dot;
        }
        sv = mb_group_sum16(sv);
        sh = mb_group_sum16(sh);
        sw_ = mb_group_sum16(sw_);
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
