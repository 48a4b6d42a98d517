/*
 * oracle/gather_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the two gather operators on RaCFormer's decoder hot path, used
 * only as the parity checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing in racformer_amd/ links, loads or calls this file.
 *
 * Parity status: PINNED in this container against outputs of the reference's own CPU path
 * (tests/golden/msmv_small.npz from models/csrc/wrapper.py:15-39 msmv_sampling_pytorch;
 * tests/golden/msda_small.npz from mmcv-1.6.0's multi_scale_deformable_attn_pytorch as
 * restated on F.grid_sample in tests/golden/ref_loader.py -- the compiled mmcv kernel is a
 * third-party dependency, mmcv-full==1.6.0 (reference README.md:44), absent from the tree).
 *
 * Semantics restated (reference file:line):
 *   msmv:  models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-164 (per-(slot,query,channel)
 *          loop over points; view = round(loc_v*(N-1)); per level h_im = v*(H-1), w_im = u*(W-1),
 *          range guard :126; accumulation order c2..c5) and :27-73 (4-tap bilinear, each tap
 *          bounds-checked, zero padding, channel-last strides).
 *   msda:  Deformable-DETR lineage kept as comments at msmv_sampling_forward.cu:119-120
 *          (h_im = loc_h*H - 0.5: align_corners=False), same guard and tap checks; call-site
 *          contract models/multi_scale_deformable_attn_function.py:93-128.
 *
 * Two instances of the same text: float (the checker: the reference's arithmetic type) and double (`*_f64`: the
 * ARBITER of tools/fp64_arbiter.py -- the same algorithm evaluated in float64, against which both the reference's
 * fp32 fixtures and the GPU's fp32 results are measured).  The file includes itself once per type.
 */
#ifndef GATHER_REF_INSTANCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#define GATHER_REF_INSTANCE 1
#define REAL float
#define FN(name) name
#define RFLOOR floorf
#define RROUND roundf
#include "gather_ref.c"
#undef REAL
#undef FN
#undef RFLOOR
#undef RROUND
#define REAL double
#define FN(name) name##_f64
#define RFLOOR floor
#define RROUND round
#include "gather_ref.c"
#else

static inline REAL FN(bilinear_cl)(const REAL *base, int H, int W, int stride_px, REAL h, REAL w, int c)
{
    /* base points at [y=0,x=0,c=0] of one channel-last map; stride_px = floats per pixel */
    const int h_low = (int)RFLOOR(h);
    const int w_low = (int)RFLOOR(w);
    const int h_high = h_low + 1;
    const int w_high = w_low + 1;
    const REAL lh = h - (REAL)h_low;
    const REAL lw = w - (REAL)w_low;
    const REAL hh = (REAL)1 - lh, hw = (REAL)1 - lw;
    REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0)
        v1 = base[((size_t)h_low * W + w_low) * stride_px + c];
    if (h_low >= 0 && w_high <= W - 1)
        v2 = base[((size_t)h_low * W + w_high) * stride_px + c];
    if (h_high <= H - 1 && w_low >= 0)
        v3 = base[((size_t)h_high * W + w_low) * stride_px + c];
    if (h_high <= H - 1 && w_high <= W - 1)
        v4 = base[((size_t)h_high * W + w_high) * stride_px + c];
    const REAL w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* feats[l]: [S,N,H_l,W_l,C]; hw: L x (H,W); loc: [S,Q,P,3] (u,v,view/(N-1)); w: [S,Q,P,L];
 * out: [S,Q,C,P] (reference layout, wrapper.py:145-153).  Returns 0, or -1 on bad sizes. */
int FN(oracle_msmv_fwd)(const REAL *const *feats, const int32_t *hw, int L, const REAL *loc, const REAL *w, REAL *out, int S,
                        int N, int Q, int P, int C)
{
    if (L < 1 || L > 8 || S < 0 || N < 1 || Q < 0 || P < 0 || C < 1)
        return -1;
    const long rows = (long)S * Q;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int s = (int)(r / Q);
        for (int p = 0; p < P; ++p) {
            const REAL *lp = loc + ((size_t)r * P + p) * 3;
            const REAL *wp = w + ((size_t)r * P + p) * L;
            const REAL lu = lp[0], lv = lp[1];
            const int view = (int)RROUND(lp[2] * (REAL)(N - 1));
            for (int c = 0; c < C; ++c) {
                REAL acc = 0;
                for (int l = 0; l < L; ++l) {
                    const int H = hw[2 * l], W = hw[2 * l + 1];
                    const REAL h_im = lv * (REAL)(H - 1);
                    const REAL w_im = lu * (REAL)(W - 1);
                    if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                        const REAL *base = feats[l] + ((size_t)s * N + view) * H * W * C;
                        acc += FN(bilinear_cl)(base, H, W, C, h_im, w_im, c) * wp[l];
                    }
                }
                out[((size_t)r * C + c) * P + p] = acc;
            }
        }
    }
    return 0;
}

/* value: [bs,keys,heads,dim]; shapes: [L,2] (h,w); starts: [L]; loc: [bs,Q,heads,L,P,2] (x,y);
 * attn: [bs,Q,heads,L,P]; out: [bs,Q,heads*dim]. */
int FN(oracle_msda_fwd)(const REAL *value, const int64_t *shapes, const int64_t *starts, const REAL *loc, const REAL *attn,
                        REAL *out, int bs, int keys, int heads, int dim, int Q, int L, int P)
{
    if (L < 1 || heads < 1 || dim < 1)
        return -1;
    const long rows = (long)bs * Q * heads;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int h = (int)(r % heads);
        const long bq = r / heads;
        const int b = (int)(bq / Q);
        const REAL *lp = loc + (size_t)r * L * P * 2;
        const REAL *ap = attn + (size_t)r * L * P;
        for (int c = 0; c < dim; ++c) {
            REAL acc = 0;
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const REAL *base = value + (((size_t)b * keys + starts[l]) * heads + h) * dim;
                for (int p = 0; p < P; ++p) {
                    const REAL x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
                    const REAL h_im = y * (REAL)H - (REAL)0.5;
                    const REAL w_im = x * (REAL)W - (REAL)0.5;
                    if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                        acc += FN(bilinear_cl)(base, H, W, heads * dim, h_im, w_im, c) * ap[l * P + p];
                }
            }
            out[(size_t)bq * heads * dim + h * dim + c] = acc;
        }
    }
    (void)keys;
    return 0;
}
#endif /* GATHER_REF_INSTANCE */
