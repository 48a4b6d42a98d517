/*
 * racformer_hip.h -- C-ABI of libracformer_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for RaCFormer's query-decoder hot path.  Plain pointers and sizes, no torch
 * types, no exceptions across the ABI.  Every entry point returns 0 on success, a negative
 * RAC_E_* code for argument errors (nothing launched) or a positive hipError_t value if the HIP
 * runtime refused the launch; rac_last_error() gives the message for the calling thread.
 * All device pointers must be valid on the current HIP device; kernels are enqueued on `stream`
 * (a hipStream_t; NULL = the legacy default stream, which is what the reference's launcher used,
 * models/csrc/msmv_sampling/msmv_sampling_forward.cu:359).  No allocation, no synchronisation:
 * every call is hipGraph-capturable.
 *
 * Reference interfaces replaced (paths relative to the reference root):
 *   rac_msmv_fwd      <- _ms_deform_attn_cuda_{c45,c2345,c23456}_forward
 *                        models/csrc/msmv_sampling/msmv_sampling.cpp:132-184 (+ :186-236, :238-300),
 *                        pybind at :499-506; Python caller models/csrc/wrapper.py:78-153
 *   rac_msda_fwd      <- mmcv-full 1.6.0 `_ext.ms_deform_attn_forward`, call site
 *                        models/multi_scale_deformable_attn_function.py:118-124
 *   rac_regroup_fwd   <- the channel-last regroup in RaCFormerTransformerDecoder.forward,
 *                        models/racformer_transformer.py:112-124
 */
#ifndef RACFORMER_HIP_H
#define RACFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAC_ABI_VERSION 1
#define RAC_MAX_LEVELS 8
#define RAC_MAX_POINTS 128 /* same limit as the reference, msmv_sampling_forward.cu:21 */

enum { RAC_F32 = 0, RAC_BF16 = 1 };

enum {
    RAC_E_ARG = -1,      /* bad size / null pointer */
    RAC_E_UNSUPPORTED = -2,
};

/* Output layouts of rac_msmv_fwd. */
enum {
    RAC_OUT_SQCP = 0,  /* [S,Q,C,P]      -- the reference op's layout (msmv_sampling.cpp:170)      */
    RAC_OUT_BQGTPC = 1 /* [B,Q,G,T*P,C]  -- what sampling_4d returns after its regroup
                          (sparsebev_sampling.py:128-131), written directly; slot s=(b*T+t)*G+g */
};

int rac_abi_version(void);
const char *rac_last_error(void);

/* Multi-scale multi-view sampling, forward.
 *   feats[l] : device ptr, [S, N, H_l, W_l, C] channel-last, dtype `dtype`
 *   hw       : HOST ptr, L x (H_l, W_l) int32
 *   loc      : device f32 [S,Q,P,3] = (u, v, view/(N-1)), u,v normalised to [0,1]
 *   w        : device f32 [S,Q,P,L] per-level weights
 *   out      : device f32, layout `out_layout`; every element is written (no pre-zeroing needed)
 *   T,G      : only used by RAC_OUT_BQGTPC (S must be a multiple of T*G); pass 1,1 otherwise
 * out[s,q,c,p] = sum_l w[s,q,p,l] * bilinear0(feats[l][s, round(view*(N-1))], u*(W_l-1), v*(H_l-1)) */
int rac_msmv_fwd(const void *const *feats, const int32_t *hw, int L, const float *loc,
                 const float *w, float *out, int S, int N, int Q, int P, int C, int dtype,
                 int out_layout, int T, int G, void *stream);

/* Multi-scale deformable attention, forward (Deformable-DETR semantics, align_corners=False).
 *   value  : device, [bs, keys, heads, dim], dtype `dtype`
 *   shapes : HOST int64 [L,2] (h,w);  starts: HOST int64 [L]
 *   loc    : device f32 [bs,Q,heads,L,P,2] (x,y) in [0,1];  attn: device f32 [bs,Q,heads,L,P]
 *   out    : device f32 [bs,Q,heads*dim] */
int rac_msda_fwd(const void *value, const int64_t *shapes, const int64_t *starts, const float *loc,
                 const float *attn, float *out, int bs, int keys, int heads, int dim, int Q, int L,
                 int P, int dtype, void *stream);

/* Pyramid regroup: in [B, T*N, G*C, H, W] f32 -> out [B*T*G, N, H, W, C] (dtype out_dtype). */
int rac_regroup_fwd(const float *in, void *out, int B, int T, int N, int G, int C, int H, int W,
                    int out_dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RACFORMER_HIP_H */
