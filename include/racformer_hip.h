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
 *   rac_box_prep_fwd  <- decode_bbox(theta_d2xy_coods(.)) models/bbox/utils.py:66-90 (shared prologue)
 *   rac_sampling4d_fwd<- RaCFormerSampling.inner_forward + sampling_4d + msmv op, fused
 *                        models/racformer_transformer.py:361-419, models/sparsebev_sampling.py:28-134
 *   rac_msmv_bwd / rac_msda_bwd <- the two operators' backward entry points (row f4)
 *   rac_bev_pool_v2_fwd/_bwd <- bev_pool_v2_ext (models/csrc/bev_pool_v2/src/bev_pool.cpp:40-111), row f2
 *   rac_add_ln_fwd    <- residual add + nn.LayerNorm (+ReLU) groups, models/racformer_transformer.py:170-258
 *   rac_layer_boundary_fwd <- rac_refine_fwd + rac_box_prep_fwd + rac_pe_head_fwd of consecutive layers, one launch
 *   rac_head_finish_fwd <- nan_to_num of the decoder outputs + box denormalisation, models/racformer_transformer.py:58, models/racformer_head.py:124-131
 *   rac_refine_fwd    <- refine_bbox + velocity scaling + theta_d2xy_coods of the outputs
 *                        models/racformer_transformer.py:230-236,265-269,134
 *   rac_mixing_fwd    <- AdaptiveMixing.inner_forward's matmul / layer_norm / relu chain
 *                        models/racformer_transformer.py:589-603
 *   rac_sasa_fwd      <- ScaleAdaptiveSelfAttention.inner_forward's mask + attention product
 *                        models/racformer_transformer.py:296-335
 *   rac_decode_fwd    <- NMSFreeCoder.decode_single + get_bboxes, models/bbox/coders/nms_free_coder.py:37-88,
 *                        models/racformer_head.py:488-507
 *   rac_outproj_fwd / rac_gemm_split_pack_fwd <- AdaptiveMixing.out_proj (nn.Linear 32768 -> 256), models/racformer_transformer.py:566,606
 *   rac_value_proj_fwd <- BEVSelfAttention.value_proj over the BEV maps, models/bev_self_attention.py:162-174
 *   rac_generator_fwd <- AdaptiveMixing.parameter_generator (nn.Linear 256 -> 65536), models/racformer_transformer.py:565,589
 *   rac_rowgemm_fwd   <- nn.Linear + its preceding add / LayerNorm / ReLU groups, models/racformer_transformer.py:170-177, 243-269
 *   rac_gru_gate_fwd / rac_upsample2x_fwd <- ConvGRUCell.forward's element-wise tail, nn.Upsample
 *                        models/racformer_transformer.py:705-720, :633-636
 *   rac_absmax_fwd / rac_conv_pack_fwd / rac_conv3x3_fwd <- RadarBEVTemporalEncoder.temporal_fusion (nn.Conv2d 3x3)
 *                        models/racformer_transformer.py:631,655
 *   rac_bev_sampling_fwd <- BEVSampling keypoints + BEVSelfAttention's MSDA + frame fusion, fused
 *                        models/racformer_transformer.py:490-529, models/bev_self_attention.py:176-213
 */
#ifndef RACFORMER_HIP_H
#define RACFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAC_ABI_VERSION 8
#define RAC_MAX_LEVELS 8
#define RAC_MAX_POINTS 128 /* same limit as the reference, msmv_sampling_forward.cu:21 */

enum { RAC_F32 = 0, RAC_BF16 = 1, RAC_I16 = 2 /* int16 block storage of a BEV value stream: rac_quant_i16_fwd */ };

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

/* The same regroup for all L levels of the pyramid in ONE launch (16-byte accesses on both sides): ins / outs HOST arrays
 * of L device pointers, hw HOST L x (H, W); C % 4 == 0 and H*W % 4 == 0 on every level. */
int rac_regroup_multi_fwd(int L, const float *const *ins, void *const *outs, const int32_t *hw, int B, int T, int N, int G,
                          int C, int out_dtype, void *stream);

/* Per-query box constants shared by the fused sampling kernels: table[b,q] = (cx, cy, cz, w, l, h,
 * cos yaw, sin yaw) = decode_bbox(theta_d2xy_coods(query_bbox)) (models/bbox/utils.py:66-90), once per
 * query and layer instead of once per keypoint.  query_bbox device f32 [n,10], table device f32 [n,8]. */
int rac_box_prep_fwd(const float *query_bbox, float *table, int num_boxes, const float *pc_range, void *stream);

/* Adaptive 4D sampling of one decoder layer, fully fused (keypoints -> projection -> first-valid-view
 * -> multi-scale gather).  Replaces RaCFormerSampling.inner_forward + sampling_4d + the msmv op
 * (models/racformer_transformer.py:361-419, models/sparsebev_sampling.py:28-134).
 *   feats[l]     : device [B*T*G, N, H_l, W_l, 64] (dtype), hw HOST L x (H,W)
 *   query_bbox   : device f32 [B,Q,10] polar boxes (theta, d, z, log w, log l, log h, sin, cos, vx, vy)
 *   box_table    : device f32 [B,Q,8] written by rac_box_prep_fwd for the same boxes
 *   offsets      : device f32, row (b,q) at offsets + (b*Q+q)*ld_off, G*NP*D*3 values (sampling_offset Linear)
 *   ray_logits   : device f32, rows of D values, stride ld_ray          (ray_points_offset Linear)
 *   scale_logits : device f32, rows of G*T*NP*D*L values, stride ld_scale (scale_weights Linear, softmax over L here)
 *   time_diff    : device f32 [B,T];  lidar2img: device f32 [B,T*N,4,4]
 *   out          : device f32 [B,Q,G,T*NP*D,64]
 *   loc_out,w_out: optional debug outputs [S,Q,P,3] (u,v,view/(N-1)) and [S,Q,P,L] (NULL,NULL to skip)
 *   view_in      : optional device u8 [S,Q,P]: the camera index to sample each point in, INSTEAD of the first valid view
 *                  (sparsebev_sampling.py:97-110).  NULL on the product path; parity tests pass the reference's own
 *                  choices to take the path's one discontinuous step out of a comparison.  loc_out then still reports
 *                  the kernel's OWN choice in its third component (and the imposed view's u, v).
 *   pc_range (6), depth_base (D = torch.linspace(-d_region,d_region,D)): HOST pointers
 *   limits       : NP*D <= 128 points per (slot, query); the kernel takes 8 queries per workgroup, fewer where their tap table
 *                  (queries * NP*D * L * 32 bytes) would exceed 64 KB of LDS (NP*D = 64, L = 4: four); 64 channels per group
 *   compact      : 1 = the variant that sets points without any tap aside (rigs that do not cover the full circle), 0 = plain,
 *                  -1 = decide by the number of cameras (<= 3).  Same results either way. */
int rac_sampling4d_fwd(const void *const *feats, const int32_t *hw, int L, const float *query_bbox,
                       const float *box_table, const float *offsets, const float *ray_logits, const float *scale_logits,
                       const float *time_diff, const float *lidar2img, float *out, float *loc_out,
                       float *w_out, const unsigned char *view_in, int ld_off, int ld_ray, int ld_scale, int B, int T, int N,
                       int G, int Q,
                       int NP, int D, int C, const float *pc_range, const float *depth_base, float d_region,
                       float image_h, float image_w, float eps, int dtype, int compact, void *stream);

/* BEV deformable cross-attention of one decoder layer, fully fused (keypoints -> per-frame
 * deformable attention -> softmax-over-frames fusion).  Replaces BEVSampling.inner_forward's keypoint
 * chain, the MSDA op and the frame fusion (models/racformer_transformer.py:490-529,
 * models/bev_self_attention.py:176-213); output_proj + identity stay outside.
 *   value        : device [B*T, H*W, heads, 64] (dtype) -- hoisted value_proj(bev + pos)
 *   offsets      : rows of heads*NP*D*2 values (stride ld_off); ray_logits rows of D (ld_ray);
 *   scale_logits : rows of heads*NP*D (ld_scale, softmax over the NP*D points of a head here);
 *   queue_logits : rows of T (ld_queue, softmax over frames here)
 *   out          : device f32 [B,Q,heads*64];  loc_out: optional [B,Q,heads,T,NP*D,2] or NULL */
int rac_bev_sampling_fwd(const void *value, const float *query_bbox, const float *box_table, const float *offsets,
                         const float *ray_logits, const float *scale_logits, const float *queue_logits,
                         const float *time_diff, float *out, float *loc_out, int ld_off, int ld_ray,
                         int ld_scale, int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W,
                         int dim, const float *pc_range, const float *depth_base, float d_region, int dtype,
                         void *stream);

/* The same kernel for the BEV streams of one decoder layer (radar, LSS) in ONE launch: same queries, boxes and time_diff,
 * per stream its own value maps, Linear outputs (same row strides) and output.  HOST arrays of nstreams (1..2) device
 * pointers.  The second stream's workgroups start as the first one's drain and its keypoint prologue runs under the first
 * stream's gathers (models/racformer_transformer.py:246-249 runs the two modules back to back). */
int rac_bev_sampling_multi_fwd(int nstreams, const void *const *values, const float *const *offsets,
                               const float *const *ray_logits, const float *const *scale_logits,
                               const float *const *queue_logits, float *const *outs, const float *query_bbox,
                               const float *box_table, const float *time_diff, int ld_off, int ld_ray, int ld_scale,
                               int ld_queue, int B, int T, int Q, int heads, int NP, int D, int H, int W, int dim,
                               const float *pc_range, const float *depth_base, float d_region, int dtype, void *stream);

/* Opt-in 16-bit BLOCK storage of a hoisted BEV value stream (round 4; the default keeps fp32): values [blocks][64] f32 -- one
 * block = the 64 channels of one head at one pixel of one frame, i.e. the unit one tap of the BEV kernel reads -- become int16
 * mantissas q [blocks][64] and one power-of-two scale per block, value = q * scale[block] (14-15 significant bits relative to
 * the block's largest value).  Not a reference interface: the reference keeps fp32 value maps (bev_self_attention.py:162-174);
 * this is the storage format of rac_bev_sampling_multi_q16_fwd below. */
int rac_quant_i16_fwd(const float *values, void *q, float *scale, int64_t blocks, void *stream);

/* rac_bev_sampling_multi_fwd over int16 block-stored value streams: values[i] int16 [B*T, H*W, heads, 64], value_scales[i] f32
 * [B*T, H*W, heads] from rac_quant_i16_fwd; everything else as above (same arithmetic in fp32; each tap's scale is folded into
 * its bilinear weight).  Halves the bytes the kernel gathers. */
int rac_bev_sampling_multi_q16_fwd(int nstreams, const void *const *values, const float *const *value_scales,
                                   const float *const *offsets, const float *const *ray_logits,
                                   const float *const *scale_logits, const float *const *queue_logits, float *const *outs,
                                   const float *query_bbox, const float *box_table, const float *time_diff, int ld_off,
                                   int ld_ray, int ld_scale, int ld_queue, int B, int T, int Q, int heads, int NP, int D,
                                   int H, int W, int dim, const float *pc_range, const float *depth_base, float d_region,
                                   void *stream);

/* Scale-adaptive self-attention core (QK^T + distance mask + softmax + AV), one kernel.
 * Replaces calc_bbox_dists, the [B*heads,Q,Q] mask and nn.MultiheadAttention's attention product
 * (models/racformer_transformer.py:296-335); in_proj / out_proj remain library GEMMs.
 *   qkv : device f32, token row (b,q) at qkv + (b*Q+q)*ld_qkv holding q|k|v, each [heads, dim]
 *   tau : device f32, rows of `heads` values, stride ld_tau (gen_tau Linear output)
 *   box_table : optional device f32 [B,Q,8] from rac_box_prep_fwd for the same boxes (NULL: centres computed here)
 *   out : device f32 [B,Q,heads*dim];  pc_range: HOST (6).  dim must be 32.  Q <= 1024: QK^T and PV run on
 *   v_mfma_f32_16x16x4_f32 (exact fp32); larger Q: an LDS-tiled fp32 VALU kernel. */
int rac_sasa_fwd(const float *qkv, const float *tau, const float *query_bbox, const float *box_table, float *out,
                 int ld_qkv, int ld_tau, int B, int Q, int heads, int dim, const float *pc_range, void *stream);

/* Layouts of the f16 hi / lo activation images that rac_add_ln_fwd and rac_rowgemm_fwd can emit beside their fp32 rows */
enum {
    RAC_SPLIT_KCAT = 0,    /* rows [hi dim | hi dim | lo dim | pad]: A operand of a K-concatenated library GEMM */
    RAC_SPLIT_LINES = 1    /* rows [dim/32 lines][hi 32 | lo 32]: X image of rac_generator_fwd (dim = 256: 1 KB per row) */
};

/* Row-wise  out = [relu]( LayerNorm( a_scale * sum_{s<num_partials} a[s] + residual + bias ) * gamma + beta ) [+ post_residual].
 * Replaces the add / bias / split-K reduce + nn.LayerNorm (+ ReLU) (+ add) launch groups of the decoder layer
 * (models/racformer_transformer.py:170-177, 199-205, 243-258).  a: device f32, row r of partial s at
 * a + s*partial_stride + r*ld_a; residual / post_residual [rows][dim], bias [dim]: optional (NULL); out row r at
 * out + r*ld_out (so results can land in a column slice of a wider buffer); dim % 4 == 0, <= 1024.
 * split_out (optional, NULL to skip): device f16 [rows][3*dim + split_pad] = [hi | hi | lo | pad] with
 * out*split_scale = hi + lo -- the K-concatenated A operand of a 3-product split GEMM (hi*Whi + hi*Wlo + lo*Whi,
 * fp32 accumulate) on the f16 matrix cores, for the Linear layers that consume this row (split_scale: a power of
 * two).  split_layout = RAC_SPLIT_LINES instead writes the line image [rows][dim/32][hi 32 | lo 32] (split_pad 0).
 * split_pad (0 or a multiple of 4) extra columns: the first two hold split_scale (the activation 1.0, which
 * meets [bias_hi | bias_lo] in the weight image, so the GEMM adds the bias itself), the others 0. */
int rac_add_ln_fwd(const float *a, int num_partials, int64_t partial_stride, int ld_a, float a_scale, const float *residual,
                   const float *bias, const float *gamma, const float *beta, const float *post_residual,
                   float *out, int ld_out, int rows, int dim, float eps, int relu, void *split_out,
                   float split_scale, int split_pad, int split_layout, void *stream);

/* The element-wise tail of the head over the stacked decoder outputs, one launch: cls <- nan_to_num(cls) in place
 * (RaCFormerTransformer.forward, models/racformer_transformer.py:58) and box <- nan_to_num(xy) with the centre scaled to
 * metres and the columns reordered to (x, y, w, l, z, h, sin, cos, vx, vy) (RaCFormer_head.forward,
 * models/racformer_head.py:124-131).  cls: device f32 [n_cls]; xy, box: device f32 [rows][10], distinct buffers;
 * pc_range: HOST float[6]. */
int rac_head_finish_fwd(float *cls, int64_t n_cls, const float *xy, float *box, int64_t rows, int code_size,
                        const float *pc_range, void *stream);

/* Position-encoder head  out = relu(LayerNorm(W x + b))  for the 3-wide box input
 * (models/racformer_transformer.py:170-173); x row r at x + r*ld_x (3 values), weight [256,3], out [rows,256]. */
int rac_pe_head_fwd(const float *x, int ld_x, const float *weight, const float *bias, const float *gamma,
                    const float *beta, float *out, int rows, int dim, float eps, void *stream);

/* Box refinement tail of a decoder layer: refine_bbox + velocity / time_diff + theta_d2xy of the emitted
 * boxes (models/racformer_transformer.py:230-236, :265-269, :134; models/bbox/utils.py:82-90).
 *   proposal [B*Q,10] (this layer's input boxes), delta [B*Q,10] (reg_branch output),
 *   time_diff_safe [B,T] (time_diff with values < 1e-5 replaced by 1)  ->
 *   bbox_pred [B*Q,10] (polar, next layer's input), bbox_xy [B*Q,10] (normalised xy, the layer's output). */
int rac_refine_fwd(const float *proposal, const float *delta, const float *time_diff_safe, float *bbox_pred,
                   float *bbox_xy, int B, int Q, int T, float num_ray, void *stream);

/* Boundary between two decoder layers in one launch: rac_refine_fwd for the finished layer and, for the boxes it produces,
 * the first two launches of the next layer -- rac_box_prep_fwd (box_table [B*Q,8]) and rac_pe_head_fwd (pe_out [B*Q,256]
 * = relu(LayerNorm(pe_weight (theta,d,z) + pe_bias))).  Same arithmetic as the three separate entry points. */
int rac_layer_boundary_fwd(const float *proposal, const float *delta, const float *time_diff_safe, float *bbox_pred,
                           float *bbox_xy, float *box_table, const float *pc_range, const float *pe_weight,
                           const float *pe_bias, const float *pe_gamma, const float *pe_beta, float *pe_out, int B, int Q,
                           int T, int dim, float num_ray, float eps, void *stream);

/* Matrix-core arithmetic of rac_mixing_fwd. */
enum {
    RAC_MIX_F32 = 0,   /* v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate (bit-for-bit an fmaf chain) */
    RAC_MIX_F16X3 = 1  /* 16-bit matrix cores on split operands, f32 accumulate, fp32-GEMM accuracy: x @ M as three
                          bf16 terms each (6 products, truncation 2^-23, any fp32 magnitude), S @ Y as two f16 terms
                          each (3 products, truncation 2^-22; needs |S * param_scale| < 6e4) */
};

/* AdaptiveMixing core on the matrix cores: per (query, group) item
 *   Y = relu(LN_{[P,64]}(x @ M)),  Z = relu(LN_{[128,64]}(S @ Y))
 * Replaces the two batched matmuls, two layer norms and two ReLUs of AdaptiveMixing.inner_forward
 * (models/racformer_transformer.py:589-603); out_proj consumes its output image through rac_outproj_fwd.
 *   x      : device f32 [num_query, groups, in_points, 64]    (sampled features, B folded into num_query)
 *   params : device f32, row q at params + q*ld_params, per group [64*64 (M, in x out) | 128*in_points (S)];
 *            every value is multiplied by param_scale on load (1.0, or the power-of-two alpha of a split GEMM)
 *   out    : device f32 [num_query, groups, 128, 64] (NULL to skip when out_split is given)
 *   out_split : optional device f16 line image [num_query][groups*256][hi 32 | lo 32] of the flattened output row
 *            (K order group, out point, channel) with out*split_scale = hi + lo -- the A operand of rac_outproj_fwd
 *            (NULL to skip) */
int rac_mixing_fwd(const float *x, const float *params, float param_scale, float *out, void *out_split,
                   float split_scale, int ld_params, int num_query, int groups, int in_points, int channels, int out_points,
                   float eps, int mfma_mode, void *stream);

/* rac_sampling4d_fwd and rac_mixing_fwd (RAC_MIX_F16X3) as ONE kernel: the mixing workgroup of a (query, group) item gathers the
 * item's T * NP * D sampling points itself -- keypoints, projection, first-valid-view selection, level softmax and bilinear taps
 * computed by the code of the stand-alone sampling kernel, so the sampled features are bit for bit what rac_sampling4d_fwd writes
 * -- while the item's generated parameters are on their way from HBM; the [B,Q,G,T*P,64] tensor between
 * RaCFormerSampling (models/racformer_transformer.py:361-408, sparsebev_sampling.py:45-131) and AdaptiveMixing (:589-603) never
 * exists.  Arguments: those of rac_sampling4d_fwd without `out` / `compact`, then those of rac_mixing_fwd without `x` / `mfma_mode`
 * (num_query = B * Q, in_points = T * NP * D <= 96).  Built for 4 fp32 levels of 64 channels per group; every level's T * G slots
 * of one sample must stay below 2 GiB.  Other shapes: the two stand-alone entry points. */
int rac_mixing_sampled_fwd(const void *const *feats, const int32_t *hw, int L, const float *query_bbox, const float *box_table,
                           const float *offsets, const float *ray_logits, const float *scale_logits, const float *time_diff,
                           const float *lidar2img, float *loc_out, float *w_out, const unsigned char *view_in, int ld_off,
                           int ld_ray, int ld_scale, int B, int T, int N, int G, int Q, int NP, int D, int C,
                           const float *pc_range, const float *depth_base, float d_region, float image_h, float image_w,
                           float eps_proj, int dtype, const float *params, float param_scale, float *out, void *out_split,
                           float split_scale, int ld_params, int out_points, float eps_ln, void *stream);

/* Split-precision GEMM operand image: nn.Linear weight [N][K] f32 -> f16 [N][K/32][hi 32 | lo 32] of weight * scale
 * (per 32 values of K one 128-byte line; hi + lo carry 22 significant bits).  Packed once per set of weights. */
int rac_gemm_split_pack_fwd(const float *weight, void *image, int N, int K, float scale, void *stream);

/* AdaptiveMixing.out_proj (nn.Linear(groups*128*64 -> 256), models/racformer_transformer.py:566,606) as a hand-written
 * split-K GEMM on the f16 matrix cores at fp32-GEMM accuracy (three products hi*hi + hi*lo + lo*hi, fp32 accumulate):
 *   partials[s][m][n] = sum_{k in slice s} Z[m][k] * W[n][k]        (unscaled: the consumer applies both powers of two)
 *   z_image : device f16 line image [M][K/32][hi 32 | lo 32]   (rac_mixing_fwd's out_split)
 *   w_image : device f16 line image [N][K/32][hi 32 | lo 32]   (rac_gemm_split_pack_fwd)
 *   partials: device f32 [slices][M][N];  K % (32 * slices) == 0 (N % 4 == 0 gives aligned 16-byte stores; other N work through
 *             unaligned stores, slower).  rac_add_ln_fwd sums the slices. */
int rac_outproj_fwd(const void *z_image, const void *w_image, float *partials, int M, int N, int K, int slices, void *stream);

/* AdaptiveMixing.parameter_generator (nn.Linear(256 -> groups*(64*64 + 128*in_points)), models/racformer_transformer.py:565,589)
 * on the same hand-written kernel:  out[m][n] = alpha * sum_k X[m][k] * W[n][k] + bias[n]   (alpha undoes the two powers of two
 * of the images).  One workgroup per 256 features walks all rows: the weights cross the fabric once.  Also the eleven Linears
 * of the three sampling modules (256 -> 2189, models/racformer_transformer.py:361-366, 490-500): for narrow outputs the rows
 * are cut into chunks so that feature blocks x chunks covers the CUs.
 *   x_image : device f16 line image [M][K/32][hi 32 | lo 32]   (rac_rowgemm_fwd's split_out, split_layout = RAC_SPLIT_LINES)
 *   w_image : device f16 line image [N][K/32][hi 32 | lo 32]   (rac_gemm_split_pack_fwd);  bias device f32 [N] or NULL
 *   out     : device f32, row m at out + m*ld_out;  K % 32 == 0, ld_out % 4 == 0, N % 4 == 0 unless K == 256 */
int rac_generator_fwd(const void *x_image, const void *w_image, const float *bias, float alpha, float *out, int64_t ld_out, int M,
                      int N, int K, void *stream);

/* BEVSelfAttention.value_proj over a whole BEV stream (nn.Linear(256 -> 256) on every pixel of every frame,
 * models/bev_self_attention.py:162-174) read straight from the channel-first maps:
 *     out[f*HW + p][n] = sum_c x[f][c][p] * W[n][c] + add[p][n]        (add NULL: + bias[n], bias NULL: nothing)
 *   x       : device f32 [frames][256][HW] (the reference's [B*T, C, H, W]);  HW % 32 == 0
 *   w_image : device f16 line image [256][8][hi 32 | lo 32] (rac_gemm_split_pack_fwd with scale 2^s); w_alpha = 2^-s
 *   add     : device f32 [HW][256] -- the frame-independent term value_proj(pos) + bias -- or NULL;  bias: device f32 [256] or NULL
 *   out     : device f32 [frames*HW][256] (= [B*T, H*W, heads, 64], the value operand of rac_bev_sampling_fwd)
 * Split-precision f16 MFMA (hi / lo per operand, fp32 accumulate), activation scale per pixel.  16-byte aligned pointers. */
int rac_value_proj_fwd(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, float *out,
                       int frames, int channels, int HW, int features, void *stream);
/* rac_value_proj_fwd writing the int16 block storage of rac_quant_i16_fwd (q [frames*HW][256] int16, scale [frames*HW][4] f32):
 * bit for bit rac_quant_i16_fwd(rac_value_proj_fwd(...)); the LSS BEV value stream of rac_bev_sampling_multi_q16_fwd. */
int rac_value_proj_q16_fwd(const float *x, const void *w_image, float w_alpha, const float *add, const float *bias, void *q,
                           float *scale, int frames, int channels, int HW, int features, void *stream);

/* The temporal-fusion convolution of RadarBEVTemporalEncoder (3x3, stride 1, pad 1, Cin -> 256; the 193-GFLOP
 * nn.Conv2d of models/racformer_transformer.py:631,655) as an implicit GEMM on the f16 matrix cores with
 * hi/lo-split operands (3 products, fp32 accumulate: fp32-convolution accuracy).  Three calls:
 *   rac_absmax_fwd    amax_out[0] = max(floor_value, max |v| over `num` device arrays) (srcs / counts: HOST arrays;
 *                     16-byte aligned sources); a one-thread launch sets amax_out to floor_value first.  Fixes the activations'
 *                     power-of-two scale; floor_value >= 0 covers sources whose bound is known without reading them.
 *   rac_conv_pack_fwd src [N,C,H,W] f32 -> channel range [c_offset, c_offset+C) of the kernel's activation image
 *                     xs = f16 [N][H+2][W+2][c_total/32][2][32] (per pixel and 32-channel chunk: hi, then lo, of
 *                     v * 2^e; e from *amax).  Only interior pixels are written: the caller zeroes xs once (border =
 *                     the convolution's zero padding).  Any H, W (16-byte loads when W % 4 == 0); channel counts multiples of 32.
 *   rac_conv3x3_fwd   out [N,H,W,256] f32 (channel-last) = conv3x3(xs) * w_alpha / 2^e + bias[c] (or + pixel_bias[h*W+w][c]
 *                     if pixel_bias != NULL: a per-pixel additive map shared by the N images), with
 *                     ws = f16 [9 taps (ky*3+kx)][Cin/32][256][2][32] holding hi / lo of weight[co][ci][ky][kx] / w_alpha
 *                     (w_alpha a power of two chosen by the packer).  Any H*W (tiles of 256 pixels, the last one of an image
 *                     ragged); a pixel_bias map needs H*W to be a multiple of 256. */
int rac_absmax_fwd(const float *const *srcs, const int64_t *counts, int num, float floor_value, float *amax_out,
                   void *stream);
int rac_conv_pack_fwd(const float *src, const float *amax, void *xs, int N, int C, int H, int W, int c_total,
                      int c_offset, void *stream);
/* rac_conv_pack_fwd with a per-channel bias [C] (or NULL) added before the split, for an image whose N frames come in groups
 * of frames_per_group of which only the first live_per_group exist in src ([N / frames_per_group * live_per_group, C, H, W]):
 * the other frames are the bias alone.  (The hidden half of the temporal-fusion input: the ConvGRU leaves the frames t >= 4 at
 * zero, so after the resize and the last convolution they are exactly that convolution's bias, racformer_transformer.py:674-693.) */
int rac_conv_pack_bias_fwd(const float *src, const float *bias, const float *amax, void *xs, int N, int C, int H, int W,
                           int c_total, int c_offset, int frames_per_group, int live_per_group, void *stream);
int rac_conv3x3_fwd(const void *xs, const void *ws, const float *bias, const float *pixel_bias, const float *amax,
                    float w_alpha, float *out, int N, int H, int W, int Cin, int Cout, void *stream);
/* rac_conv3x3_fwd whose epilogue writes the int16 block storage of rac_quant_i16_fwd instead of fp32 (the radar BEV value stream,
 * value_proj composed into the weights: the convolution's result IS the stream rac_bev_sampling_multi_q16_fwd reads):
 *   q [N*H*W][256] int16, scale [N*H*W][4] f32 -- bit for bit rac_quant_i16_fwd(rac_conv3x3_fwd(...)), without the fp32 stream. */
int rac_conv3x3_q16_fwd(const void *xs, const void *ws, const float *bias, const float *pixel_bias, const float *amax,
                        float w_alpha, void *q, float *scale, int N, int H, int W, int Cin, int Cout, void *stream);

/* Producer-side pyramid layout: the last stage of the image neck -- the per-level 3x3 / pad 1 /
 * Cin -> 256 output convolution of the FPN (mmdet 2.28.2 FPN.fpn_convs[i], the same structure as the in-tree CustomFPN,
 * models/necks/fpn.py:109-132,180) -- writing the layout the decoder samples from directly, so that the reshape / permute copy
 * of models/racformer_transformer.py:112-124 (1.47 GB of traffic per sample at f8) never runs:
 *   xs / ws / amax / w_alpha : as for rac_conv3x3_fwd (activation image of the laterals [num_images, Cin, H, W], image index
 *                              (b*T + t) * num_cams + cam; packed weights [9][Cin/32][256][2][32])
 *   out : device f32 [num_images / num_cams * 4][num_cams][H][W][64]: slot (b*T + t) * 4 + g holds output channels g*64..g*64+63
 *         of the num_cams images of (b, t), channel-last -- the `feats` operand of rac_sampling4d_fwd / rac_msmv_fwd.
 * bias: device f32 [256] or NULL.  num_images % num_cams == 0; any H, W. */
int rac_fpn_conv_fwd(const void *xs, const void *ws, const float *bias, const float *amax, float w_alpha, float *out,
                     int num_images, int H, int W, int Cin, int num_cams, void *stream);

/* Element-wise pieces of RadarBEVTemporalEncoder (models/racformer_transformer.py:618-720).
 *   rac_gru_gate_fwd   ConvGRUCell update after the gates convolution (:705-720): gates [B,3C,H,W] (z | r | cand),
 *                      h_prev [B,C,H,W] (batch stride h_prev_bstride floats) -> h_out (batch stride h_out_bstride):
 *                      h = (1 - sigmoid(z)) * h_prev + sigmoid(z) * tanh(cand + sigmoid(r) * h_prev);
 *                      bias_map (optional [3C,H,W]) is added to the gates first; h_out2 (optional) receives h as well
 *   rac_upsample2x_fwd nn.Upsample(scale_factor=2, bilinear, align_corners=True) (:633-636) on `planes` = N*C maps
 *                      [h,w] -> [2h,2w] */
int rac_gru_gate_fwd(const float *gates, const float *h_prev, int64_t h_prev_bstride, float *h_out, int64_t h_out_bstride,
                     const float *bias_map, float *h_out2, int64_t h_out2_bstride, int B, int C, int HW, void *stream);
int rac_upsample2x_fwd(const float *src, float *dst, int64_t planes, int h, int w, void *stream);

/* The small dense layers of a decoder layer with their surrounding row-wise work in one launch
 * (models/racformer_transformer.py:170-177, 243-269: nn.Linear + the add / split-K sum / nn.LayerNorm / ReLU before it):
 *   X   = per 256-wide segment s:  [relu]( LN( a_scale * sum_p a[p] + bias0 + residual ) * gamma + beta ) [+ post]
 *   out = [relu on columns >= relu_from]( X @ w^T + b ),   w [N][256*num_seg] (torch Linear layout), exact-fp32 MFMA.
 * Segment sources are rows of 256 floats: row r of partial p at a + p*partial_stride + r*ld_a; residual / post / x_out
 * rows at their own strides; gamma == NULL skips the LayerNorm (relu / post still apply).  x_out (optional) receives
 * the finished segment; split_out (optional) its f16 [hi | hi | lo | pad] image (layout of rac_add_ln_fwd's split_out).
 * Up to RAC_ROWGEMM_MAX_BATCH independent GEMMs over the same `rows` share the launch (descs: HOST array). */
#define RAC_ROWGEMM_MAX_BATCH 3
typedef struct {
    const float *a;
    int64_t partial_stride;
    const float *bias0, *residual, *gamma, *beta, *post;
    float *x_out;
    void *split_out;
    int ld_a, num_partials, ld_res, ld_post, ld_xout, relu, split_pad, split_layout;
    float a_scale, eps, split_scale;
} rac_rowseg;
typedef struct {
    rac_rowseg seg[3];
    const float *w, *b;
    float *out;
    int num_seg, N, ld_out, relu_from;
} rac_rowgemm;
int rac_rowgemm_fwd(const rac_rowgemm *descs, int num, int rows, void *stream);

/* The downsample convolution of RadarBEVTemporalEncoder (3x3, stride 2, pad 1, Cin -> 64; models/racformer_transformer.py:632,646)
 * on the activation image of rac_conv_pack_fwd (its first Cin channels; the image holds Cin_image >= Cin channels) with the
 * arithmetic of rac_conv3x3_fwd.  ws = f16 [9 taps][Cin/32][64][2][32];  out: channels 0..63 of an NCHW f32 buffer
 * [N, out_channels_total, H/2, W/2] (64 for a plain output);  (H/2)*(W/2) % 128 == 0. */
int rac_conv3x3s2_fwd(const void *xs, const void *ws, const float *bias, const float *amax, float w_alpha, float *out,
                      int out_channels_total, int N, int H, int W, int Cin, int Cin_image, int Cout, void *stream);

/* NMS-free decode of one sample in one launch: sigmoid, top-max_num of the num_query x num_classes scores (sorted by
 * score, ties by flat index), label = idx % C, query = idx / C, denormalize_bbox (exp of the log sizes, atan2 of sin / cos),
 * centre-range and score masks, z moved to the box bottom.  Replaces NMSFreeCoder.decode_single + the reshuffle of
 * get_bboxes (models/bbox/coders/nms_free_coder.py:37-88, models/bbox/utils.py:26-46, models/racformer_head.py:488-507).
 *   cls_scores [Q,C] logits, bbox_preds [Q,10] = (cx, cy, log w, log l, cz, log h, sin, cos, vx, vy) of the last layer
 *   out [max_num,11] = (x, y, z_bottom, w, l, h, yaw, vx, vy, score, label); masked rows carry score = -1
 *   post_center_range: HOST (6);  Q*C <= 16384, max_num <= 512 */
int rac_decode_fwd(const float *cls_scores, const float *bbox_preds, float *out, int num_query, int num_classes,
                   int max_num, const float *post_center_range, float score_threshold, int use_threshold, void *stream);

/* Backward of the two gather operators (SURVEY.md section 8 "next" row f4; fp32 features only).
 * rac_msmv_bwd  <- _ms_deform_attn_cuda_{c45,c2345,c23456}_backward, models/csrc/msmv_sampling/msmv_sampling.cpp:302-497
 *                  (kernels msmv_sampling_backward.cu:108-440): grad_out [S,Q,C,P]; grad_feats[l] like feats[l] and
 *                  ZERO-FILLED by the caller; grad_loc [S,Q,P,3] (view component = 0), grad_w [S,Q,P,L] overwritten.
 * rac_msda_bwd  <- mmcv `_ext.ms_deform_attn_backward`, call site models/multi_scale_deformable_attn_function.py:148-158:
 *                  grad_out [bs,Q,heads*dim]; grad_value like value, ZERO-FILLED by the caller; grad_loc / grad_attn
 *                  like loc / attn, overwritten.
 * grad_loc / grad_w / grad_attn have one writer per element (deterministic); the feature / value scatter uses float
 * atomics, so those two gradients can differ in the last bits from run to run (as in the reference). */
int rac_msmv_bwd(const float *grad_out, const void *const *feats, const int32_t *hw, int L, const float *loc,
                 const float *w, void *const *grad_feats, float *grad_loc, float *grad_w, int S, int N, int Q,
                 int P, int C, void *stream);
int rac_msda_bwd(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *starts,
                 const float *loc, const float *attn, float *grad_value, float *grad_loc, float *grad_attn,
                 int bs, int keys, int heads, int dim, int Q, int L, int P, void *stream);

/* BEVPoolv2 (Lift-Splat-Shoot voxel pooling) -- SURVEY.md section 8 "next" row f2.  Replaces
 * bev_pool_v2_forward / bev_pool_v2_backward of models/csrc/bev_pool_v2/src/bev_pool.cpp:40-111
 * (kernels bev_pool_cuda.cu:21-136); argument order follows those entry points.
 *   depth [b,n,d,h,w] f32, feat [b,n,h,w,c] f32, out [b,z,y,x,c] f32 (pre-zeroed by the caller, as in
 *   bev_pool.py:29), ranks_* int32 [n_points], interval_* int32 [n_intervals]; all device pointers.
 * Backward expects the intervals regrouped by ranks_feat (bev_pool.py:50-63) and depth_grad / feat_grad
 * pre-zeroed.  Deterministic: one writer per output element, no atomics. */
int rac_bev_pool_v2_fwd(const float *depth, const float *feat, float *out, const int32_t *ranks_depth,
                        const int32_t *ranks_feat, const int32_t *ranks_bev, const int32_t *interval_lengths,
                        const int32_t *interval_starts, int c, int n_intervals, void *stream);
int rac_bev_pool_v2_bwd(const float *out_grad, float *depth_grad, float *feat_grad, const float *depth,
                        const float *feat, const int32_t *ranks_depth, const int32_t *ranks_feat,
                        const int32_t *ranks_bev, const int32_t *interval_lengths, const int32_t *interval_starts,
                        int c, int n_intervals, void *stream);

/* ---- Round 5: the ConvGRU branch of RadarBEVTemporalEncoder without library convolutions -------------------------------------
 * (models/racformer_transformer.py:645-656 inner_forward, :674-693 ConvGRU, :705-720 ConvGRUCell)
 *
 * rac_conv_direct_fwd: a 3x3 convolution (stride 1 or 2, pad 1) for the SMALL maps of that branch (64 x 64 x 64 channels, one
 * frame at a time through the recurrence), with the split-precision arithmetic of rac_conv3x3_fwd (three f16 MFMA products of
 * hi / lo operands, fp32 accumulate: fp32-convolution accuracy).  Unlike rac_conv3x3_fwd nothing is staged through LDS and no
 * workgroup barrier is executed: a wave owns 16 output pixels x (16 * tiles) output channels and loads its MFMA fragments
 * straight from the activation image / weight image (both L2-resident at these sizes) through a register ring several K steps
 * ahead -- these launches are latency-bound chains, not throughput kernels.
 *
 *   in_img   f16 [frames][H+2][W+2][in_chunks_total][hi 32 | lo 32]  zero border; K runs over chunks in_chunk0 .. +chunks-1
 *            (chunks == 0: no convolution, the accumulators stay zero -- the recurrence's first step, h_0 = 0)
 *   ws       f16 [9 taps][chunks][Cout][hi 32 | lo 32]  (racformer_amd.fused.pack_conv3x3_weight), w_alpha = its 2^-s
 *   scales   an image's power-of-two scale is rac's act_scale(bound) with bound = mul * (*amax) + add  (amax may be NULL):
 *            a DEVICE word plus host constants, so that a bound derived from the weights follows the measured input maximum
 *   frame maps  frame(n) = (n / live) * stride + n % live + first:  the n-th processed frame inside a [groups][stride] stack
 * mode RAC_CD_IMAGE : out_img <- (conv + bias) as an activation image [frames][OH+2][OW+2][out_chunks_total][hi|lo], channels
 *                     out_chunk0 * 32 .. (interior pixels only; the caller zeroed the border once)
 * mode RAC_CD_F32   : out_f32 [N][OH*OW][Cout] channel-last <- conv + bias + pixel_map[OH*OW][Cout] (either may be NULL)
 * mode RAC_CD_GRU   : Cout = 3 * 64 gate channels (z | r | candidate); pre = conv(h_prev image) + xpart[frame][pixel][192];
 *                     z = sigmoid, r = sigmoid, cand = tanh(pre_c + r * h_prev), h = (1 - z) h_prev + z cand  (:714-720);
 *                     h -> h_out f32 [frames][OH*OW][64] AND out_img (the next step's convolution input); h_prev NULL = zeros
 */
enum { RAC_CD_IMAGE = 0, RAC_CD_F32 = 1, RAC_CD_GRU = 2 };
typedef struct {
    const float *amax;     /* device word or NULL */
    float mul, add;
} rac_cd_scale;
typedef struct {
    int live, stride, first;
} rac_cd_frames;
typedef struct {
    int mode, conv_stride;           /* RAC_CD_*, 1 | 2 */
    int N, H, W;                     /* frames processed; INPUT map size (output = H / conv_stride x W / conv_stride) */
    const void *in_img;
    int in_chunks_total, in_chunk0, chunks;
    rac_cd_frames in_frames;
    rac_cd_scale in_scale;
    const void *ws;
    float w_alpha;
    int Cout;
    const float *bias;               /* [Cout] or NULL (IMAGE, F32) */
    void *out_img;                   /* IMAGE, GRU */
    int out_chunks_total, out_chunk0;
    rac_cd_frames out_frames;
    rac_cd_scale out_scale;
    float *out_f32;                  /* F32 */
    const float *pixel_map;          /* F32: [OH*OW][Cout] or NULL */
    const float *xpart;              /* GRU: [frames][OH*OW][192] */
    rac_cd_frames xpart_frames;
    const float *h_prev;             /* GRU: [frames][OH*OW][64] or NULL */
    rac_cd_frames h_prev_frames;
    float *h_out;                    /* GRU */
    rac_cd_frames h_out_frames;
} rac_conv_direct;
int rac_conv_direct_fwd(const rac_conv_direct *desc, void *stream);

/* nn.Upsample(scale_factor=2, bilinear, align_corners=True) (models/racformer_transformer.py:633-636) of channel-last maps
 * src f32 [frames][h*w][C] straight into an activation image f16 [frames][2h+2][2w+2][C/32][hi 32 | lo 32] (interior pixels)
 * with the scale act_scale(bound); C % 32 == 0. */
int rac_upsample2x_image_fwd(const float *src, void *img, int frames, int h, int w, int C, float bound, void *stream);

/* rac_conv3x3_fwd / rac_conv3x3_q16_fwd for a stack of [groups][frames_per_group] images of which only the first live_per_group
 * of a group carry all Cin channels: the others' last Cin - Cin_dead channels are a per-channel constant (the ConvGRU leaves
 * frames >= 4 at zero, so their hidden half is the bias of the convolution behind the resize, :674-693), whose contribution
 * the caller has folded into THEIR per-pixel map (border-aware: composed through the zero padding) -- those images run
 * Cin_dead / 32 chunks per tap and add pixel_bias_dead instead of pixel_bias_live.  Exactly one of out / (q, scale). */
int rac_conv3x3_temporal_fwd(const void *xs, const void *ws, const float *pixel_bias_live, const float *pixel_bias_dead,
                             const float *amax, float w_alpha, float *out, void *q, float *scale, int N, int H, int W, int Cin,
                             int Cin_dead, int frames_per_group, int live_per_group, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RACFORMER_HIP_H */
