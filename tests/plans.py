"""Alternate execution plans of a decoder layer that only the tests take (TEST INFRASTRUCTURE, moved out of the product module in
round 5).  They import the product; the product does not import them: ``racformer_amd.transformer`` keeps a registry of hooks
(``register_alternate_plan``) that this module fills when it is imported.

  "reference_ops"  (``decoder_layer.fused = False``)   the reference's own op decomposition: torch keypoint chains
                   (models/racformer_transformer.py:361-408, 490-529), ``sampling_4d`` on the msmv operator, the MSDA operator + torch
                   frame fusion, torch layers -- the plan the CPU host-logic test runs with the operators replaced by the oracle's, and
                   the operator-boundary cross-check of the fused kernels;
  "library_chain"  (``decoder_layer.rowgemm = False``) the fused gather kernels with library GEMMs + rac_add_ln_fwd launches between
                   them instead of rac_rowgemm_fwd.
"""
import torch
import torch.nn.functional as F

from racformer_amd import transformer as T
from racformer_amd.bbox_utils import theta_d2xy_coods
from racformer_amd.fused import add_ln, bev_sampling_fused, box_prep, pe_head, refine_fused, sasa_fused
from racformer_amd.transformer import _warp_to_polar, make_sample_points


# ------------------------------------------------------------------------------------------------ keypoint chains (torch)
def image_keypoints(self, query_ray, query_feat, time_diff, d_region):
    """-> metric points [B,Q,T,G,P,3], scale weights [B,Q,G,T,P,L] (softmax over L)."""
    B, Q, _ = query_ray.shape
    T, G, NP, D, L = self.num_frames, self.num_groups, self.num_points, self.depth_num, self.num_levels
    pc = self.pc_range
    qb = theta_d2xy_coods(query_ray)
    off = self.sampling_offset(query_feat).view(B, Q, G * NP * D, 3)
    pts = make_sample_points(qb, off, pc).view(B, Q, 1, G, NP * D, 3)
    theta, dist = _warp_to_polar(pts[..., 0:2], query_ray[..., 8:], time_diff, pc)
    base = torch.linspace(-d_region, d_region, D, device=query_feat.device, dtype=query_feat.dtype)
    d_off = base + (torch.sigmoid(self.ray_points_offset(query_feat)) * 2 - 1) * d_region / D / 2  # [B,Q,D]
    dist = (dist.view(B, Q, T, G, NP, D) + d_off[:, :, None, None, None, :]).reshape(B, Q, T, G, NP * D, 1)
    xy = theta_d2xy_coods(torch.cat([theta, dist], dim=-1))
    px = xy[..., 0:1] * (pc[3] - pc[0]) + pc[0]
    py = xy[..., 1:2] * (pc[4] - pc[1]) + pc[1]
    pz = pts[..., 2:3].expand(B, Q, T, G, NP * D, 1)
    points = torch.cat([px, py, pz], dim=-1)
    sw = self.scale_weights(query_feat).view(B, Q, G, T, D * NP, L)
    return points, torch.softmax(sw, dim=-1)

def sampling_reference_ops(self, query_ray, query_feat, mlvl_feats, img_metas, d_region=0.1):
    """torch keypoint chain + sampling_4d on the msmv operator (the reference's decomposition)."""
    image_h, image_w, _ = img_metas[0]["img_shape"][0]
    points, sw = image_keypoints(self, query_ray, query_feat, img_metas[0]["time_diff"], d_region)
    return T.sampling_4d(points, mlvl_feats, sw, img_metas[0]["lidar2img"], image_h, image_w, loc_tap=self.capture_loc,
                       view_in=self._next_forced())



def bev_keypoints(self, query_ray, query_feat, time_diff, d_region):
    """-> loc [B,Q,heads,T,P,2] in [0,1], weights [B,Q,heads,T,1,P] (:490-529)."""
    B, Q, _ = query_ray.shape
    T, Hn, NP, D = self.num_frames, self.num_heads, self.num_points, self.depth_num
    pc = self.pc_range
    qb = theta_d2xy_coods(query_ray)
    off = self.sampling_offset(query_feat).view(B, Q, Hn * NP * D, 2)
    off = torch.cat([off, torch.zeros_like(off[..., 0:1])], dim=-1)
    pts = make_sample_points(qb, off, pc).view(B, Q, 1, Hn, NP * D, 3)
    theta, dist = _warp_to_polar(pts[..., 0:2], query_ray[..., 8:], time_diff, pc)
    base = torch.linspace(-d_region, d_region, D, device=query_feat.device, dtype=query_feat.dtype)
    d_off = base + (torch.sigmoid(self.ray_points_offset(query_feat)) * 2 - 1) * d_region / D / 2
    dist = (dist.view(B, Q, T, Hn, NP, D) + d_off[:, :, None, None, None, :]).reshape(B, Q, T, Hn, NP * D, 1)
    loc = theta_d2xy_coods(torch.cat([theta, dist], dim=-1)).permute(0, 1, 3, 2, 4, 5).contiguous()
    sw = self.scale_weights(query_feat).view(B, Q, Hn, 1, self.num_levels, D * NP)
    sw = torch.softmax(sw, dim=-1).expand(B, Q, Hn, T, self.num_levels, D * NP).contiguous()
    return loc, sw

def bev_attend_reference_ops(self, query_ray, query_feat, value, hw, time_diff, d_region):
    """torch keypoint chain + MSDA operator + torch frame fusion (the reference's decomposition)."""
    loc, sw = bev_keypoints(self, query_ray, query_feat, time_diff, d_region)
    return self.attention.attend(query_feat, value, loc, sw, hw)



# ------------------------------------------------------------------------------------------------ the reference's op decomposition
def layer_reference_ops(self, query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, layer, prepared, stages=None):
    """RaCFormerTransformerDecoderLayer.forward as the reference decomposes it (racformer_transformer.py:239-279): torch modules,
    the two gather operators, torch refine.  ``self`` is the product's decoder layer (weights, hoisted ``prepared`` tensors)."""
    meta = img_metas[0]
    time_diff, d_region = meta["time_diff"], self.d_region_list[layer]
    query_pos = self.position_encoder(query_bbox[..., :3])
    query_feat = query_feat + query_pos
    sa = self.self_attn.forward_unfused(query_bbox, query_feat, attn_mask)
    query_feat = self.norm1(sa)
    radar_raw = bev_attend_reference_ops(self.sampling_radar_bev, query_bbox, query_feat, prepared["radar_value"], prepared["radar_hw"],
                                         time_diff, d_region)
    lss_raw = bev_attend_reference_ops(self.sampling_lss_bev, query_bbox, query_feat, prepared["lss_value"], prepared["lss_hw"], time_diff,
                                       d_region)
    sampled_feat = sampling_reference_ops(self.sampling, query_bbox, query_feat, mlvl_feats, img_metas, d_region=d_region)
    query_radar_feat = self.norm_radar_bev(radar_raw)
    query_lss_feat = self.norm_lss_bev(lss_raw)
    mixed = self.mixing(sampled_feat, query_feat, None)
    query_feat = self.norm2(mixed)
    query_feat = self.norm_fusion(self.fusion(torch.cat((query_feat, query_radar_feat, query_lss_feat), dim=-1)))
    ffn_out = self.ffn(query_feat)
    query_feat = self.norm3(ffn_out)
    cls_score = self.cls_branch(query_feat)
    bbox_pred = self.refine_bbox(query_bbox, self.reg_branch(query_feat))
    if time_diff.shape[1] > 1:
        td = meta["time_diff_safe"][:, 1:2, None]
        bbox_pred = torch.cat([bbox_pred[..., :8], bbox_pred[..., 8:] / td], dim=-1)
    bbox_xy = theta_d2xy_coods(bbox_pred)
    if stages is not None:
        stages.update(position_encoder=query_pos, self_attn=sa, sampling_radar_bev=radar_raw, sampling_lss_bev=lss_raw,
                      sampling=sampled_feat, mixing=mixed, ffn=ffn_out)
    self.last_bbox_xy = bbox_xy   # theta_d2xy_coods(bbox_pred), the per-layer output of the decoder (:134)
    return query_feat, cls_score, bbox_pred


# ------------------------------------------------------------------------------------------------ library GEMMs between the fused kernels
def layer_library_chain(self, query_bbox, query_feat, mlvl_feats, img_metas, layer, prepared, stages=None):
    """The layer as a chain of library GEMMs and hand-written HIP kernels only: every LayerNorm is fused
    with the add / split-K reduction / bias / ReLU around it (rac_add_ln_fwd), the box tail is one
    kernel (rac_refine_fwd).  Same arithmetic as ``forward`` (racformer_transformer.py:239-279)."""
    meta = img_metas[0]
    time_diff, d_region = meta["time_diff"], self.d_region_list[layer]
    qb = query_bbox.contiguous()
    pe = self.position_encoder
    h = pe_head(qb[..., :3], pe[0], pe[1])                         # relu(LN(Linear(3->256)))
    x = add_ln(pe[3](h), pe[4], relu=True, post=query_feat)       # query_feat + relu(LN(Linear(h)))
    # scale-adaptive self-attention
    p = self.self_attn.attention.attn
    E = self.embed_dims
    table = box_prep(qb, self.pc_range)      # decode_bbox(theta_d2xy(.)) once for SASA and the 3 sampling kernels
    lin = F.linear(x, prepared["sasa_w"][0], prepared["sasa_w"][1])
    attn = p.out_proj(sasa_fused(lin[..., :3 * E], lin[..., 3 * E:], qb, self.self_attn.num_heads, self.pc_range,
                                 box_table=table))
    packs = prepared.get("split_packs")
    x1, x1_split = add_ln(attn, self.norm1, residual=x, split=True) if packs else (add_ln(attn, self.norm1, residual=x), None)
    # the three sampling modules: one wide GEMM, one box table, three fused kernels
    lin = F.linear(x1, prepared["wide_w"], prepared["wide_b"]).split(prepared["wide_widths"], dim=-1)
    rb, lb = self.sampling_radar_bev, self.sampling_lss_bev
    r_off, r_ray, r_sc, r_qu = lin[3:7]
    l_off, l_ray, l_sc, l_qu = lin[7:11]
    B, Q = x1.shape[:2]
    bev = torch.empty(2, B, Q, E, device=x1.device, dtype=torch.float32)
    bev_sampling_fused(prepared["radar_value"], prepared["radar_hw"], qb, r_off, r_ray, r_sc, r_qu, time_diff,
                       rb.num_frames, rb.num_heads, rb.num_points, rb.depth_num, rb.pc_range, d_region,
                       box_table=table, out=bev[0])
    bev_sampling_fused(prepared["lss_value"], prepared["lss_hw"], qb, l_off, l_ray, l_sc, l_qu, time_diff,
                       lb.num_frames, lb.num_heads, lb.num_points, lb.depth_num, lb.pc_range, d_region,
                       box_table=table, out=bev[1])
    sampled_feat = self._sample(qb, x1, mlvl_feats, img_metas, d_region, lin[0:3], table)
    # adaptive mixing: generator GEMM -> MFMA kernel -> split-K partial products of out_proj
    partials = self.mixing.out_proj_partials(sampled_feat, x1, prepared["out_proj_split"], None, packs, x1_split)
    p_scale = packs["out_alpha"] if packs else 1.0
    # both BEV output projections as one batched GEMM; the three normalised branches land directly in the
    # [x2 | radar | lss] buffer the fusion Linear reads (no torch.cat)
    proj = torch.baddbmm(prepared["bev_ob"], bev.view(2, B * Q, E), prepared["bev_owt"]).view(2, B, Q, E)
    cat = torch.empty(B, Q, 3 * E, device=x1.device, dtype=torch.float32)
    add_ln(proj[0], self.norm_radar_bev, residual=x1, out=cat[..., E:2 * E])
    add_ln(proj[1], self.norm_lss_bev, residual=x1, out=cat[..., 2 * E:])
    add_ln(partials, self.norm2, residual=x1, bias=self.mixing.out_proj.bias, num_partials=partials.shape[0],
           out=cat[..., :E], a_scale=p_scale)
    f = add_ln(self.fusion(cat), self.norm_fusion)
    ffn_lin = self.ffn.layers[1](F.relu(self.ffn.layers[0][0](f)))
    x3 = add_ln(ffn_lin, self.norm3, residual=f)
    # first Linear of the cls and reg branches as one GEMM
    cb, rg = self.cls_branch, self.reg_branch
    c0r0 = F.linear(x3, prepared["c0r0_w"], prepared["c0r0_b"])
    c = add_ln(c0r0[..., :E], cb[1], relu=True)
    c = add_ln(cb[3](c), cb[4], relu=True)
    cls_score = cb[6](c)
    delta = rg[4](F.relu(rg[2](F.relu(c0r0[..., E:]))))
    bbox_pred, bbox_xy = refine_fused(qb, delta, meta["time_diff_safe"], self.num_ray)
    if stages is not None:
        stages.update(position_encoder=x - query_feat, self_attn=x + attn, sampling_radar_bev=proj[0] + x1,
                      sampling_lss_bev=proj[1] + x1, sampling=sampled_feat,
                      mixing=x1 + p_scale * partials.sum(0).view_as(x1) + self.mixing.out_proj.bias, ffn=f + ffn_lin)
    self.last_bbox_xy = bbox_xy
    return x3, cls_score, bbox_pred



T.register_alternate_plan("reference_ops", layer_reference_ops)
T.register_alternate_plan("library_chain", layer_library_chain)
