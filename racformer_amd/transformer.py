"""MI355X-native counterparts of the reference's decoder modules
(``models/racformer_transformer.py``, ``models/bev_self_attention.py``,
``models/sparsebev_sampling.py``): same class names, constructor signatures, forward signatures
and ``state_dict`` keys, so a reference checkpoint's
``pts_bbox_head.transformer.*`` tensors load unchanged and configs can name the classes as before.

What is different is the execution plan, designed for one MI355X per sample:
  * the two gather operators run as hand-written HIP kernels through the C-ABI
    (``rac_msmv_fwd`` writing the mixing-ready ``[B,Q,G,T*P,C]`` layout, ``rac_msda_fwd``);
  * everything that does not depend on the queries is computed ONCE per forward instead of once
    per decoder layer -- the six layers share one set of weights
    (racformer_transformer.py:84-89), so ``temporal_encoder(radar_bev)``, ``bev + pos`` and
    ``value_proj`` are identical in all six iterations (254 GFLOP/layer in the reference);
  * the pyramid regroup is one HIP transpose kernel (``rac_regroup_fwd``);
  * no host<->device traffic inside the layer loop (the reference uploads ``linspace`` and shape
    tensors every layer, racformer_transformer.py:395,515, bev_self_attention.py:189-190).
There is no CPU fallback: inputs must live on the GPU and the HIP library must be built.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .bbox_utils import decode_bbox, inverse_sigmoid, theta_d2xy_coods, xy2theta_d_coods
from .fused import (SPLIT_ACT_SCALE, SPLIT_BIAS_PAD, SPLIT_SLICE, ConvImage, act_image, add_ln, bev_sampling_fused, bev_sampling_multi_fused,
                    box_prep, conv_direct, quantize_values_i16, upsample2x_image,
                    generator_fused, gru_gate_fused, layer_boundary_fused, mixing_fused, mixing_sampled_fused, mixing_sampled_supported, outproj_fused,
                    pack_conv3x3_weight,
                    pack_gemm_split_weight,
                    pe_head, refine_fused, row_gemm,
                    row_seg, rowgemm_launch, sampling4d_fused, sasa_fused, split_weight_f16, upsample2x_fused, value_proj_fused)
from .msda import msda_forward
from .msmv import msmv_forward

try:  # registry decorators are applied only if mmdet happens to be importable
    from mmdet.models.utils.builder import TRANSFORMER as _TRANSFORMER
    _register = _TRANSFORMER.register_module()
except Exception:  # noqa: BLE001
    def _register(cls):
        return cls

_TWO_PI = 2 * math.pi

# Alternate execution plans of a decoder layer that only the parity tests take -- the reference's op decomposition
# (``decoder_layer.fused = False``) and the library-GEMM chain (``decoder_layer.rowgemm = False``) -- live in tests/plans.py and
# register themselves here: the product module holds the plan it ships and the hooks, not the cross-checks.
_ALTERNATE_PLANS = {}


def register_alternate_plan(name, fn):
    _ALTERNATE_PLANS[name] = fn


def alternate_plan(name):
    if name not in _ALTERNATE_PLANS:
        raise RuntimeError(f"execution plan '{name}' is a cross-check of the parity tests: import tests/plans.py to register it")
    return _ALTERNATE_PLANS[name]


# ------------------------------------------------------------------------------- small blocks
class _AttnParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's key names (in_proj_*, out_proj.*)."""

    def __init__(self, embed_dims):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dims, embed_dims))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dims))
        self.out_proj = nn.Linear(embed_dims, embed_dims)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _MHAHolder(nn.Module):
    """Stands where mmcv's MultiheadAttention wrapper stands (key prefix ``attention.attn``)."""

    def __init__(self, embed_dims):
        super().__init__()
        self.attn = _AttnParams(embed_dims)


class _FFN(nn.Module):
    """mmcv FFN(embed, feedforward_channels) key layout: layers.0.0 / layers.1, + identity."""

    def __init__(self, embed_dims, feedforward_channels):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(0.0)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.0))

    def forward(self, x):
        return x + self.layers(x)


class _LearnedPositionalEncoding(nn.Module):
    """mmdet LearnedPositionalEncoding key layout (row_embed / col_embed)."""

    def __init__(self, num_feats, row_num_embed, col_num_embed):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)

    def grid(self, h, w):
        col = self.col_embed.weight[:w]
        row = self.row_embed.weight[:h]
        pos = torch.cat([col[None].expand(h, w, -1), row[:, None].expand(h, w, -1)], dim=-1)
        return pos.permute(2, 0, 1)  # [2*num_feats, h, w]


# ------------------------------------------------------------------------------- self attention
class ScaleAdaptiveSelfAttention(nn.Module):
    """racformer_transformer.py:282-335.  mask[b,h,i,j] = -||c_i-c_j|| * tau[b,h,i]."""

    def __init__(self, embed_dims=256, num_heads=8, dropout=0.1, pc_range=[], init_cfg=None):
        super().__init__()
        self.pc_range = pc_range
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.attention = _MHAHolder(embed_dims)
        self.gen_tau = nn.Linear(embed_dims, num_heads)
        self.fused = True

    @torch.no_grad()
    def init_weights(self):
        nn.init.zeros_(self.gen_tau.weight)
        nn.init.uniform_(self.gen_tau.bias, 0.0, 2.0)

    def wide_in_proj(self):
        """in_proj and gen_tau as one [776,256] GEMM operand (built once per forward)."""
        p = self.attention.attn
        return (torch.cat([p.in_proj_weight, self.gen_tau.weight], dim=0),
                torch.cat([p.in_proj_bias, self.gen_tau.bias], dim=0))

    def forward(self, query_bbox, query_feat, pre_attn_mask=None, prepared_w=None):
        if self.fused and pre_attn_mask is None and self.embed_dims // self.num_heads == 32:
            # one GEMM for in_proj + gen_tau, one HIP kernel for mask + QK^T + softmax + AV
            p = self.attention.attn
            if prepared_w is None:
                prepared_w = self.wide_in_proj()
            lin = F.linear(query_feat, prepared_w[0], prepared_w[1])
            E = self.embed_dims
            o = sasa_fused(lin[..., :3 * E], lin[..., 3 * E:], query_bbox.contiguous(), self.num_heads, self.pc_range)
            return query_feat + p.out_proj(o)
        return self.forward_unfused(query_bbox, query_feat, pre_attn_mask)

    def forward_unfused(self, query_bbox, query_feat, pre_attn_mask=None):
        B, Q, E = query_feat.shape
        Hn, d = self.num_heads, E // self.num_heads
        centers = decode_bbox(theta_d2xy_coods(query_bbox), self.pc_range)[..., :2]
        dist = -torch.cdist(centers, centers, compute_mode="donot_use_mm_for_euclid_dist")  # [B,Q,Q]
        tau = self.gen_tau(query_feat).permute(0, 2, 1)                                     # [B,H,Q]
        mask = dist[:, None] * tau[..., None]
        if pre_attn_mask is not None:
            mask[:, :, pre_attn_mask] = float("-inf")
        p = self.attention.attn
        qkv = F.linear(query_feat, p.in_proj_weight, p.in_proj_bias).view(B, Q, 3, Hn, d)
        q = qkv[:, :, 0].permute(0, 2, 1, 3) * math.sqrt(1.0 / d)
        k = qkv[:, :, 1].permute(0, 2, 1, 3)
        v = qkv[:, :, 2].permute(0, 2, 1, 3)
        attn = torch.softmax(mask + q @ k.transpose(-1, -2), dim=-1)
        o = (attn @ v).permute(0, 2, 1, 3).reshape(B, Q, E)
        return query_feat + p.out_proj(o)


# ------------------------------------------------------------------------------- keypoints
def make_sample_points(query_bbox, offset, pc_range):
    """sparsebev_sampling.py:8-25: p = xyz + R_z(yaw) (wlh * offset)."""
    d = decode_bbox(query_bbox, pc_range)
    xyz, wlh, ang = d[..., 0:3], d[..., 3:6], d[..., 6:7]
    delta = wlh[:, :, None, :] * offset[..., 0:3]
    c, s = torch.cos(ang)[..., None, :], torch.sin(ang)[..., None, :]
    rot = torch.cat([delta[..., 0:1] * c - delta[..., 1:2] * s,
                     delta[..., 0:1] * s + delta[..., 1:2] * c, delta[..., 2:3]], dim=-1)
    return xyz[:, :, None, :] + rot


COMPACT_BELOW_COVERAGE = 0.8


def compact_variant(coverage):
    """Which variant of the fused sampling kernel a rig gets: the one that sets points without any tap aside pays where the
    cameras leave part of the circle unseen (measured: 3-cam front rig, coverage 0.42, 76.8 -> 67.9 us per launch) and costs
    2.5 us of 89 where every point is live (6-cam ring, coverage 1.0).  None (no measurement staged): the kernel's own rule."""
    return None if coverage is None else bool(coverage < COMPACT_BELOW_COVERAGE)


def rig_coverage(lidar2img, num_cams, image_hw, pc_range):
    """Share of a ring of probe points (72 azimuths x 4 ranges x 2 heights inside pc_range) that at least one camera of the
    first frame sees, under the validity rule of sparsebev_sampling.py:66-80 -- the live-point fraction the sampling kernel
    will meet, measured on the sample's own projection matrices (host numpy, once per staged sample).  A 6-camera rig with
    failed cameras or a 4-camera rig is judged by what its cameras cover, not by their number."""
    l2i = np.asarray(lidar2img, dtype=np.float64).reshape(-1, 4, 4)[:num_cams]
    H, W = image_hw
    az = np.linspace(0.0, 2.0 * np.pi, 72, endpoint=False)
    rmax = 0.9 * min(abs(pc_range[0]), abs(pc_range[1]), pc_range[3], pc_range[4])
    pts = np.array([[r * np.cos(a), r * np.sin(a), z, 1.0] for a in az for r in (0.15 * rmax, 0.4 * rmax, 0.7 * rmax, rmax) for z in (0.0, 1.5)])
    cam = np.einsum("nij,pj->npi", l2i, pts)                       # [N, P, 4]
    z = cam[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = cam[..., 0] / np.maximum(z, 1e-5) / W, cam[..., 1] / np.maximum(z, 1e-5) / H
    seen = (z > 1e-5) & (u > 0) & (u < 1) & (v > 0) & (v < 1)
    return float(seen.any(axis=0).mean())


def _warp_to_polar(points_xy, vel, time_diff, pc_range):
    """velocity warp + normalise + polar (racformer_transformer.py:379-393 / :501-512).
    points_xy [B,Q,1,G,P,2] -> theta, d each [B,Q,T,G,P,1]."""
    shift = (vel[:, :, None, :] * time_diff[:, None, :, None])[:, :, :, None, None, :]   # [B,Q,T,1,1,2]
    xy = points_xy - shift
    x = (xy[..., 0:1] - pc_range[0]) / (pc_range[3] - pc_range[0])
    y = (xy[..., 1:2] - pc_range[1]) / (pc_range[4] - pc_range[1])
    pol = xy2theta_d_coods(torch.cat([x, y], dim=-1))
    return pol[..., 0:1], pol[..., 1:2]


class RaCFormerSampling(nn.Module):
    """Adaptive spatio-temporal sampling (racformer_transformer.py:338-427)."""

    def __init__(self, embed_dims=256, num_frames=4, num_groups=4, num_points=8, num_levels=4,
                 depth_num=15, pc_range=[], init_cfg=None):
        super().__init__()
        self.num_frames, self.num_points, self.num_groups = num_frames, num_points, num_groups
        self.num_levels, self.pc_range, self.depth_num = num_levels, pc_range, depth_num
        self.ray_points_offset = nn.Linear(embed_dims, depth_num)
        self.sampling_offset = nn.Linear(embed_dims, depth_num * num_groups * num_points * 3)
        self.scale_weights = nn.Linear(embed_dims, num_groups * num_frames * depth_num * num_points * num_levels)
        # a list: every call appends the (u, v, view / (N-1)) locations it sampled at, [S,Q,P,3] -- the kernel's own loc_out on
        # the fused path; the reference's DUMP hook (sparsebev_sampling.py:82-87).  Parity tests use them to show view flips.
        self.capture_loc = None
        # a list of u8 [S,Q,P] tensors, consumed one per call: the camera index every point is to be sampled in, replacing
        # the first-valid-view selection (parity tests impose the reference's own choices: tests/parity.py)
        self.force_views = None
        # True: the list is walked round and round instead of consumed (one entry per decoder layer, every forward the same:
        # what a captured plan needs, whose warm-up and capture forwards all have to see the imposed choices)
        self.force_views_cyclic = False
        self._force_i = 0
        self._last_forced = None

    def _next_forced(self):
        self._last_forced = None
        if not self.force_views:
            return None
        if self.force_views_cyclic:
            v = self.force_views[self._force_i % len(self.force_views)]
            self._force_i += 1
        else:
            v = self.force_views.pop(0)
        self._last_forced = v
        return v

    def init_weights(self):
        bias = self.sampling_offset.bias.data.view(self.depth_num * self.num_groups * self.num_points, 3)
        nn.init.zeros_(self.sampling_offset.weight)
        nn.init.uniform_(bias[:, 0:3], -0.5, 0.5)

    def forward(self, query_ray, query_feat, mlvl_feats, img_metas, d_region=0.1, linear_out=None, debug=False,
                box_table=None):
        """One fused HIP kernel (rac_sampling4d_fwd).  ``linear_out`` = (offsets, ray logits, scale
        logits) if the caller already ran the three Linears as part of a wider GEMM."""
        image_h, image_w, _ = img_metas[0]["img_shape"][0]
        if linear_out is None:
            linear_out = (self.sampling_offset(query_feat), self.ray_points_offset(query_feat),
                          self.scale_weights(query_feat))
        off, ray, sc = linear_out
        res = sampling4d_fused(mlvl_feats, query_ray.contiguous(), off, ray, sc, img_metas[0]["time_diff"],
                               img_metas[0]["lidar2img"], self.num_frames, self.num_groups, self.num_points,
                               self.depth_num, self.pc_range, d_region, image_h, image_w,
                               debug=debug or self.capture_loc is not None, box_table=box_table, view_in=self._next_forced(),
                               compact=compact_variant(img_metas[0].get("_rac_coverage")))
        if self.capture_loc is not None:
            self.capture_loc.append(res[1])
            return res if debug else res[0]
        return res


    def forward_into_mixing(self, query_ray, mlvl_feats, img_metas, d_region, linear_out, box_table, params, out_points=128):
        """The sampling of forward() inside the AdaptiveMixing kernel (rac_mixing_sampled_fwd): -> the mixing output's f16 line
        image for outproj_fused; the sampled features stay on chip.  Same hooks as forward() (capture_loc, force_views)."""
        image_h, image_w, _ = img_metas[0]["img_shape"][0]
        off, ray, sc = linear_out
        res = mixing_sampled_fused(mlvl_feats, query_ray.contiguous(), off, ray, sc, img_metas[0]["time_diff"], img_metas[0]["lidar2img"],
                                   self.num_frames, self.num_groups, self.num_points, self.depth_num, self.pc_range, d_region, image_h,
                                   image_w, params, out_points=out_points, debug=self.capture_loc is not None, box_table=box_table,
                                   view_in=self._next_forced())
        if self.capture_loc is not None:
            self.capture_loc.append(res[1])
            return res[0]
        return res


def sampling_4d(sample_points, mlvl_feats, scale_weights, lidar2img, image_h, image_w, aggregate=True,
                eps=1e-5, loc_tap=None, view_in=None):
    """sparsebev_sampling.py:28-134 on the HIP msmv operator.
    sample_points [B,Q,T,G,P,3]; mlvl_feats[l] [B*T*G,N,H,W,C] channel-last; scale_weights
    [B,Q,G,T,P,L]; lidar2img [B,T*N,4,4] -> [B,Q,G,T*P,C].  Projection, validity, first-valid-view
    selection and the (b,g,t)-vs-(b,t,g) weight slot order (:113-120) are as in the reference."""
    if not aggregate:
        raise NotImplementedError("sampling_4d(aggregate=False) is not on the inference path")
    B, Q, T, G, P, _ = sample_points.shape
    N = lidar2img.shape[1] // T
    m = lidar2img.view(B, T, N, 1, 1, 4, 4)
    p = sample_points.permute(0, 2, 1, 3, 4, 5).reshape(B, T, 1, Q, G * P, 3)
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    cx = m[..., 0, 0] * x + m[..., 0, 1] * y + m[..., 0, 2] * z + m[..., 0, 3]
    cy = m[..., 1, 0] * x + m[..., 1, 1] * y + m[..., 1, 2] * z + m[..., 1, 3]
    homo = m[..., 2, 0] * x + m[..., 2, 1] * y + m[..., 2, 2] * z + m[..., 2, 3]     # [B,T,N,Q,GP]
    hz = torch.clamp(homo, min=eps)
    u = cx / hz / image_w
    v = cy / hz / image_h
    valid = (homo > eps) & (v > 0.0) & (v < 1.0) & (u > 0.0) & (u < 1.0)
    i_view = own_view = torch.argmax(valid.to(torch.uint8), dim=2, keepdim=True)      # first valid / 0
    if view_in is not None:   # imposed camera choice, u8 [S=(b,t,g),Q,P] -> [B,T,1,Q,GP]
        i_view = view_in.long().view(B, T, G, Q, P).permute(0, 1, 3, 2, 4).reshape(B, T, 1, Q, G * P)
    u_sel = torch.gather(u, 2, i_view)[:, :, 0]
    v_sel = torch.gather(v, 2, i_view)[:, :, 0]                                      # [B,T,Q,GP]
    loc = torch.stack([u_sel, v_sel, i_view[:, :, 0].to(u.dtype) / (N - 1)], dim=-1)
    loc = loc.view(B, T, Q, G, P, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * T * G, Q, P, 3).contiguous()
    if loc_tap is not None:   # (reports this implementation's OWN camera choice beside the sampled u, v, as the kernel does)
        own = torch.stack([u_sel, v_sel, own_view[:, :, 0].to(u.dtype) / (N - 1)], dim=-1)
        loc_tap.append(own.view(B, T, Q, G, P, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * T * G, Q, P, 3))
    L = scale_weights.shape[-1]
    w = scale_weights.reshape(B, Q, G, T, P, L).permute(0, 2, 3, 1, 4, 5).reshape(B * G * T, Q, P, L).contiguous()
    return msmv_forward(mlvl_feats, loc, w, out_layout=_lib.OUT_BQGTPC, num_frames=T, num_groups=G)


# ------------------------------------------------------------------------------- BEV branch
class ConvGRUCell(nn.Module):
    def __init__(self, input_channels, hidden_channels, kernel_size):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.gates_conv = nn.Conv2d(input_channels + hidden_channels, 3 * hidden_channels,
                                    kernel_size=kernel_size, padding=kernel_size // 2)
        self.matching_layer = nn.Conv2d(hidden_channels, input_channels, 1)

    def gates(self, x, h_prev):
        return self.gates_conv(torch.cat([x, self.matching_layer(h_prev)], dim=1))

    def forward(self, x, h_prev):
        gates = self.gates(x, h_prev)
        z_gate, r_gate, cand = torch.split(gates, self.hidden_channels, dim=1)
        z, r = torch.sigmoid(z_gate), torch.sigmoid(r_gate)
        cand = torch.tanh(cand + r * h_prev)
        return (1 - z) * h_prev + z * cand


class ConvGRU(nn.Module):
    """racformer_transformer.py:665-693: only frames t < min(4,T) are updated."""

    def __init__(self, input_channels, hidden_channels, kernel_size):
        super().__init__()
        self.convGRUCell = ConvGRUCell(input_channels, hidden_channels, kernel_size)
        self.hidden_channels = hidden_channels

    def forward(self, x):
        B, T, C, H, W = x.shape
        h = torch.zeros(B, self.hidden_channels, H, W, device=x.device, dtype=x.dtype)
        out = torch.zeros(B, T, self.hidden_channels, H, W, device=x.device, dtype=x.dtype)
        fused = x.is_cuda and x.dtype == torch.float32 and (self.hidden_channels * H * W) % 4 == 0
        for t in range(min(4, T)):
            if fused:
                # gates convolution (MIOpen), then the whole element-wise update in one launch, written straight into
                # the step's output slot (which is the next step's hidden state)
                h = gru_gate_fused(self.convGRUCell.gates(x[:, t], h), h, out[:, t])
            else:
                h = self.convGRUCell(x[:, t], h)
                out[:, t] = h
        return out


def convgru_fused_pack(gru, h, w):
    """Launch-lean form of the ConvGRU step (weights only): gates_conv(cat[x, matching(h_prev)]) is one 3x3 convolution of
    cat[x, h_prev] with the matching layer composed into the hidden half of the weights, plus a per-pixel map that carries
    the gates bias and the matching bias as seen through the zero-padded convolution.  -> (W' [3C, 2C, 3, 3], map [3C, h, w])."""
    cell = gru.convGRUCell
    C = cell.hidden_channels
    dev = cell.gates_conv.weight.device
    wg = cell.gates_conv.weight.detach().double().cpu()          # (float64 on the host: a one-off, weights only)
    bg = cell.gates_conv.bias.detach().double().cpu() if cell.gates_conv.bias is not None else wg.new_zeros(wg.shape[0])
    m = cell.matching_layer.weight.detach().double().cpu()[:, :, 0, 0]
    bm = cell.matching_layer.bias.detach().double().cpu() if cell.matching_layer.bias is not None else m.new_zeros(m.shape[0])
    if wg.shape[1] != 2 * C or m.shape != (C, C):
        return None, None
    wgx, wgh = wg[:, :C], wg[:, C:]
    wp = torch.cat([wgx, torch.einsum("omkl,mc->ockl", wgh, m)], dim=1).float().contiguous()
    bmap = F.conv2d(bm.view(1, C, 1, 1).expand(1, C, h, w).contiguous(), wgh, bg, padding=1)[0].float().contiguous()
    return wp.to(dev), bmap.to(dev)


def dead_frame_bias_map(w_hidden, b_hidden, H, W):
    """Contribution of a per-channel CONSTANT input b_hidden [Ch] through a 3x3 / pad 1 convolution with weights w_hidden
    [Cout, Ch, 3, 3] (float64): [H*W, Cout] -- the same everywhere except at the border, where the zero padding cuts taps off.
    What the frames past the ConvGRU's live ones contribute to the temporal-fusion convolution through their hidden half
    (racformer_transformer.py:674-693: those frames of the ConvGRU output are zero, so their hidden half is the bias of the
    convolution behind the resize)."""
    ch = w_hidden.shape[1]
    const = b_hidden.double().view(1, ch, 1, 1).expand(1, ch, H, W).contiguous()
    return F.conv2d(const, w_hidden.double(), None, padding=1)[0].reshape(w_hidden.shape[0], H * W).t().contiguous()


class RadarBEVTemporalEncoder(nn.Module):
    """racformer_transformer.py:618-663"""

    def __init__(self, embed_dims=256, hidden_dims=64, num_frames=8, kernel_size=3, downsample_ratio=2,
                 init_cfg=None):
        super().__init__()
        self.num_frames, self.embed_dims, self.hidden_dims = num_frames, embed_dims, hidden_dims
        self.convGRU = ConvGRU(hidden_dims, hidden_dims, kernel_size)
        self.temporal_fusion = nn.Conv2d(embed_dims + hidden_dims, embed_dims, kernel_size, padding=kernel_size // 2)
        self.downsample_ratio = downsample_ratio
        self.downsample = nn.Conv2d(embed_dims, hidden_dims, kernel_size=3, stride=downsample_ratio, padding=1)
        self.upsample = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                                      nn.Conv2d(hidden_dims, hidden_dims, kernel_size=3, padding=1))

    def init_weights(self):
        pass

    def hidden_stream(self, bev_feats):
        """downsample -> ConvGRU -> upsample: [B,T,C,H,W] -> ([B*T,C,H,W] view of the input, [B*T,hidden,H,W])."""
        B, T, C, H, W = bev_feats.shape
        r = self.downsample_ratio
        x = bev_feats.flatten(0, 1)   # NCHW: an NHWC (channels_last) pipeline measured 3 % slower end to end
        down = self.downsample(x).reshape(B, T, self.hidden_dims, H // r, W // r)
        return x, self.hidden_from_down(down, H, W)

    def hidden_from_down(self, down, H, W):
        """ConvGRU -> upsample on the downsampled maps [B,T,hidden,H/r,W/r] -> [B*T,hidden,H,W]."""
        B, T = down.shape[:2]
        r = self.downsample_ratio
        gru = self.convGRU(down)
        Tv = min(4, T)
        if not (down.is_cuda and down.dtype == torch.float32 and r == 2 and Tv < T):
            return self.upsample(gru.flatten(0, 1))
        return self.hidden_from_gru(gru[:, :Tv], T, H, W)

    def hidden_from_gru(self, gru_live, T, H, W):
        """ConvGRU leaves the frames t >= 4 at zero (:674-693): their upsample is zero and the 3x3 convolution of an
        all-zero map is exactly its bias, so only the live frames [B,Tv,hidden,H/2,W/2] go through the resize (one HIP
        launch instead of torch's generic kernel) and the convolution.  -> [B*T,hidden,H,W]."""
        B, Tv = gru_live.shape[:2]
        conv = self.upsample[1]
        hv = conv(upsample2x_fused(gru_live.reshape(B * Tv, self.hidden_dims, H // 2, W // 2)))
        hid = torch.empty(B, T, self.hidden_dims, H, W, device=gru_live.device, dtype=gru_live.dtype)
        hid[:, :Tv] = hv.view(B, Tv, self.hidden_dims, H, W)
        hid[:, Tv:] = (conv.bias if conv.bias is not None else hv.new_zeros(self.hidden_dims)).view(1, 1, -1, 1, 1)
        return hid.flatten(0, 1)

    def forward(self, bev_feats):
        B, T, C, H, W = bev_feats.shape
        x, hid = self.hidden_stream(bev_feats)
        cat = torch.cat([x, hid], dim=1)
        return self.temporal_fusion(cat).reshape(B, T, C, H, W)

    # temporal_fusion (193 of the encoder's 220 GFLOP) on the hand-written implicit-GEMM kernel (rac_conv3x3_fwd:
    # f16 matrix cores, hi/lo-split operands, fp32-convolution accuracy); no concatenation, no layout transposes,
    # output already channel-last for value_proj.
    fused_conv = True

    def downsample_pack(self, H=None, W=None):
        """Packed weights of the stride-2 downsample convolution for rac_conv3x3s2_fwd ({} if not 3x3 / 64 channels) and,
        for H x W input maps, the launch-lean ConvGRU operands (convgru_fused_pack)."""
        d = self.downsample
        if self.hidden_dims != 64 or d.kernel_size != (3, 3) or d.stride != (2, 2) or d.padding != (1, 1):
            return {}
        ws, alpha = pack_conv3x3_weight(d.weight, cout=64)
        if ws is None:
            return {}
        pack = dict(down_ws=ws, down_alpha=alpha)
        if H is not None and self.convGRU.convGRUCell.gates_conv.kernel_size == (3, 3):
            gw, gmap = convgru_fused_pack(self.convGRU, H // 2, W // 2)
            if gw is not None:
                pack.update(gru_w=gw, gru_bmap=gmap)
                # round 5, the branch on own kernels (rac_conv_direct_fwd): the gates convolution split into its x half (all live
                # frames in one launch, the bias map in its epilogue, channel-last) and its h half (the recurrence), the convolution
                # behind the resize, and the bound of |downsample(x)| per unit of max|x| that gives the downsampled image its scale
                up = self.upsample[1]
                hd = self.hidden_dims
                gx, gx_a = pack_conv3x3_weight(gw[:, :hd].contiguous(), cout=3 * hd)
                gh, gh_a = pack_conv3x3_weight(gw[:, hd:].contiguous(), cout=3 * hd)
                uw, uw_a = pack_conv3x3_weight(up.weight, cout=hd) if up.kernel_size == (3, 3) and up.padding == (1, 1) else (None, None)
                if gx is not None and gh is not None and uw is not None and hd == 64:
                    pack.update(gx_ws=gx, gx_alpha=gx_a, gh_ws=gh, gh_alpha=gh_a, up_ws=uw, up_alpha=uw_a,
                                gru_bmap_cl=gmap.reshape(3 * hd, -1).t().contiguous(),
                                down_l1=float(d.weight.detach().abs().sum(dim=(1, 2, 3)).max()),
                                down_bmax=float(d.bias.detach().abs().max()) if d.bias is not None else 0.0)
        return pack

    def temporal_bias_maps(self, H, W):
        """The per-pixel maps of rac_conv3x3_temporal_fwd for the UN-composed convolution: the bias for the live frames, the bias
        plus the constant hidden half's contribution for the frames past them ({} for shapes the kernel's map does not tile)."""
        tf, up = self.temporal_fusion, self.upsample[1]
        if (H * W) % 256 != 0 or tf.weight.shape[0] != 256:
            return {}
        b = tf.bias.detach().double() if tf.bias is not None else tf.weight.new_zeros(tf.weight.shape[0]).double()
        b_up = up.bias.detach().double() if up.bias is not None else b.new_zeros(self.hidden_dims)
        live = b.view(1, -1).expand(H * W, -1)
        dead = live + dead_frame_bias_map(tf.weight.detach().double()[:, self.embed_dims:], b_up, H, W).to(b.device)
        return dict(pixel_bias=live.float().contiguous(), pixel_bias_dead=dead.float().contiguous(), composed=False)

    def hidden_bound(self):
        """Upper bound of |hidden_stream(.)[1]| from the weights of the last convolution (inputs bounded by 1)."""
        conv = self.upsample[1]
        bound = float(conv.weight.detach().abs().sum(dim=(1, 2, 3)).max())
        return bound + (float(conv.bias.detach().abs().max()) if conv.bias is not None else 0.0)

    def fused_conv_supported(self, bev_feats):
        B, T, C, H, W = bev_feats.shape
        return (self.fused_conv and bev_feats.is_cuda and bev_feats.dtype == torch.float32 and self.embed_dims == 256
                and C == 256 and self.hidden_dims % 32 == 0 and (H * W) % 256 == 0 and W % 4 == 0 and W <= 128
                and self.temporal_fusion.kernel_size == (3, 3) and self.temporal_fusion.padding == (1, 1))

    def forward_channel_last(self, bev_feats, packed, hidden=None, q16=False):
        """-> [B*T, H, W, C] (channel-last); with ``q16`` (q int16 [B*T, H*W, 4, 64], scale [B*T, H*W, 4]): the fusion convolution's
        epilogue writes the int16 block storage of the BEV value stream (ConvImage.conv).  ``packed``: dict(ws, alpha = pack_conv3x3_weight(.), bound = hidden_bound(),
        optional pixel_bias [H*W, C] replacing the convolution's bias); ``hidden``: hidden_stream(bev_feats) if the
        caller already ran it (on a side stream)."""
        # |ConvGRU state| <= 1 (convex combinations of tanh values, zero start), bilinear resizing keeps that, so
        # |hid| <= max_row ||W_up||_1 + max|b_up|: known from the weights, only the input maps are scanned for the scale
        B, T, C, H, W = bev_feats.shape
        pb = packed.get("pixel_bias")
        img = ConvImage(B * T, H, W, C + self.hidden_dims, bev_feats.device)
        if hidden is not None:
            x, hid = hidden
            img.begin([x.contiguous()], packed["bound"]).pack(x.contiguous(), 0)
        else:
            x = bev_feats.flatten(0, 1).contiguous()
            img.begin([x], packed["bound"]).pack(x, 0)
            hd, Tv = self.hidden_dims, min(4, T)
            if (packed.get("gx_ws") is not None and packed.get("pixel_bias_dead") is not None and self.downsample_ratio == 2
                    and hd == 64 and C % 32 == 0 and H % 2 == 0 and W % 2 == 0
                    and tuple(packed["gru_bmap_cl"].shape) == ((H // 2) * (W // 2), 3 * hd)):
                return self._forward_own_gru(img, packed, B, T, Tv, C, H, W, q16)
            own_down = packed.get("down_ws") is not None and self.downsample_ratio == 2 and ((H // 2) * (W // 2)) % 128 == 0
            if own_down and packed.get("gru_w") is not None and Tv < T and tuple(packed["gru_bmap"].shape[1:]) == (H // 2, W // 2):
                # launch-lean ConvGRU: the downsample kernel writes the x half of every step's convolution input
                # [x_t | h_{t-1}], the gate kernel writes h_t into the next step's hidden half: per step one library
                # convolution + one HIP launch (matching layer, concatenation and bias adds are composed / folded)
                xh = torch.empty(B, T, 2 * hd, H // 2, W // 2, device=x.device, dtype=torch.float32)
                xh[:, 0, hd:].zero_()
                img.conv_s2(packed["down_ws"], packed["down_alpha"], self.downsample.bias, C, out=xh.view(B * T, 2 * hd, H // 2, W // 2))
                live = torch.empty(B, Tv, hd, H // 2, W // 2, device=x.device, dtype=torch.float32)
                for t in range(Tv):
                    gates = F.conv2d(xh[:, t], packed["gru_w"], None, padding=1)
                    gru_gate_fused(gates, xh[:, t, hd:], live[:, t], bias_map=packed["gru_bmap"],
                                   h_out2=xh[:, t + 1, hd:] if t + 1 < Tv else None)
                # hidden half of the image straight from the live frames: the last convolution without its bias, the bias added (and
                # the frames past the live ones synthesised from it) by the pack kernel -- no bias-add, copy and fill launches
                conv_up = self.upsample[1]
                hv = F.conv2d(upsample2x_fused(live.reshape(B * Tv, hd, H // 2, W // 2)), conv_up.weight, None, conv_up.stride,
                              conv_up.padding, conv_up.dilation, conv_up.groups)
                img.pack_live(hv, conv_up.bias, C, T)
                return img.conv(packed["ws"], packed["alpha"], None if pb is not None else self.temporal_fusion.bias, pb, q16=q16)
            else:
                if own_down:
                    # downsample (3x3, stride 2) on the image that is being built for the fusion convolution anyway
                    down = img.conv_s2(packed["down_ws"], packed["down_alpha"], self.downsample.bias, C)
                    down = down.view(B, T, hd, H // 2, W // 2)
                else:
                    down = self.downsample(x).reshape(B, T, hd, H // self.downsample_ratio, W // self.downsample_ratio)
                hid = self.hidden_from_down(down, H, W)
        img.pack(hid.contiguous(), C)
        return img.conv(packed["ws"], packed["alpha"], None if pb is not None else self.temporal_fusion.bias, pb, q16=q16)


    def _forward_own_gru(self, img, packed, B, T, Tv, C, H, W, q16):
        """Round 5: the whole ConvGRU branch on rac_conv_direct_fwd launches (no library convolution, no LDS staging: the maps are
        64 x 64 and the chain is sequential), every intermediate an activation image whose power-of-two scale follows
        from the weights (|h| <= 1; |downsample(x)| <= ||W||_1 max|x| + max|b|) -- the x half of ``img`` has been packed by the caller:
          downsample (stride 2) of the Tv live frames -> image;  gates convolution's x half for the Tv frames + bias map -> xpart;
          Tv recurrence launches (h half of the gates convolution + the GRU update in the epilogue; step 0 has h = 0: no K loop);
          2x bilinear resize -> image;  64 -> 64 convolution + bias -> the hidden chunks of the live frames of ``img``;
          temporal-fusion convolution: live frames all 10 chunks, the others 8 chunks + their own per-pixel map (their hidden half is
          the constant b_up)."""
        dev, hd = img.xs.device, self.hidden_dims
        h, w, cx = H // 2, W // 2, C // 32
        live = B * Tv
        fus = (img.amax, 1.0, 0.0)                                         # the fusion image's scale: its measured maximum
        dsc = (img.amax, packed["down_l1"], packed["down_bmax"])          # |downsample(x)| <= l1 * max|x| + max|b|
        one = (None, 0.0, 1.0)                                            # |ConvGRU state| <= 1 (and its bilinear resize)
        down_img, h_img = act_image("gru_down", live, h, w, hd, dev), act_image("gru_h", live, h, w, hd, dev)
        up_img = act_image("gru_up", live, H, W, hd, dev)
        conv_direct(_lib.CD_IMAGE, live, H, W, img.xs, img.cin // 32, cx, packed["down_ws"], packed["down_alpha"], hd, fus, conv_stride=2,
                    in_frames=(Tv, T, 0), bias=self.downsample.bias, out_img=down_img, out_chunks_total=hd // 32, out_scale=dsc)
        xpart = torch.empty(live, h * w, 3 * hd, device=dev, dtype=torch.float32)
        conv_direct(_lib.CD_F32, live, h, w, down_img, hd // 32, hd // 32, packed["gx_ws"], packed["gx_alpha"], 3 * hd, dsc,
                    out_f32=xpart, pixel_map=packed["gru_bmap_cl"])
        hs = torch.empty(live, h, w, hd, device=dev, dtype=torch.float32)
        for t in range(Tv):
            conv_direct(_lib.CD_GRU, B, h, w, h_img, hd // 32, 0 if t == 0 else hd // 32, packed["gh_ws"], packed["gh_alpha"], 3 * hd, one,
                        in_frames=(1, Tv, max(t - 1, 0)), out_img=h_img, out_chunks_total=hd // 32, out_frames=(1, Tv, t), out_scale=one,
                        xpart=xpart, xpart_frames=(1, Tv, t), h_prev=hs if t > 0 else None, h_prev_frames=(1, Tv, max(t - 1, 0)),
                        h_out=hs, h_out_frames=(1, Tv, t))
        upsample2x_image(hs, up_img, 1.0)
        conv_up = self.upsample[1]
        conv_direct(_lib.CD_IMAGE, live, H, W, up_img, hd // 32, hd // 32, packed["up_ws"], packed["up_alpha"], hd, one,
                    bias=conv_up.bias, out_img=img.xs, out_chunks_total=img.cin // 32, out_chunk0=cx, out_frames=(Tv, T, 0), out_scale=fus)
        return img.conv_temporal(packed["ws"], packed["alpha"], packed["pixel_bias"], packed["pixel_bias_dead"], C, T, Tv, q16=q16)


class BEVSelfAttention(nn.Module):
    """bev_self_attention.py:22-225 (deformable attention over T BEV maps + learned frame fusion)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2,
                 im2col_step=64, dropout=0.1, queue_weight=False, batch_first=True, norm_cfg=None,
                 init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.im2col_step, self.embed_dims, self.num_levels = im2col_step, embed_dims, num_levels
        self.num_heads, self.num_points, self.num_bev_queue = num_heads, num_points, num_bev_queue
        self.queue_weight = queue_weight
        if queue_weight:
            self.bev_queue_weight = nn.Linear(embed_dims, num_bev_queue)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        for m in (self.value_proj, self.output_proj) + ((self.bev_queue_weight,) if self.queue_weight else ()):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0.0)

    def project_value(self, value_maps, pos=None, channel_last=False, pos_term=None):
        """value_maps [B,T,C,H,W] ([B,T,H,W,C] with channel_last) (+ optional positional map [C,H,W] added to every frame) ->
        [B*T, H*W, heads, C/heads] (bev_self_attention.py:162-174).  Query-independent: the decoder
        calls this once per forward, not once per layer.  value_proj is linear, so
        value_proj(bev + pos) = bev^T W^T + (pos^T W^T + b): the frame-independent term is projected
        once ([HW,C] GEMM) and enters as the GEMM's additive operand, and the [C,HW] -> [HW,C]
        transpose is the GEMM's own operand transposition -- no bev+pos tensor, no permute copy."""
        if channel_last:
            B, T, H, W, C = value_maps.shape
        else:
            B, T, C, H, W = value_maps.shape
        wt = self.value_proj.weight.t()
        if pos_term is not None:       # value_proj(pos) [H*W, C] precomputed by the caller (depends on weights only)
            bias = pos_term.unsqueeze(0).expand(B * T, H * W, C)
        elif pos is None:
            bias = self.value_proj.bias.view(1, 1, C).expand(B * T, H * W, C)
        else:
            bias = self.value_proj(pos.reshape(C, H * W).t()).unsqueeze(0).expand(B * T, H * W, C)
        a = value_maps.reshape(B * T, H * W, C) if channel_last else value_maps.reshape(B * T, C, H * W).transpose(1, 2)
        v = torch.baddbmm(bias, a, wt.unsqueeze(0).expand(B * T, C, C))
        return v.view(B * T, H * W, self.num_heads, C // self.num_heads)

    def attend(self, query, value, sampling_locations, attention_weights, spatial_shapes, identity=None):
        """value: projected [B*T, HW, heads, D]; sampling_locations [B,Q,heads,T,P,2];
        attention_weights [B,Q,heads,T,L=1,P]."""
        B, Q, C = query.shape
        T, Hn, P = self.num_bev_queue, self.num_heads, self.num_points
        if identity is None:
            identity = query
        loc = sampling_locations.view(B, Q, Hn, T, self.num_levels, P, 2).permute(3, 0, 1, 2, 4, 5, 6) \
            .reshape(B * T, Q, Hn, self.num_levels, P, 2).contiguous()
        aw = attention_weights.view(B, Q, Hn, T, self.num_levels, P).permute(3, 0, 1, 2, 4, 5) \
            .reshape(B * T, Q, Hn, self.num_levels, P).contiguous()
        out = msda_forward(value, [list(spatial_shapes)], [0], loc, aw)                   # [B*T,Q,C]
        out = out.permute(1, 2, 0).reshape(Q, C, B, T)
        if self.queue_weight:
            qw = self.bev_queue_weight(query).permute(1, 0, 2).reshape(Q, 1, B, T)
            out = torch.sum(out * torch.softmax(qw, dim=-1), dim=-1)
        else:
            out = out.sum(-1) / T
        return self.output_proj(out.permute(2, 0, 1)) + identity

    def forward(self, query, value, sampling_locations, attention_weights, key_padding_mask=None,
                identity=None, spatial_shapes=None, **kwargs):
        """Reference signature: ``value`` is the raw [B,T,C,H,W] map stack (already bev+pos)."""
        if key_padding_mask is not None:
            raise NotImplementedError("key_padding_mask is unused on the RaCFormer path")
        return self.attend(query, self.project_value(value), sampling_locations, attention_weights,
                           spatial_shapes, identity)


class BEVSampling(nn.Module):
    """racformer_transformer.py:429-546"""

    def __init__(self, embed_dims=256, num_frames=4, num_points=8, num_heads=4, num_levels=4, pc_range=[],
                 spatial_shapes=(128, 128), depth_num=30, temp_radar=False, init_cfg=None):
        super().__init__()
        self.num_frames, self.num_points, self.num_heads = num_frames, num_points, num_heads
        self.num_levels, self.embed_dims, self.pc_range, self.depth_num = num_levels, embed_dims, pc_range, depth_num
        self.ray_points_offset = nn.Linear(embed_dims, depth_num)
        self.sampling_offset = nn.Linear(embed_dims, depth_num * num_heads * num_points * 2)
        self.scale_weights = nn.Linear(embed_dims, num_heads * num_levels * depth_num * num_points)
        self.positional_encoding = _LearnedPositionalEncoding(128, row_num_embed=spatial_shapes[1],
                                                              col_num_embed=spatial_shapes[0])
        self.attention = BEVSelfAttention(embed_dims=embed_dims, num_heads=4, num_levels=1,
                                          num_points=num_points * depth_num, num_bev_queue=num_frames,
                                          queue_weight=True)
        self.temp_radar = temp_radar
        if temp_radar:
            self.temporal_encoder = RadarBEVTemporalEncoder(embed_dims, 64, num_frames)

    def init_weights(self):
        bias = self.sampling_offset.bias.data.view(self.depth_num * self.num_heads * self.num_points, 2)
        nn.init.zeros_(self.sampling_offset.weight)
        nn.init.uniform_(bias[:, 0:2], -0.5, 0.5)
        self.attention.init_weights()

    def prepare_value(self, bev_feats, conv_pack=None, q16=False):
        """``q16`` (only honoured where the convolution's output IS the value stream, i.e. with a composed pack): returns
        ((q, scale), (H, W)) in the int16 block storage instead of the fp32 stream.
        Query-independent half of inner_forward (:484-485, :532-537 + value_proj): temporal
        encoder (radar only), + learned positional encoding, value projection.  ``conv_pack``: the packed
        temporal_fusion weights (fused convolution kernel, channel-last output) or None (MIOpen).  With
        ``conv_pack["pixel_bias"]`` the pack holds value_proj o temporal_fusion (composed_value_pack) and the
        convolution's output IS the value stream."""
        H, W = bev_feats.shape[-2:]
        if self.temp_radar and conv_pack is not None and conv_pack.get("ws") is not None and \
                self.temporal_encoder.fused_conv_supported(bev_feats):
            B, T = bev_feats.shape[:2]
            composed = bool(conv_pack.get("composed"))
            nhwc = self.temporal_encoder.forward_channel_last(bev_feats, conv_pack, q16=q16 and composed and self.attention.num_heads == 4)
            if isinstance(nhwc, tuple):
                return nhwc, (H, W)
            if composed:
                return nhwc.view(B * T, H * W, self.attention.num_heads, -1), (H, W)
            pos = self.positional_encoding.grid(H, W).to(bev_feats.dtype)
            return self.attention.project_value(nhwc.view(B, T, H, W, -1), pos, channel_last=True), (H, W)
        pos = self.positional_encoding.grid(H, W).to(bev_feats.dtype)
        if self.temp_radar:
            bev_feats = self.temporal_encoder(bev_feats)
        return self.attention.project_value(bev_feats, pos), (H, W)

    def composed_value_pack(self, H, W):
        """value_proj(temporal_fusion(cat) + pos) is one linear map of the concatenated maps: the 3x3 convolution with
        weights W' = W_v . W_conv (composed per tap) plus the per-pixel term P' = pos^T W_v^T + W_v b_conv + b_v.  Composing
        them once per set of weights (float64, then fp32) removes the 17-GFLOP value projection of the radar stream and
        its positional-term GEMM from every forward; the convolution kernel then writes the value stream
        [B*T, H*W, heads, 64] directly.  -> dict(ws, alpha, bound, pixel_bias) for prepare_value, or {} if it cannot be packed."""
        te, at = self.temporal_encoder, self.attention
        wv, bv = at.value_proj.weight.detach().double(), at.value_proj.bias.detach().double()
        wc = te.temporal_fusion.weight.detach().double()
        bc = te.temporal_fusion.bias.detach().double() if te.temporal_fusion.bias is not None else wv.new_zeros(wc.shape[0])
        ws, alpha = pack_conv3x3_weight(torch.einsum("oc,cikl->oikl", wv, wc).float())
        if ws is None:
            return {}
        pos = self.positional_encoding.grid(H, W).detach().double().reshape(-1, H * W)            # [C, HW]
        pixel_bias = pos.t() @ wv.t() + (wv @ bc + bv)                                                # [HW, C]
        # frames past the ConvGRU's live ones: their hidden half is the constant b_up; its contribution goes into THEIR map and
        # the kernel skips those two chunks for them (rac_conv3x3_temporal_fwd)
        up = te.upsample[1]
        b_up = up.bias.detach().double() if up.bias is not None else wv.new_zeros(te.hidden_dims)
        w_comp = torch.einsum("oc,cikl->oikl", wv, wc)
        dead = pixel_bias + dead_frame_bias_map(w_comp[:, te.embed_dims:], b_up, H, W).to(pixel_bias.device)
        return dict(ws=ws, alpha=alpha, bound=te.hidden_bound(), pixel_bias=pixel_bias.float().contiguous(),
                    pixel_bias_dead=dead.float().contiguous(), composed=True, **te.downsample_pack(H, W))

    def attend_prepared(self, query_ray, query_feat, value, hw, time_diff, d_region, linear_out=None, box_table=None):
        """One fused HIP kernel (rac_bev_sampling_fwd) + output_proj + identity."""
        if linear_out is None:
            linear_out = (self.sampling_offset(query_feat), self.ray_points_offset(query_feat),
                          self.scale_weights(query_feat), self.attention.bev_queue_weight(query_feat))
        off, ray, sc, qu = linear_out
        fused = bev_sampling_fused(value, hw, query_ray.contiguous(), off, ray, sc, qu, time_diff, self.num_frames,
                                   self.num_heads, self.num_points, self.depth_num, self.pc_range, d_region,
                                   box_table=box_table)
        return self.attention.output_proj(fused) + query_feat

    def forward(self, query_ray, query_feat, bev_feats, img_metas, d_region=0.1):
        value, hw = self.prepare_value(bev_feats)
        return self.attend_prepared(query_ray, query_feat, value, hw, img_metas[0]["time_diff"], d_region)


# ------------------------------------------------------------------------------- mixing
class AdaptiveMixing(nn.Module):
    """racformer_transformer.py:549-616"""

    def __init__(self, in_dim, in_points, n_groups=1, query_dim=None, out_dim=None, out_points=None):
        super().__init__()
        out_dim = out_dim if out_dim is not None else in_dim
        out_points = out_points if out_points is not None else in_points
        query_dim = query_dim if query_dim is not None else in_dim
        self.query_dim, self.in_dim, self.in_points, self.n_groups = query_dim, in_dim, in_points, n_groups
        self.out_dim, self.out_points = out_dim, out_points
        self.eff_in_dim, self.eff_out_dim = in_dim // n_groups, out_dim // n_groups
        self.m_parameters = self.eff_in_dim * self.eff_out_dim
        self.s_parameters = self.in_points * self.out_points
        self.total_parameters = self.m_parameters + self.s_parameters
        self.parameter_generator = nn.Linear(self.query_dim, self.n_groups * self.total_parameters)
        self.out_proj = nn.Linear(self.eff_out_dim * self.out_points * self.n_groups, self.query_dim)

    @torch.no_grad()
    def init_weights(self):
        nn.init.zeros_(self.parameter_generator.weight)

    SPLIT_K = 32

    def split_out_proj(self):
        """out_proj is a 900 x 32768 x 256 GEMM: K is 128x larger than N, and rocBLAS runs it as one
        un-split GEMM at 37 TFLOP/s.  Re-laid out once per forward as [S, N, K/S] it becomes a batched
        GEMM (S=32 partial products, summed) at 3x the rate and with a shorter fp32 accumulation chain."""
        w = self.out_proj.weight
        N, K = w.shape
        S = self.SPLIT_K if K % self.SPLIT_K == 0 else 1
        return w.view(N, S, K // S).permute(1, 0, 2).contiguous()

    def fused_supported(self, x):
        B, Q, G, P, C = x.shape
        return x.is_cuda and C == 64 and self.eff_out_dim == 64 and self.out_points == 128 and P <= 96

    # The two big Linears of the mixing (30 + 15 GFLOP per layer) on the f16 matrix cores at fp32-GEMM accuracy:
    # every operand is split as v = hi + lo (two f16, 22 significant bits) and the three leading products
    # hi*Whi + hi*Wlo + lo*Whi are accumulated in fp32 by ONE library GEMM over the K-concatenated operands
    # ([hi | hi | lo] x [Whi | Wlo | Whi]); measured error vs float64 equals the fp32 GEMM's (tools/exp_splitgemm.py)
    # at 2.3x the fp32-MFMA GEMM rate.  Power-of-two scalings keep the lo parts out of f16 subnormals and are
    # undone exactly by the GEMM's alpha.
    SPLIT_SLICE = SPLIT_SLICE   # K slice of the fp32 library path's split-K batches
    OUT_SLICE_K = 1024          # K per workgroup of rac_outproj_fwd: 8 row tiles x 32 slices = 256 workgroups at f8

    def split_packs(self, act_bound=None):
        """-> dict(gen_w [N,3K+64] f16, gen_alpha, out_w (line image [256, K/32, 64] f16), out_alpha, out_slices), or {} if f16 cannot hold the
        operands: a weight or bias overflows, |activation| bound * SPLIT_ACT_SCALE >= 6e4, or the generated
        parameters -- bounded by max_row ||W_row||_1 * act_bound + max|bias| -- could reach the f16 range that
        rac_mixing_fwd's RAC_MIX_F16X3 mode needs for S.  (One-off host reads; the result is cached.)"""
        gen = self.parameter_generator
        w = self.out_proj.weight
        N, K = w.shape
        if K % self.SPLIT_SLICE != 0 or K % self.OUT_SLICE_K != 0:
            return {}
        if act_bound is not None:
            param_bound = float(gen.weight.detach().abs().sum(dim=1).max()) * act_bound + float(gen.bias.detach().abs().max())
            if not (act_bound * SPLIT_ACT_SCALE < 6.0e4 and param_bound < 6.0e4):
                return {}
        gen_w, gen_alpha = split_weight_f16(gen.weight, gen.bias)
        if gen_w is None:
            return {}
        ow, out_alpha = pack_gemm_split_weight(w)        # line image [N, K/32, hi 32 | lo 32] for rac_outproj_fwd
        if ow is None:
            return {}
        gimg, gimg_alpha = pack_gemm_split_weight(gen.weight)      # line image [65536, 8, hi 32 | lo 32] for rac_generator_fwd
        if gimg is None:
            return {}
        return dict(gen_w=gen_w, gen_alpha=gen_alpha, gen_img=gimg, gen_img_alpha=gimg_alpha, out_w=ow, out_alpha=out_alpha,
                    out_slices=K // self.OUT_SLICE_K)

    def fused_supported_shape(self, in_points):
        return self.eff_in_dim == 64 and self.eff_out_dim == 64 and self.out_points == 128 and in_points == self.in_points <= 96

    def out_proj_partials(self, x, query, out_proj_split, params=None, packs=None, query_split=None):
        """Fused plan without the epilogue: generator GEMM -> MFMA mixing kernel -> split-K batched
        out_proj.  Returns the S partial products [S, B*Q, query_dim]; their sum + out_proj.bias + query
        is inner_forward's result (the caller folds that sum into its LayerNorm kernel).  ``params``:
        the generator output if the caller already produced it (on a side stream); with ``packs`` the partials
        still carry the power-of-two factor 1/packs["out_alpha"] (folded into the caller's add_ln).  ``packs`` +
        ``query_split`` ([B*Q, 3*query_dim] f16 from add_ln(split=True)): both GEMMs as split-precision
        f16-MFMA GEMMs (see split_packs)."""
        B, Q, G, P, C = x.shape
        timer = _lib.timer
        split = bool(packs) and query_split is not None
        params_scaled = False
        if params is None:
            own = split and query_split.shape[-1] == 2 * self.query_dim
            ev = timer.record("mixing_generator_gemm") if timer is not None and not own else None
            if ev:
                ev[0].record()
            if own:
                # line image: the hand-written split-precision GEMM (bias and alpha in its epilogue; timed inside)
                params = generator_fused(query_split, packs["gen_img"], self.parameter_generator.bias,
                                         packs["gen_img_alpha"]).view(B, Q, -1)
            elif split:
                # bias rides in the K-concatenated operands; alpha (a power of two) is applied by the mixing kernel
                params = torch.mm(query_split, packs["gen_w"].t(), out_dtype=torch.float32).view(B, Q, -1)
                params_scaled = True
            else:
                params = self.parameter_generator(query)
            if ev:
                ev[1].record()
        out = mixing_fused(x.contiguous(), params, P, G, self.out_points, split=split,
                           param_scale=packs["gen_alpha"] if split and params_scaled else 1.0, f16x3=split)
        if split:
            # hand-written split-K GEMM on the line images; the caller's add_ln applies packs["out_alpha"] to the summed partials
            return outproj_fused(out, packs["out_w"], packs["out_slices"])
        ev = timer.record("mixing_out_proj_gemm") if timer is not None else None
        if ev:
            ev[0].record()
        S_, N, k = out_proj_split.shape
        partials = torch.bmm(out.view(B * Q, S_, k).transpose(0, 1), out_proj_split.transpose(1, 2))
        if ev:
            ev[1].record()
        return partials

    def forward(self, x, query, out_proj_split=None):
        B, Q, G, P, C = x.shape
        assert G == self.n_groups and P == self.in_points and C == self.eff_in_dim
        params = self.parameter_generator(query)
        if out_proj_split is not None and self.fused_supported(x):
            # fused plan: one MFMA kernel for both mixings + norms + ReLUs, split-K out_proj
            out = mixing_fused(x.contiguous(), params, P, G, self.out_points)
            S_, N, k = out_proj_split.shape
            a3 = out.view(B * Q, S_, k).transpose(0, 1)
            proj = torch.bmm(a3, out_proj_split.transpose(1, 2)).sum(0) + self.out_proj.bias
            return query + proj.view(B, Q, N)
        params = params.reshape(B * Q, G, -1)
        M, S = params.split([self.m_parameters, self.s_parameters], 2)
        M = M.reshape(B * Q, G, self.eff_in_dim, self.eff_out_dim)
        S = S.reshape(B * Q, G, self.out_points, self.in_points)
        out = torch.matmul(x.reshape(B * Q, G, P, C), M)
        out = F.relu(F.layer_norm(out, [out.size(-2), out.size(-1)]))
        out = torch.matmul(S, out)
        out = F.relu(F.layer_norm(out, [out.size(-2), out.size(-1)]))
        if out_proj_split is None:
            return query + self.out_proj(out.reshape(B, Q, -1))
        S, N, k = out_proj_split.shape
        a3 = out.reshape(B * Q, S, k).transpose(0, 1)                       # [S, B*Q, k] strided view
        proj = torch.bmm(a3, out_proj_split.transpose(1, 2)).sum(0) + self.out_proj.bias
        return query + proj.view(B, Q, N)


# ------------------------------------------------------------------------------- decoder
class RaCFormerTransformerDecoderLayer(nn.Module):
    """racformer_transformer.py:145-279"""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_points_bev=4, num_levels=4, num_classes=10,
                 code_size=10, num_cls_fcs=2, num_reg_fcs=2, img_depth_num=3, bev_depth_num=5, num_ray=150,
                 pc_range=[], d_region_list=[0.15, 0.1, 0.1, 0.08, 0.08, 0.05], spatial_shapes=(128, 128),
                 init_cfg=None):
        super().__init__()
        self.embed_dims, self.num_classes, self.code_size, self.pc_range = embed_dims, num_classes, code_size, pc_range
        self.position_encoder = nn.Sequential(
            nn.Linear(3, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True),
            nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True))
        self.self_attn = ScaleAdaptiveSelfAttention(embed_dims, num_heads=8, dropout=0.1, pc_range=pc_range)
        self.sampling = RaCFormerSampling(embed_dims, num_frames=num_frames, num_groups=4, num_points=num_points,
                                          num_levels=num_levels, depth_num=img_depth_num, pc_range=pc_range)
        self.sampling_radar_bev = BEVSampling(embed_dims, num_frames=num_frames, num_heads=4,
                                              num_points=num_points_bev, num_levels=1, pc_range=pc_range,
                                              depth_num=bev_depth_num, spatial_shapes=spatial_shapes,
                                              temp_radar=True)
        self.sampling_lss_bev = BEVSampling(embed_dims, num_frames=num_frames, num_heads=4,
                                            num_points=num_points_bev, num_levels=1, pc_range=pc_range,
                                            depth_num=bev_depth_num, spatial_shapes=spatial_shapes)
        self.mixing = AdaptiveMixing(in_dim=embed_dims, in_points=num_points * num_frames * img_depth_num,
                                     n_groups=4, out_points=128)
        self.ffn = _FFN(embed_dims, 512)
        self.norm1 = nn.LayerNorm(embed_dims)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.norm3 = nn.LayerNorm(embed_dims)
        self.fusion = nn.Linear(embed_dims * 3, embed_dims)
        self.norm_radar_bev = nn.LayerNorm(embed_dims)
        self.norm_lss_bev = nn.LayerNorm(embed_dims)
        self.norm_fusion = nn.LayerNorm(embed_dims)
        cls_branch = []
        for _ in range(num_cls_fcs):
            cls_branch += [nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True)]
        cls_branch.append(nn.Linear(embed_dims, num_classes))
        self.cls_branch = nn.Sequential(*cls_branch)
        reg_branch = []
        for _ in range(num_reg_fcs):
            reg_branch += [nn.Linear(embed_dims, embed_dims), nn.ReLU(inplace=True)]
        reg_branch.append(nn.Linear(embed_dims, code_size))
        self.reg_branch = nn.Sequential(*reg_branch)
        self.d_region_list = d_region_list
        self.num_ray = num_ray
        self.fused = True  # False: the reference's op decomposition (torch keypoints + msmv / MSDA operators; tests/plans.py)
        # The mixing generator and out_proj as split-precision f16-MFMA GEMMs (AdaptiveMixing.split_packs).  False: fp32 rocBLAS.
        self.split_gemm = True
        # The ~17 small Linears of the layer with the add / LayerNorm / ReLU before them as rac_rowgemm_fwd launches (the
        # producer's normalisation runs as the prologue of its consumer GEMM): 21 launches per layer instead of ~50.
        # False: library GEMMs + rac_add_ln_fwd launches (the "library_chain" plan of tests/plans.py).
        self.rowgemm = True
        # parameter generator on the hand-written split-precision GEMM (rac_generator_fwd); False: hipBLASLt over K-concatenated images
        self.own_generator = True
        # ((next layer index, box pointer, shape, version), pe_head output, box table, the box tensor) handed from a layer's
        # boundary launch to the next call; the decoder clears it before layer 0, and it is only honoured for the matching
        # layer index and (live, unmodified) tensor
        self._carry = None
        # radar stream: value_proj composed into the temporal-fusion convolution (BEVSampling.composed_value_pack)
        self.compose_radar_value = True
        # storage of the two hoisted BEV value streams (value_proj's outputs): "f32" (default, the reference's fp32 maps,
        # bev_self_attention.py:162-174), or "i16" -- int16 mantissas with one power-of-two scale per (pixel, head) block of 64
        # channels (csrc/quant.hip), written by the two producers' own epilogues (rac_conv3x3_q16_fwd, rac_value_proj_q16_fwd): half
        # the bytes the BEV kernel gathers (80 -> 60 us per launch), fp32 arithmetic everywhere.  OPT-IN: both reference-initialised
        # rigs stay literal and the f8 random rig stays inside the fp32 path's tail budget (tests/test_lowprec_storage_gpu.py), but
        # the storage is not free -- the median box error of the random rig's last layer is ~3x the fp32 path's, and the reduced
        # 30-query head fixture misses the literal 1e-3 on one query (1.1e-3; fp32: 5.5e-4) -- so the product keeps the reference's
        # storage and bench.py reports the int16 mode beside the headline.
        self.value_storage = "f32"
        # True: the adaptive sampling runs INSIDE the mixing kernel (rac_mixing_sampled_fwd: the mixing workgroup of an item gathers its
        # own 96 points, bit for bit what rac_sampling4d_fwd writes; the 88 MB [B,Q,G,T*P,C] tensor of a layer is never written).
        # Correct (tests/test_fused_gpu.py::test_mixing_sampled_equals_sampling_then_mixing, every decoder-level parity test), but NOT
        # faster: 181 us against 88 + 92 us for the two kernels -- a workgroup that also holds the mixing's operands can keep 8 taps
        # in flight per lane on three workgroups per CU, the stand-alone gather 16 on five, and the texture path is
        # latency x concurrency bound -- so the default stays False; the environment variable is the A/B switch of bench.py.
        self.fuse_sampling_mixing = os.environ.get("RAC_FUSE_SAMPLING_MIXING", "0") == "1"
        # with "i16": the producers' own epilogues quantise (True) / separate rac_quant_i16_fwd launches over fp32 streams (False: tests)
        self.fused_q16_producers = True
        self._pack_cache = {}

    def _cached(self, key, params, fn):
        """Weight-derived operands (concatenations, re-layouts, f16 splits) are functions of the parameters only:
        built at the first forward and reused until a parameter is replaced or modified in place."""
        sig = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, fn())
            self._pack_cache[key] = hit
        return hit[1]

    @torch.no_grad()
    def init_weights(self):
        self.self_attn.init_weights()
        self.sampling.init_weights()
        self.mixing.init_weights()
        self.sampling_radar_bev.init_weights()
        self.sampling_lss_bev.init_weights()
        nn.init.constant_(self.cls_branch[-1].bias, float(-math.log((1 - 0.01) / 0.01)))
        nn.init.xavier_uniform_(self.fusion.weight)
        nn.init.constant_(self.fusion.bias, 0.0)

    def mfma_report(self, cfg, split):
        """(timer key, kernel description, algorithmic flops, executed flops, peak class 16 | 32) of the dense contractions
        of one forward that run on the matrix cores -- what bench.py prices against the gfx950 MFMA peaks."""
        Qn, E, G_, C_ = cfg.num_query, cfg.embed_dims, cfg.num_groups, cfg.channels
        Pin = cfg.num_points * cfg.num_frames * cfg.img_depth_num
        gen_cols = G_ * (C_ * C_ + 128 * Pin)
        bev_h, bev_w = cfg.bev_hw
        conv = 2.0 * cfg.num_frames * bev_h * bev_w * 256 * 320 * 9
        return [
            ("mixing_fwd", "mixing_c64_f16x3_kernel (hand-written; x@M: 6 bf16 products, S@Y: 3 f16 products)" if split
             else "mixing_c64_kernel (hand-written, v_mfma_f32_16x16x4_f32)",
             2.0 * Qn * G_ * (Pin * C_ * C_ + 128 * Pin * C_),
             2.0 * Qn * G_ * (6 * 96 * C_ * C_ + 3 * 128 * 96 * C_) if split else 2.0 * Qn * G_ * (96 * C_ * C_ + 128 * 96 * C_),
             16 if split else 32),
            ("mixing_sampled_fwd", "mixing_c64_f16x3_kernel<4> (the same two products; the workgroup gathers its own sampled features first: "
             "rac_mixing_sampled_fwd)", 2.0 * Qn * G_ * (Pin * C_ * C_ + 128 * Pin * C_),
             2.0 * Qn * G_ * (6 * 96 * C_ * C_ + 3 * 128 * 96 * C_), 16),
            ("mixing_generator_gemm", ("gemm_split_kernel (hand-written, 3 f16 products, loader waves + LDS-DMA ring)" if self.own_generator
                                       else "parameter_generator GEMM (hipBLASLt f16, K-concatenated hi/lo operands)") if split
             else "parameter_generator GEMM (rocBLAS fp32)",
             2.0 * Qn * E * gen_cols, 2.0 * Qn * gen_cols * ((3 * E + (0 if self.own_generator else 64)) if split else E),
             16 if split else 32),
            ("mixing_out_proj_gemm", "gemm_split_kernel (hand-written split-K GEMM, 3 f16 products, LDS-DMA staging)" if split
             else "out_proj split-K batched GEMM (rocBLAS fp32)",
             2.0 * Qn * (G_ * 128 * C_) * E, 2.0 * Qn * (G_ * 128 * C_) * E * (3 if split else 1), 16 if split else 32),
            ("temporal_fusion_conv", "conv3x3_f16x3_kernel (hand-written implicit GEMM, 3 f16 products; value_proj composed in)",
             conv, 3 * conv, 16)]

    def refine_bbox(self, bbox_proposal, bbox_delta):
        dz_new = torch.sigmoid(bbox_delta[..., 1:3] + inverse_sigmoid(bbox_proposal[..., 1:3]))
        theta = bbox_proposal[..., 0:1] + (torch.sigmoid(bbox_delta[..., 0:1]) * 2 - 1) / self.num_ray
        return torch.cat([theta, dz_new, bbox_delta[..., 3:]], dim=-1)

    def _sample(self, qb, x1, mlvl_feats, img_metas, d_region, linear_out, table):
        return self.sampling(qb, x1, mlvl_feats, img_metas, d_region=d_region, linear_out=linear_out, box_table=table)

    def _wide_linears(self):
        """The eleven Linear(256 -> .) layers that all read the post-norm1 query features, as one
        [2189,256] GEMM operand (one rocBLAS call per layer instead of eleven)."""
        mods = [self.sampling.sampling_offset, self.sampling.ray_points_offset, self.sampling.scale_weights]
        for x in (self.sampling_radar_bev, self.sampling_lss_bev):
            mods += [x.sampling_offset, x.ray_points_offset, x.scale_weights, x.attention.bev_queue_weight]
        w = torch.cat([m.weight for m in mods], dim=0)
        b = torch.cat([m.bias for m in mods], dim=0)
        widths = [m.weight.shape[0] for m in mods]
        return w, b, widths

    def prepare(self, lss_bev_feats, radar_bev_feats):
        """Layer-invariant tensors (computed once per forward)."""
        if self.value_storage not in ("f32", "i16"):
            raise RuntimeError(f"value_storage must be 'f32' or 'i16', got {self.value_storage!r}")
        # "i16": the two producers of the value streams quantise in their epilogues where they are the hand-written kernels
        # (rac_conv3x3_q16_fwd, rac_value_proj_q16_fwd); a stream that arrives as fp32 anyway goes through rac_quant_i16_fwd below
        # ("i16" is a preference: shapes the int16 BEV kernel is not built for -- streams of different geometry, heads of other than
        #  64 channels, CPU tensors, the unfused plans -- keep fp32 streams)
        rb_, lb_ = self.sampling_radar_bev, self.sampling_lss_bev
        i16 = (self.value_storage == "i16" and self.fused and self.rowgemm and radar_bev_feats.is_cuda and lss_bev_feats.is_cuda
               and radar_bev_feats.dtype == torch.float32 and lss_bev_feats.dtype == torch.float32
               and tuple(radar_bev_feats.shape) == tuple(lss_bev_feats.shape)
               and (rb_.num_frames, rb_.num_heads, rb_.num_points, rb_.depth_num) == (lb_.num_frames, lb_.num_heads, lb_.num_points, lb_.depth_num)
               and self.embed_dims == 64 * rb_.attention.num_heads == 64 * lb_.attention.num_heads)
        want_q16 = i16 and self.fused_q16_producers
        rbs = self.sampling_radar_bev
        te, up = rbs.temporal_encoder, rbs.temporal_encoder.upsample[1]
        conv_pack = None
        if radar_bev_feats.is_cuda and self.fused and te.fused_conv:
            Hr, Wr = radar_bev_feats.shape[-2:]
            cell = te.convGRU.convGRUCell
            conv_params = [te.temporal_fusion.weight, up.weight, te.downsample.weight, cell.gates_conv.weight,
                           cell.matching_layer.weight] + \
                [m.bias for m in (te.temporal_fusion, up, cell.gates_conv, cell.matching_layer, te.downsample) if m.bias is not None]
            if self.compose_radar_value:
                pe_, vp = rbs.positional_encoding, rbs.attention.value_proj
                conv_pack = self._cached(f"conv_value_pack_{Hr}x{Wr}", conv_params + [vp.weight, vp.bias, pe_.row_embed.weight,
                                                                                     pe_.col_embed.weight],
                                         lambda: rbs.composed_value_pack(Hr, Wr))
            if not conv_pack:
                def plain_pack():
                    ws, alpha = pack_conv3x3_weight(te.temporal_fusion.weight)
                    return dict(ws=ws, alpha=alpha, bound=te.hidden_bound(), **te.temporal_bias_maps(Hr, Wr), **te.downsample_pack(Hr, Wr))
                conv_pack = self._cached(f"conv_pack_{Hr}x{Wr}", conv_params, plain_pack)
        lbs = self.sampling_lss_bev
        if lss_bev_feats.is_cuda and self.fused:
            # the positional term value_proj(pos) of the LSS stream depends on weights only: built once, not per forward
            Hl, Wl = lss_bev_feats.shape[-2:]
            lvp, lpe = lbs.attention.value_proj, lbs.positional_encoding
            pos_term = self._cached(f"lss_pos_term_{Hl}x{Wl}", [lvp.weight, lvp.bias, lpe.row_embed.weight, lpe.col_embed.weight],
                                    lambda: lvp(lpe.grid(Hl, Wl).to(lvp.weight.dtype).reshape(-1, Hl * Wl).t()).contiguous())
            Bl, Tl, Cl = lss_bev_feats.shape[:3]
            vp_img = self._cached("lss_vp_img", [lvp.weight], lambda: pack_gemm_split_weight(lvp.weight)) \
                if Cl == 256 and lvp.weight.shape[0] == 256 and (Hl * Wl) % 32 == 0 and lss_bev_feats.dtype == torch.float32 \
                else (None, None)
            if vp_img[0] is not None:
                # hand-written split-precision GEMM straight from the channel-first maps (transpose, hi / lo split and the
                # positional term inside the kernel); alpha of the pack carries 1 / SPLIT_ACT_SCALE, which this kernel does not use
                # (value_storage "i16": its epilogue writes the int16 block storage, no fp32 stream and no quantiser launch)
                lss_value = value_proj_fused(lss_bev_feats.reshape(Bl * Tl, Cl, Hl, Wl).contiguous(), vp_img[0], vp_img[1] * SPLIT_ACT_SCALE,
                                             add=pos_term, q16=want_q16 and lbs.attention.num_heads == 4)
                if not isinstance(lss_value, tuple):
                    lss_value = lss_value.view(Bl * Tl, Hl * Wl, lbs.attention.num_heads, -1)
            else:
                lss_value = lbs.attention.project_value(lss_bev_feats, pos_term=pos_term)
            lss_hw = (Hl, Wl)
        else:
            lss_value, lss_hw = lbs.prepare_value(lss_bev_feats)
        radar_value, radar_hw = self.sampling_radar_bev.prepare_value(radar_bev_feats, conv_pack, q16=want_q16)
        rb, lb, mix = self.sampling_radar_bev, self.sampling_lss_bev, self.mixing
        wide_mods = [self.sampling.sampling_offset, self.sampling.ray_points_offset, self.sampling.scale_weights]
        for x in (rb, lb):
            wide_mods += [x.sampling_offset, x.ray_points_offset, x.scale_weights, x.attention.bev_queue_weight]
        w, b, widths = self._cached("wide", [p for m in wide_mods for p in (m.weight, m.bias)], self._wide_linears)
        ro, lo = rb.attention.output_proj, lb.attention.output_proj
        bev_owt, bev_ob = self._cached("bev_o", [ro.weight, lo.weight, ro.bias, lo.bias], lambda: (
            torch.stack([ro.weight.t(), lo.weight.t()]).contiguous(), torch.stack([ro.bias, lo.bias])[:, None, :].contiguous()))
        c0, r0 = self.cls_branch[0], self.reg_branch[0]
        c0r0_w, c0r0_b = self._cached("c0r0", [c0.weight, r0.weight, c0.bias, r0.bias], lambda: (
            torch.cat([c0.weight, r0.weight], dim=0), torch.cat([c0.bias, r0.bias], dim=0)))
        at = self.self_attn
        sasa_w = self._cached("sasa", [at.attention.attn.in_proj_weight, at.gen_tau.weight, at.attention.attn.in_proj_bias,
                                       at.gen_tau.bias], at.wide_in_proj)
        # K-slices of the two Linears with K > 256 (fusion: 768, FFN w2: 512): each slice runs as its own 256-deep GEMM in
        # the same launch, the consumer's prologue sums the partial outputs
        E_ = self.embed_dims
        w2_ = self.ffn.layers[1]
        kslices = self._cached("kslices", [self.fusion.weight, w2_.weight], lambda: (
            [self.fusion.weight[:, i * E_:(i + 1) * E_].contiguous() for i in range(self.fusion.weight.shape[1] // E_)],
            [w2_.weight[:, i * E_:(i + 1) * E_].contiguous() for i in range(w2_.weight.shape[1] // E_)]))
        packs = {}
        if self.split_gemm and radar_bev_feats.is_cuda and self.fused:
            # |norm1 output| <= sqrt(E) * max|gamma| + max|beta| bounds the generator's A operand
            packs = self._cached("split_packs", [mix.parameter_generator.weight, mix.parameter_generator.bias,
                                                 mix.out_proj.weight, self.norm1.weight, self.norm1.bias], lambda: mix.split_packs(
                float(self.norm1.weight.abs().max()) * math.sqrt(self.embed_dims) + float(self.norm1.bias.abs().max())))
        # the wide Linear on the same split-precision kernel as the generator (it reads the same f16 image of norm1's output)
        wide_img = self._cached("wide_img", [w], lambda: pack_gemm_split_weight(w)) if packs and self.own_generator else (None, None)
        out_proj_split = None if packs else self._cached("out_proj_split", [mix.out_proj.weight], mix.split_out_proj)
        value_scales = None
        if i16:
            # (a stream its producer did not quantise -- the library formulations of unusual shapes -- goes through rac_quant_i16_fwd)
            (radar_value, rsc), (lss_value, lsc) = [v if isinstance(v, tuple) else quantize_values_i16(v.contiguous())
                                                    for v in (radar_value, lss_value)]
            value_scales = (rsc, lsc)
        return dict(radar_value=radar_value, radar_hw=radar_hw, lss_value=lss_value, lss_hw=lss_hw, value_scales=value_scales,
                    wide_w=w, wide_b=b, wide_widths=widths, wide_img=wide_img, out_proj_split=out_proj_split, split_packs=packs,
                    sasa_w=sasa_w, bev_owt=bev_owt, bev_ob=bev_ob, c0r0_w=c0r0_w, c0r0_b=c0r0_b,
                    fusion_k=kslices[0], ffn2_k=kslices[1])

    def forward_fused(self, query_bbox, query_feat, mlvl_feats, img_metas, layer, prepared, stages=None, out_slots=None):
        """The layer as hand-written HIP kernels plus the three big library GEMMs of the mixing: every small Linear is
        a rac_rowgemm_fwd launch whose prologue performs the residual add / split-K sum / LayerNorm / ReLU that
        precedes it in the reference (racformer_transformer.py:239-279); same arithmetic, fp32 throughout."""
        if not self.rowgemm:      # (library GEMMs + rac_add_ln_fwd launches: a cross-check plan of the parity tests, tests/plans.py)
            return alternate_plan("library_chain")(self, query_bbox, query_feat, mlvl_feats, img_metas, layer, prepared, stages)
        self.wrote_slots = False
        meta = img_metas[0]
        time_diff, d_region = meta["time_diff"], self.d_region_list[layer]
        qb = query_bbox.contiguous()
        B, Q, E = query_feat.shape
        n = B * Q
        dev = query_feat.device
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)   # noqa: E731
        pe, p = self.position_encoder, self.self_attn.attention.attn
        packs = prepared.get("split_packs")
        # position encoder: relu(LN(Linear(3->256))) in one kernel (for layers > 0 already produced, with the box table, by
        # the previous layer's boundary launch), second Linear raw
        carried = self._carry if self._carry is not None and self._carry[0] == (layer, qb.data_ptr(), tuple(qb.shape), qb._version) \
            else None
        self._carry = None
        h = carried[1] if carried is not None else pe_head(qb[..., :3], pe[0], pe[1])
        y2 = new(n, E)
        rowgemm_launch([row_gemm([row_seg(h)], pe[3].weight, pe[3].bias, y2)], n)
        # x = query_feat + relu(LN(y2));  q|k|v|tau = x @ [in_proj; gen_tau]^T
        x, lin = new(B, Q, E), new(B, Q, 3 * E + self.self_attn.num_heads)
        rowgemm_launch([row_gemm([row_seg(y2, norm=pe[4], relu=True, post=query_feat.contiguous(), x_out=x)],
                                 prepared["sasa_w"][0], prepared["sasa_w"][1], lin)], n)
        # decode_bbox(theta_d2xy(.)) once for SASA and the 3 sampling kernels
        table = carried[2] if carried is not None else box_prep(qb, self.pc_range)
        o = sasa_fused(lin[..., :3 * E], lin[..., 3 * E:], qb, self.self_attn.num_heads, self.pc_range, box_table=table)
        attn = new(n, E)
        rowgemm_launch([row_gemm([row_seg(o)], p.out_proj.weight, p.out_proj.bias, attn)], n)
        # x1 = norm1(x + attn) (+ its f16 image for the generator GEMM);  the eleven Linears of the three sampling modules
        x1 = new(B, Q, E)
        own_gen = bool(packs) and self.own_generator
        x1_split = torch.empty(n, 2 * E if own_gen else 3 * E + SPLIT_BIAS_PAD, device=dev, dtype=torch.float16) if packs else None
        wide_n = prepared["wide_w"].shape[0]
        wimg, walpha = prepared.get("wide_img", (None, None))
        if own_gen and wimg is not None:
            # norm1 as its own row-wise launch (fp32 rows + f16 line image), then the 2189 outputs on the weights-stationary
            # split-precision kernel: 6 + 7 us against 30 for the fp32-MFMA row GEMM, whose 57 row tiles each re-read all weights
            add_ln(attn.view(B, Q, E), self.norm1, residual=x, out=x1, split=True, split_lines=True, split_out=x1_split)
            wide = generator_fused(x1_split, wimg, prepared["wide_b"], walpha, timer_name=None, ld_out=(wide_n + 3) // 4 * 4)
            wide = wide.view(B, Q, -1)[..., :wide_n]
        else:
            wide = new(B, Q, wide_n)
            rowgemm_launch([row_gemm([row_seg(attn, residual=x, norm=self.norm1, x_out=x1, split_out=x1_split, split_lines=own_gen)],
                                     prepared["wide_w"], prepared["wide_b"], wide)], n)
        lin = wide.split(prepared["wide_widths"], dim=-1)
        rb, lb = self.sampling_radar_bev, self.sampling_lss_bev
        r_off, r_ray, r_sc, r_qu = lin[3:7]
        l_off, l_ray, l_sc, l_qu = lin[7:11]
        bev = new(2, B, Q, E)
        same = prepared["radar_hw"] == prepared["lss_hw"] and prepared["radar_value"].dtype == prepared["lss_value"].dtype and \
            (rb.num_frames, rb.num_heads, rb.num_points, rb.depth_num) == (lb.num_frames, lb.num_heads, lb.num_points, lb.depth_num)
        if same:
            # radar and LSS stream in one launch (same queries, same boxes): the second stream's workgroups fill the CUs as
            # the first one's drain
            bev_sampling_multi_fused([(prepared["radar_value"], r_off, r_ray, r_sc, r_qu), (prepared["lss_value"], l_off, l_ray, l_sc, l_qu)],
                                     prepared["radar_hw"], qb, time_diff, rb.num_frames, rb.num_heads, rb.num_points, rb.depth_num,
                                     rb.pc_range, d_region, table, bev, value_scales=prepared.get("value_scales"))
        elif prepared.get("value_scales") is not None:
            raise RuntimeError("value_storage='i16' is served by the two-stream launch only")
        else:
            bev_sampling_fused(prepared["radar_value"], prepared["radar_hw"], qb, r_off, r_ray, r_sc, r_qu, time_diff,
                               rb.num_frames, rb.num_heads, rb.num_points, rb.depth_num, rb.pc_range, d_region,
                               box_table=table, out=bev[0])
            bev_sampling_fused(prepared["lss_value"], prepared["lss_hw"], qb, l_off, l_ray, l_sc, l_qu, time_diff,
                               lb.num_frames, lb.num_heads, lb.num_points, lb.depth_num, lb.pc_range, d_region,
                               box_table=table, out=bev[1])
        smp, mix = self.sampling, self.mixing
        if (self.fuse_sampling_mixing and own_gen and mix.out_points == 128 and mix.eff_in_dim == 64 and mix.eff_out_dim == 64
                and mix.in_points == smp.num_frames * smp.num_points * smp.depth_num and smp.num_groups == mix.n_groups
                and mixing_sampled_supported(mlvl_feats, smp.num_frames, smp.num_points * smp.depth_num)):
            # generator -> ONE kernel for the adaptive sampling and both mixings (the mixing workgroup gathers its own item while its
            # parameters stream in: the [B,Q,G,T*P,C] tensor is never written) -> out_proj
            params = generator_fused(x1_split, packs["gen_img"], mix.parameter_generator.bias, packs["gen_img_alpha"]).view(B, Q, -1)
            img = smp.forward_into_mixing(qb, mlvl_feats, img_metas, d_region, lin[0:3], table, params, out_points=mix.out_points)
            partials = outproj_fused(img, packs["out_w"], packs["out_slices"])
            sampled_feat = None
            if stages is not None:      # (the parity probes want the sampled features themselves: the stand-alone kernel, same choices)
                image_h, image_w, _ = img_metas[0]["img_shape"][0]
                sampled_feat = sampling4d_fused(mlvl_feats, qb, *lin[0:3], img_metas[0]["time_diff"], img_metas[0]["lidar2img"], smp.num_frames,
                                                smp.num_groups, smp.num_points, smp.depth_num, smp.pc_range, d_region, image_h, image_w,
                                                box_table=table, view_in=smp._last_forced)
        else:
            sampled_feat = self._sample(qb, x1, mlvl_feats, img_metas, d_region, lin[0:3], table)
            partials = self.mixing.out_proj_partials(sampled_feat, x1, prepared["out_proj_split"], None, packs, x1_split)
        p_scale = packs["out_alpha"] if packs else 1.0
        # both BEV output projections in one launch
        proj = new(2, n, E)
        ro, lo = rb.attention.output_proj, lb.attention.output_proj
        rowgemm_launch([row_gemm([row_seg(bev[0])], ro.weight, ro.bias, proj[0]),
                        row_gemm([row_seg(bev[1])], lo.weight, lo.bias, proj[1])], n)
        # fusion Linear over [norm2(mixing) | norm_radar(radar) | norm_lss(lss)]: the split-K sum + norm2 is its own row-wise
        # launch (16 partial rows per row are latency-bound inside a GEMM prologue: 32 us against 5 + 15), the two BEV
        # thirds are normalised in the GEMM's prologue
        x2 = add_ln(partials, self.norm2, residual=x1, bias=self.mixing.out_proj.bias, num_partials=partials.shape[0],
                    a_scale=p_scale)
        # The fusion Linear (K = 768) as three 256-deep GEMMs side by side in one launch (3 x 228 workgroups, each with a
        # third of the prologue and of the k-loop); its consumer sums the three partial outputs in its prologue.
        fk, f2k = prepared["fusion_k"], prepared["ffn2_k"]
        if len(fk) != 3 or len(f2k) != 2:
            raise RuntimeError("forward_fused: fusion must take 3 * embed_dims, feedforward_channels must be 2 * embed_dims")
        f_parts = new(3, n, E)
        rowgemm_launch([row_gemm([row_seg(x2)], fk[0], self.fusion.bias, f_parts[0]),
                        row_gemm([row_seg(proj[0], residual=x1, norm=self.norm_radar_bev)], fk[1], None, f_parts[1]),
                        row_gemm([row_seg(proj[1], residual=x1, norm=self.norm_lss_bev)], fk[2], None, f_parts[2])], n)
        # FFN: f = norm_fusion(f_raw); h1 = relu(W1 f); ffn_lin = W2 h1 (two K-slices); x3 = norm3(f + ffn_lin)
        f, h1, ffn_parts = new(n, E), new(n, 2 * E), new(2, n, E)
        w1, w2 = self.ffn.layers[0][0], self.ffn.layers[1]
        rowgemm_launch([row_gemm([row_seg(f_parts, num_partials=3, norm=self.norm_fusion, x_out=f)], w1.weight, w1.bias, h1, relu_from=0)], n)
        rowgemm_launch([row_gemm([row_seg(h1[:, :E])], f2k[0], w2.bias, ffn_parts[0]),
                        row_gemm([row_seg(h1[:, E:])], f2k[1], None, ffn_parts[1])], n)
        x3, c0r0 = new(B, Q, E), new(n, 2 * E)
        rowgemm_launch([row_gemm([row_seg(ffn_parts, num_partials=2, residual=f, norm=self.norm3, x_out=x3)], prepared["c0r0_w"],
                                 prepared["c0r0_b"], c0r0, relu_from=E)], n)       # (ReLU only on the reg half)
        # cls / reg branches side by side
        cb, rg = self.cls_branch, self.reg_branch
        c3, r2 = new(n, E), new(n, E)
        rowgemm_launch([row_gemm([row_seg(c0r0[:, :E], norm=cb[1], relu=True)], cb[3].weight, cb[3].bias, c3),
                        row_gemm([row_seg(c0r0[:, E:])], rg[2].weight, rg[2].bias, r2, relu_from=0)], n)
        cls_score = out_slots[0] if out_slots is not None else new(B, Q, self.num_classes)
        delta = new(B, Q, self.code_size)
        rowgemm_launch([row_gemm([row_seg(c3, norm=cb[4], relu=True)], cb[6].weight, cb[6].bias, cls_score),
                        row_gemm([row_seg(r2)], rg[4].weight, rg[4].bias, delta)], n)
        # refine_bbox of this layer + box table and position-encoder head of the next one, one launch
        bbox_pred, bbox_xy, next_table, next_h = layer_boundary_fused(qb, delta, meta["time_diff_safe"], self.num_ray,
                                                                      self.pc_range, pe[0], pe[1],
                                                                      xy_out=out_slots[1] if out_slots is not None else None)
        self.wrote_slots = out_slots is not None
        # (the carry holds bbox_pred itself: its storage cannot be freed and handed to another tensor while the key is live)
        self._carry = ((layer + 1, bbox_pred.data_ptr(), tuple(bbox_pred.shape), bbox_pred._version), next_h, next_table, bbox_pred)
        if stages is not None:
            mixed = x1 + p_scale * partials.sum(0).view_as(x1) + self.mixing.out_proj.bias
            stages.update(position_encoder=x - query_feat, self_attn=x + attn.view_as(x),
                          sampling_radar_bev=proj[0].view_as(x1) + x1, sampling_lss_bev=proj[1].view_as(x1) + x1,
                          sampling=sampled_feat, mixing=mixed, ffn=(f + ffn_parts.sum(0)).view_as(x1))
        self.last_bbox_xy = bbox_xy
        return x3, cls_score, bbox_pred

    def forward(self, query_bbox, query_feat, mlvl_feats, lss_bev_feats, radar_bev_feats, attn_mask, img_metas,
                layer=0, prepared=None, stages=None, out_slots=None):
        """``out_slots``: optional (cls_score [B,Q,classes], bbox_xy [B,Q,code]) destinations -- slices of the decoder's stacked
        outputs -- that the fused plan writes directly (no torch.stack afterwards)."""
        if prepared is None:
            prepared = self.prepare(lss_bev_feats, radar_bev_feats)
        if not self.fused:
            # the reference's op decomposition (torch keypoint chains + the msmv / MSDA operators + torch layers): a cross-check plan
            # of the parity tests, which register it (tests/plans.py) -- not product code
            return alternate_plan("reference_ops")(self, query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, layer, prepared, stages)
        if attn_mask is None and query_feat.is_cuda and self.embed_dims == 256 and self.mixing.in_points <= 96:
            return self.forward_fused(query_bbox, query_feat, mlvl_feats, img_metas, layer, prepared, stages, out_slots)
        # shapes the one-launch-per-stage plan is not built for (another embedding width, an attention mask, more than 96 sampling
        # points): the fused gather kernels with torch layers around them
        meta = img_metas[0]
        time_diff, d_region = meta["time_diff"], self.d_region_list[layer]
        query_pos = self.position_encoder(query_bbox[..., :3])
        query_feat = query_feat + query_pos
        sa = self.self_attn(query_bbox, query_feat, attn_mask, prepared.get("sasa_w"))
        query_feat = self.norm1(sa)
        lin = F.linear(query_feat, prepared["wide_w"], prepared["wide_b"]).split(prepared["wide_widths"], dim=-1)
        qb = query_bbox.contiguous()
        table = box_prep(qb, self.pc_range)       # decode_bbox(theta_d2xy(.)) once for the 3 sampling kernels
        radar_raw = self.sampling_radar_bev.attend_prepared(qb, query_feat, prepared["radar_value"],
                                                            prepared["radar_hw"], time_diff, d_region, lin[3:7], table)
        lss_raw = self.sampling_lss_bev.attend_prepared(qb, query_feat, prepared["lss_value"],
                                                        prepared["lss_hw"], time_diff, d_region, lin[7:11], table)
        sampled_feat = self._sample(qb, query_feat, mlvl_feats, img_metas, d_region, lin[0:3], table)
        query_radar_feat = self.norm_radar_bev(radar_raw)
        query_lss_feat = self.norm_lss_bev(lss_raw)
        mixed = self.mixing(sampled_feat, query_feat, prepared.get("out_proj_split"))
        query_feat = self.norm2(mixed)
        query_feat = self.norm_fusion(self.fusion(torch.cat((query_feat, query_radar_feat, query_lss_feat), dim=-1)))
        ffn_out = self.ffn(query_feat)
        query_feat = self.norm3(ffn_out)
        cls_score = self.cls_branch(query_feat)
        bbox_pred, bbox_xy = refine_fused(query_bbox, self.reg_branch(query_feat), meta["time_diff_safe"], self.num_ray)
        if stages is not None:
            stages.update(position_encoder=query_pos, self_attn=sa, sampling_radar_bev=radar_raw,
                          sampling_lss_bev=lss_raw, sampling=sampled_feat, mixing=mixed, ffn=ffn_out)
        self.last_bbox_xy = bbox_xy   # theta_d2xy_coods(bbox_pred), the per-layer output of the decoder (:134)
        return query_feat, cls_score, bbox_pred


def regroup_pyramid(mlvl_feats, num_cams, groups=4, out_dtype=torch.float32):
    """racformer_transformer.py:112-124 as ONE HIP transpose launch over all levels:
    [B,T*N,G*C,H,W] -> [B*T*G, N, H, W, C]  (rac_regroup_multi_fwd; levels whose sizes are not multiples of 4 go through
    the scalar per-level kernel rac_regroup_fwd)."""
    import ctypes
    feats, outs, dims = [], [], None
    for feat in mlvl_feats:
        B, TN, GC, H, W = feat.shape
        if TN % num_cams != 0 or GC % groups != 0:
            raise RuntimeError("regroup_pyramid: expected [B, T*N, G*C, H, W]")
        N, T, C = num_cams, TN // num_cams, GC // groups
        if dims is None:
            dims = (B, T, N, C)
        elif dims != (B, T, N, C):
            raise RuntimeError("regroup_pyramid: levels must share B, T*N and G*C")
        feat = feat.float().contiguous()
        _lib.require_gpu(feat, what="regroup_pyramid")
        feats.append(feat)
        outs.append(torch.empty(B * T * groups, N, H, W, C, device=feat.device, dtype=out_dtype))
    if not feats:
        return []
    B, T, N, C = dims
    code = _lib.RAC_F32 if out_dtype == torch.float32 else _lib.RAC_BF16
    if C % 4 == 0 and all((f.shape[3] * f.shape[4]) % 4 == 0 for f in feats) and len(feats) <= 8:
        L = len(feats)
        ins = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
        dst = (ctypes.c_void_p * L)(*[o.data_ptr() for o in outs])
        hw = (ctypes.c_int32 * (2 * L))(*[int(x) for f in feats for x in f.shape[3:5]])
        _lib.check(_lib.lib().rac_regroup_multi_fwd(L, ins, dst, hw, B, T, N, groups, C, code, _lib.stream_ptr()), "rac_regroup_multi_fwd")
        return outs
    for feat, dst in zip(feats, outs):
        H, W = feat.shape[3:5]
        _lib.check(_lib.lib().rac_regroup_fwd(_lib.ptr(feat), _lib.ptr(dst), B, T, N, groups, C, H, W, code, _lib.stream_ptr()),
                   "rac_regroup_fwd")
    return outs


class RaCFormerTransformerDecoder(nn.Module):
    """racformer_transformer.py:61-142"""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_points_bev=4, num_layers=6, num_levels=4,
                 num_classes=10, code_size=10, img_depth_num=3, bev_depth_num=5, pc_range=[], num_ray=150,
                 d_region_list=[0.15, 0.1, 0.1, 0.08, 0.08, 0.05], spatial_shapes=(128, 128), init_cfg=None,
                 num_cams=6):
        super().__init__()
        self.num_layers, self.pc_range, self.num_cams = num_layers, pc_range, num_cams
        # params are shared across all decoder layers (racformer_transformer.py:84-89)
        self.decoder_layer = RaCFormerTransformerDecoderLayer(
            embed_dims, num_frames, num_points, num_points_bev, num_levels, num_classes, code_size,
            img_depth_num=img_depth_num, bev_depth_num=bev_depth_num, num_ray=num_ray, pc_range=pc_range,
            d_region_list=d_region_list, spatial_shapes=spatial_shapes)
        self.feature_dtype = torch.float32
        # True: ``mlvl_feats`` arrive already in the sampling layout [B*T*G, N, H, W, C] (a producer that writes the
        # grouped channel-last pyramid directly skips the 1.47 GB regroup)
        self.pregrouped = False

    @torch.no_grad()
    def init_weights(self):
        self.decoder_layer.init_weights()

    def stage_metas(self, img_metas, B, device):
        """Host-side numerics of :99-109 (float64 timestamps -> float32 time_diff; lidar2img), one upload; also the
        clamped divisor of :266-269.  Like the reference, metas[0] receives ``time_diff`` and a device ``lidar2img``; a
        metas list that comes back with the same timestamps and its staged matrices is not staged again, one whose
        timestamps changed gets a new ``time_diff`` (the reference recomputes on every forward)."""
        m0 = img_metas[0]
        ts = np.array([m["img_timestamp"] for m in img_metas], dtype=np.float64)
        l2i_staged = isinstance(m0.get("lidar2img"), torch.Tensor) and m0["lidar2img"].device == device \
            and m0["lidar2img"].dim() == 4 and m0["lidar2img"].shape[0] == B
        if l2i_staged and "time_diff_safe" in m0 and np.array_equal(m0.get("_rac_staged_ts"), ts):
            return
        m0["_rac_staged_ts"] = ts.copy()
        ts = np.reshape(ts, [B, -1, self.num_cams])
        td = np.mean(ts[:, :1, :] - ts, axis=-1).astype(np.float32)
        td_safe = td.copy()
        td_safe[td_safe < 1e-5] = 1.0
        parts = [td.ravel(), td_safe.ravel()]
        l2i = None
        if not l2i_staged:
            l2i = np.asarray([m["lidar2img"].cpu().numpy() if isinstance(m["lidar2img"], torch.Tensor) else m["lidar2img"]
                              for m in img_metas]).astype(np.float32)
            parts.append(l2i.ravel())
        flat = torch.from_numpy(np.concatenate(parts))
        if device.type == "cuda":
            # one pinned staging block, one asynchronous copy: a pageable .to(device) would block the host until the
            # stream has drained, i.e. serialise this sample's launches behind the previous sample's kernels
            # (the caching host allocator keeps the pinned block until the copy has run)
            dev = flat.pin_memory().to(device, non_blocking=True)
        else:
            dev = flat.to(device)
        n0, n1 = td.size, td.size + td_safe.size
        if l2i is not None and "img_shape" in m0:
            h_img, w_img = m0["img_shape"][0][:2]
            m0["_rac_coverage"] = rig_coverage(l2i[0], self.num_cams, (h_img, w_img), self.pc_range)
        m0["_rac_meta_block"] = dev if l2i is not None else None      # (racformer_amd/graph.py restages into this block)
        m0["time_diff"] = dev[:n0].view(td.shape)
        m0["time_diff_safe"] = dev[n0:n1].view(td_safe.shape)
        if l2i is not None:
            m0["lidar2img"] = dev[n1:].view(l2i.shape)

    def forward(self, query_bbox, query_feat, mlvl_feats, lss_bev_feats, radar_bev_feats, attn_mask, img_metas,
                stages_per_layer=None):
        self.stage_metas(img_metas, query_bbox.shape[0], query_bbox.device)
        if self.pregrouped:
            # producer-side layout: the FPN already wrote [B*T*G, N, H, W, C]
            for f in mlvl_feats:
                if f.dim() != 5 or f.shape[1] != self.num_cams or not f.is_contiguous():
                    raise RuntimeError("pregrouped pyramid levels must be contiguous [B*T*G, N, H, W, C]")
        else:
            grouped = regroup_pyramid(mlvl_feats, self.num_cams, 4, self.feature_dtype)
            for lvl, g in enumerate(grouped):
                mlvl_feats[lvl] = g  # the reference mutates the caller's list too (:124)
        prepared = self.decoder_layer.prepare(lss_bev_feats, radar_bev_feats)
        self.decoder_layer._carry = None
        cls_scores, bbox_preds = [], []
        # the stacked outputs are allocated up front and every layer of the fused plan writes its slice (no torch.stack launches)
        stacked = None
        if query_feat.is_cuda and not torch.is_grad_enabled():
            B, Q = query_bbox.shape[:2]
            dl = self.decoder_layer
            stacked = (torch.empty(self.num_layers, B, Q, dl.num_classes, device=query_feat.device, dtype=torch.float32),
                       torch.empty(self.num_layers, B, Q, dl.code_size, device=query_feat.device, dtype=torch.float32))
        for i in range(self.num_layers):
            st = {} if stages_per_layer is not None else None
            self.decoder_layer.wrote_slots = False
            query_feat, cls_score, bbox_pred = self.decoder_layer(
                query_bbox, query_feat, mlvl_feats, lss_bev_feats, radar_bev_feats, attn_mask, img_metas,
                layer=i, prepared=prepared, stages=st, out_slots=(stacked[0][i], stacked[1][i]) if stacked is not None else None)
            if stacked is not None and not self.decoder_layer.wrote_slots:
                stacked = None                      # (a plan that does not write in place: fall back to stacking)
            if stages_per_layer is not None:
                stages_per_layer.append(st)
            query_bbox = bbox_pred.detach()
            cls_scores.append(cls_score)
            bbox_preds.append(self.decoder_layer.last_bbox_xy)
        if stacked is not None:
            return stacked
        return torch.stack(cls_scores), torch.stack(bbox_preds)


@_register
class RaCFormerTransformer(nn.Module):
    """racformer_transformer.py:17-58"""

    def __init__(self, embed_dims, num_frames=8, num_points=4, num_points_bev=4, num_layers=6, num_levels=4,
                 num_classes=10, code_size=10, img_depth_num=3, bev_depth_num=5, pc_range=[], num_ray=150,
                 d_region_list=[0.15, 0.1, 0.1, 0.08, 0.08, 0.05], spatial_shapes=(128, 128), init_cfg=None,
                 num_cams=6):
        assert init_cfg is None, "To prevent abnormal initialization behavior, init_cfg is not allowed to be set"
        super().__init__()
        self.embed_dims, self.pc_range, self.num_cams = embed_dims, pc_range, num_cams
        self.decoder = RaCFormerTransformerDecoder(
            embed_dims, num_frames, num_points, num_points_bev, num_layers, num_levels, num_classes, code_size,
            img_depth_num=img_depth_num, bev_depth_num=bev_depth_num, pc_range=pc_range, num_ray=num_ray,
            d_region_list=d_region_list, spatial_shapes=spatial_shapes, num_cams=num_cams)

    @torch.no_grad()
    def init_weights(self):
        self.decoder.init_weights()

    def forward(self, query_bbox, query_feat, mlvl_feats, lss_bev_feats, radar_bev_feats, attn_mask, img_metas,
                stages_per_layer=None, raw=False):
        """``raw``: return the decoder's stacked outputs without the nan_to_num of :58 -- for a caller that applies it together
        with its own element-wise tail (RaCFormer_head.forward: one rac_head_finish_fwd launch)."""
        cls_scores, bbox_preds = self.decoder(query_bbox, query_feat, mlvl_feats, lss_bev_feats,
                                              radar_bev_feats, attn_mask, img_metas, stages_per_layer)
        if raw:
            return cls_scores, bbox_preds
        return torch.nan_to_num(cls_scores), torch.nan_to_num(bbox_preds)
