"""Seeded synthetic nuScenes-shaped inputs for the decoder hot path.

All random data comes from numpy's PCG64 ``default_rng`` (bit-stable across hosts), so
the golden generator (build container) and the GPU box regenerate identical inputs
from a seed instead of shipping hundreds of MB of fixtures.

Shapes follow ``configs/racformer_r50_nuimg_704x256_f8.py:23-43`` of the reference
(Q=900 = 150 rays x 6 clusters, T=8 frames, 4 FPN levels, 6 cameras, 128x128 BEV).
"""
from dataclasses import dataclass, field, replace
import zlib

import numpy as np
import torch

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
D_REGION_LIST = [0.08, 0.07, 0.06, 0.05, 0.04, 0.03]


@dataclass(frozen=True)
class RigConfig:
    """Decoder-shape configuration (names follow the reference's config keys)."""
    batch: int = 1
    num_query: int = 900
    num_clusters: int = 6
    num_frames: int = 8
    num_cams: int = 6
    num_groups: int = 4
    embed_dims: int = 256
    num_levels: int = 4
    num_points: int = 4
    num_points_bev: int = 4
    img_depth_num: int = 3
    bev_depth_num: int = 5
    num_layers: int = 6
    num_classes: int = 10
    code_size: int = 10
    num_ray: int = 150
    image_hw: tuple = (256, 704)
    fpn_hw: tuple = ((64, 176), (32, 88), (16, 44), (8, 22))
    bev_hw: tuple = (128, 128)
    pc_range: tuple = tuple(PC_RANGE)
    d_region_list: tuple = tuple(D_REGION_LIST)
    three_cam_front: bool = False  # configs/..._3cam_3rad.py rig

    @property
    def channels(self):
        return self.embed_dims // self.num_groups

    def transformer_kwargs(self):
        return dict(embed_dims=self.embed_dims, num_frames=self.num_frames,
                    num_points=self.num_points, num_points_bev=self.num_points_bev,
                    num_layers=self.num_layers, num_levels=self.num_levels,
                    num_classes=self.num_classes, code_size=self.code_size,
                    img_depth_num=self.img_depth_num, bev_depth_num=self.bev_depth_num,
                    pc_range=list(self.pc_range), num_ray=self.num_ray,
                    d_region_list=list(self.d_region_list), spatial_shapes=tuple(self.bev_hw),
                    num_cams=self.num_cams)


F8 = RigConfig()
F8_3CAM = replace(F8, num_cams=3, three_cam_front=True)
# reduced shapes for committed goldens / fast CPU tests (embed_dims must stay 256: the
# reference hard-codes num_feats=128 for the BEV positional encoding).
SMALL = replace(F8, num_query=30, num_clusters=5, num_ray=6, num_frames=3, num_cams=3,
                image_hw=(64, 176), fpn_hw=((16, 44), (8, 22), (4, 11), (2, 6)),
                bev_hw=(16, 16))
SMALL6 = replace(SMALL, num_cams=6, num_frames=2)


def rng_normal(seed, shape, scale=1.0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(size=shape, dtype=np.float32)
    if scale != 1.0:
        a *= np.float32(scale)
    return a.astype(dtype, copy=False)


def rng_uniform(seed, shape, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return (rng.random(size=shape, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo))


# ------------------------------------------------------------------ cameras / metas
def ring_lidar2img(num_frames, num_cams, image_hw=(256, 704), three_cam_front=False,
                   fx=560.0, cam_height=1.5):
    """lidar2img = K [R|t] for a ring rig (lidar x-fwd, y-left, z-up -> cam z-fwd, x-right,
    y-down).  Identical across frames.  Returns list of T*N float64 4x4 arrays, frame-major
    (index t*N+n), the order the reference reshapes with (racformer_transformer.py:99-100)."""
    H, W = image_hw
    scale = W / 704.0
    f = fx * scale
    K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    if three_cam_front:
        assert num_cams == 3
        yaws = np.deg2rad([55.0, 0.0, -55.0])
    else:
        yaws = 2 * np.pi * np.arange(num_cams) / num_cams
    mats = []
    for yaw in yaws:
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[s, -c, 0.0], [0.0, 0.0, -1.0], [c, s, 0.0]], np.float64)
        pos = np.array([0.0, 0.0, cam_height])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ pos
        mats.append(K @ E)
    return [m.copy() for _ in range(num_frames) for m in mats]


def make_img_metas(cfg: RigConfig, sample=0):
    """One meta dict per sample with the three fields the decoder consumes
    (racformer_transformer.py:99,107,367).  ``sample`` > 0: another nuScenes-shaped sample -- the ego frame yawed by
    0.07 rad per sample index (lidar2img right-multiplied by the rotation) and another frame spacing / time origin, so that
    time_diff and every projection matrix differ from sample 0's (samples in flight per GPU carry metas of their own)."""
    T, N = cfg.num_frames, cfg.num_cams
    l2i = ring_lidar2img(T, N, cfg.image_hw, cfg.three_cam_front)
    dt, t0 = 0.5, 10.0
    if sample:
        yaw = 0.07 * sample
        c, s = np.cos(yaw), np.sin(yaw)
        ego = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
        l2i = [m @ ego for m in l2i]
        dt, t0 = 0.5 + 0.01 * (sample % 7), 10.0 + 0.25 * sample
    metas = []
    for b in range(cfg.batch):
        ts = [t0 - dt * t + 0.001 * n + 0.01 * b for t in range(T) for n in range(N)]
        metas.append(dict(
            img_timestamp=ts,
            lidar2img=[m.copy() for m in l2i],
            img_shape=[(cfg.image_hw[0], cfg.image_hw[1], 3)] * (T * N),
        ))
    return metas


# ------------------------------------------------------------------ feature maps
def _lerp_axis(c, axis, factor):
    """Exact-order float32 linear interpolation x``factor`` along ``axis`` (n+1 knots -> n*factor
    samples): two multiplies and one add per element, no reductions, so bit-stable across hosts."""
    n = c.shape[axis] - 1
    a = np.moveaxis(c, axis, -1)
    w1 = (np.arange(factor, dtype=np.float32) / np.float32(factor))
    w0 = np.float32(1.0) - w1
    out = a[..., :-1, None] * w0 + a[..., 1:, None] * w1
    return np.moveaxis(out.reshape(a.shape[:-1] + (n * factor,)), -1, axis)


def smooth_noise(seed, prefix, h, w, factor=4):
    """N(0,1) knots on a grid ``factor`` x coarser than (h,w), linearly interpolated: feature maps
    that vary over ~``factor`` pixels like real FPN / BEV features (white per-pixel noise makes the
    6-layer decoder amplify fp32 rounding ~3.5x per layer, far beyond the 1e-3 parity budget)."""
    ch, cw = -(-h // factor) + 1, -(-w // factor) + 1
    c = rng_normal(seed, tuple(prefix) + (ch, cw))
    c = _lerp_axis(_lerp_axis(c, -2, factor), -1, factor)
    return np.ascontiguousarray(c[..., :h, :w])


def make_pyramid(cfg: RigConfig, seed=0):
    """4 tensors [B, T*N, 256, H_l, W_l] (reference layout at the decoder input), smooth noise."""
    out = []
    for lvl, (h, w) in enumerate(cfg.fpn_hw):
        a = smooth_noise(seed * 1000 + 11 + lvl,
                         (cfg.batch, cfg.num_frames * cfg.num_cams, cfg.embed_dims), h, w)
        out.append(torch.from_numpy(a))
    return out


def make_bev(cfg: RigConfig, seed=0, which=0):
    h, w = cfg.bev_hw
    a = smooth_noise(seed * 1000 + 31 + which, (cfg.batch, cfg.num_frames, cfg.embed_dims), h, w)
    return torch.from_numpy(a)


# ------------------------------------------------------------------ Lift-Splat ranks (row f2: bev_pool_v2 at the f8 shape)
def make_lss_ranks(n_cams=6, D=96, H=16, W=44, grid=128, cell=0.8, depth=(1.0, 65.0), fx=560.0 / 16.0):
    """(ranks_depth, ranks_feat, ranks_bev) of a Lift-Splat frustum like the reference's LSSViewTransformer builds them
    (necks/view_transformer_racformer.py voxel_pooling_prepare_v2: every (cam, depth bin, h, w) frustum point lands in one BEV
    cell or outside the grid; points sorted by cell): ring rig of n_cams pinhole cameras at the feature stride 16."""
    d = depth[0] + (np.arange(D) + 0.5) * (depth[1] - depth[0]) / D
    u = (np.arange(W) + 0.5 - W / 2.0) / fx
    cams = 2 * np.pi * np.arange(n_cams) / n_cams
    dd, uu, cc = np.meshgrid(d, u, cams, indexing="ij")                  # [D, W, N]
    x = dd * np.cos(cc) + dd * uu * np.sin(cc)
    y = dd * np.sin(cc) - dd * uu * np.cos(cc)
    ix, iy = np.floor(x / cell + grid / 2).astype(np.int64), np.floor(y / cell + grid / 2).astype(np.int64)
    ok = (ix >= 0) & (ix < grid) & (iy >= 0) & (iy < grid)               # [D, W, N]
    n, dbin, h, w = np.meshgrid(np.arange(n_cams), np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    keep = ok[dbin, w, n]
    rd = (((n * D + dbin) * H + h) * W + w)[keep]
    rf = ((n * H + h) * W + w)[keep]
    rb = (iy[dbin, w, n] * grid + ix[dbin, w, n])[keep]
    order = np.argsort(rb, kind="stable")
    return tuple(torch.from_numpy(a[order].astype(np.int32)) for a in (rd, rf, rb))


# ------------------------------------------------------------------ queries
def head_query_grid(cfg: RigConfig):
    """Polar query grid of RaCFormer_head._init_layers / generate_points
    (racformer_head.py:51-79): [Q,10] = [theta, d, z=.5, 0, 0, logh=.2, 0, 0, vx=0, vy=0]
    with the Embedding's remaining columns drawn N(0,1) from a fixed seed."""
    Q, K = cfg.num_query, cfg.num_clusters
    n_ang = Q // K
    w = rng_normal(4242, (Q, 10))
    ang = np.linspace(0.0, 1.0, n_ang + 1, dtype=np.float32)[:-1]
    dist = np.linspace(0.0, 1.0, K + 2, dtype=np.float32)[1:-1]
    theta_d = np.stack(np.broadcast_arrays(ang[:, None], dist[None, :]), -1).reshape(-1, 2)
    w[:, 2] = 0.5
    w[:, 8:10] = 0.0
    w[:, 5] = 0.2
    w[:, :2] = theta_d
    return torch.from_numpy(w)


def make_queries(cfg: RigConfig, seed=0, feat_scale=0.1, vel_scale=1.0):
    """query_bbox [B,Q,10] (head grid; log-sizes / sin-cos tamed, small velocities so the
    temporal warp is exercised) and query_feat [B,Q,256] ~ feat_scale*N(0,1)."""
    qb = head_query_grid(cfg).numpy().copy()
    qb[:, 3:5] = 0.3 * qb[:, 3:5] + 0.5       # log w, log l
    qb[:, 8:10] = rng_normal(seed * 1000 + 51, (cfg.num_query, 2), vel_scale)
    qb = np.broadcast_to(qb[None], (cfg.batch,) + qb.shape).copy()
    qf = rng_normal(seed * 1000 + 52, (cfg.batch, cfg.num_query, cfg.embed_dims), feat_scale)
    return torch.from_numpy(qb), torch.from_numpy(qf)


# ------------------------------------------------------------------ weights by name
def _param_values(name, shape, seed):
    g_seed = (seed * 7919 + zlib.crc32(name.encode())) & 0x7FFFFFFF
    n = rng_normal(g_seed, tuple(shape))
    leaf = name.split(".")[-1]
    if name.endswith("gen_tau.bias"):
        return np.abs(n) + 0.5
    if name.endswith("gen_tau.weight"):
        return n * 0.01
    if "sampling_offset.bias" in name:
        return n * 0.3
    if "sampling_offset.weight" in name:
        return n * (0.25 / np.sqrt(shape[1]))
    if name.endswith("reg_branch.4.weight") or name.endswith("reg_branch.4.bias"):
        # rows 3:6 are log-sizes (exp'd, then multiply every sampling offset) and rows 8:10 are
        # velocities (x time_diff up to 3.5 s): N(0,1) there gives 20 m boxes / 10 m warps and a
        # heavy-tailed fp32-noise amplification no trained model has.  Keep them tame.
        v = n * (0.1 if leaf == "bias" else 1.0 / np.sqrt(shape[-1]))
        v[3:6] *= 0.25
        v[8:10] *= 0.25
        return v
    if "embed.weight" in name:           # row_embed / col_embed of the BEV pos-enc
        return n * 0.5
    if leaf == "weight" and len(shape) == 1:   # LayerNorm gamma
        return 1.0 + 0.1 * n
    if leaf in ("bias", "in_proj_bias"):
        return 0.1 * n
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return n * (1.0 / np.sqrt(max(fan_in, 1)))


def _param_values_torch_default(name, shape, seed):
    """The DISTRIBUTIONS torch's constructors draw from (nn.Linear / nn.Conv2d: weight and bias U(+-1/sqrt(fan_in));
    nn.MultiheadAttention: xavier-uniform in_proj, zero in_proj / out_proj bias; nn.Embedding N(0,1); nn.LayerNorm 1 / 0),
    drawn name-keyed from PCG64 like everything else here -- the state a reference model has before its own
    init_weights() runs (SURVEY 8d's second rig; the reference's init_weights is then applied on top by the golden
    generator, and what it wrote is kept in tests/golden/init_params_w*.npz)."""
    g_seed = (seed * 7919 + zlib.crc32(name.encode()) + 104729) & 0x7FFFFFFF
    leaf = name.split(".")[-1]
    if "embed.weight" in name:
        return rng_normal(g_seed, tuple(shape))
    if len(shape) == 1 and leaf == "weight":            # LayerNorm gamma
        return np.ones(shape, np.float32)
    if leaf == "in_proj_bias" or name.endswith("attn.out_proj.bias"):
        return np.zeros(shape, np.float32)
    if leaf == "bias":
        if any(k in name for k in ("norm", "position_encoder.1", "position_encoder.4", "cls_branch.1", "cls_branch.4")):
            return np.zeros(shape, np.float32)          # LayerNorm beta
        fan_in = _FAN_IN_OF_BIAS.get(name.rsplit(".", 1)[0].split("decoder_layer.")[-1])
        assert fan_in, name
        b = 1.0 / np.sqrt(fan_in)
        return rng_uniform(g_seed, tuple(shape), -b, b)
    fan_in = int(np.prod(shape[1:]))
    if leaf == "in_proj_weight":                        # xavier_uniform on the [3E, E] matrix
        b = np.sqrt(6.0 / (shape[0] + shape[1]))
    else:
        b = 1.0 / np.sqrt(fan_in)
    return rng_uniform(g_seed, tuple(shape), -b, b)


_FAN_IN_OF_BIAS = {}


def _note_fan_in(shapes):
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) > 1:
            _FAN_IN_OF_BIAS[k[:-len(".weight")].split("decoder_layer.")[-1]] = int(np.prod(shp[1:]))


@torch.no_grad()
def fill_params(module, seed=0, scheme="tamed_normal"):
    """Deterministic, name-keyed parameter fill: any two modules with the reference's
    state_dict keys get bit-identical weights without shipping
    a 28 M-parameter checkpoint.  ``scheme``: "tamed_normal" (the random-everything rig) or "torch_default"
    (the distributions of torch's constructors: the state before the reference's init_weights())."""
    named = sorted(module.named_parameters(), key=lambda kv: kv[0])
    _note_fan_in({k: tuple(p.shape) for k, p in named})
    fn = _param_values if scheme == "tamed_normal" else _param_values_torch_default
    for name, p in named:
        v = fn(name, tuple(p.shape), seed).astype(np.float32)
        p.copy_(torch.from_numpy(v).to(p.dtype))
    return module


# ------------------------------------------------------------------ state_dict contract
def transformer_param_shapes(cfg: RigConfig):
    """(name -> shape) of a RaCFormerTransformer state_dict (enumerated
    from the reference's constructors, models/racformer_transformer.py:170-212, 288-289,
    350-352, 451-468, 572-573, 631-639, 701-707 and models/bev_self_attention.py:96-101)."""
    E, T, G = cfg.embed_dims, cfg.num_frames, cfg.num_groups
    C = E // G
    heads_bev = 4
    hid = 64
    s = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    def ln(name, n=E):
        s[name + ".weight"] = (n,)
        s[name + ".bias"] = (n,)

    def conv(name, o, i, k):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    lin("position_encoder.0", E, 3); ln("position_encoder.1")
    lin("position_encoder.3", E, E); ln("position_encoder.4")
    s["self_attn.attention.attn.in_proj_weight"] = (3 * E, E)
    s["self_attn.attention.attn.in_proj_bias"] = (3 * E,)
    lin("self_attn.attention.attn.out_proj", E, E)
    lin("self_attn.gen_tau", 8, E)
    D, NP, L = cfg.img_depth_num, cfg.num_points, cfg.num_levels
    lin("sampling.ray_points_offset", D, E)
    lin("sampling.sampling_offset", D * G * NP * 3, E)
    lin("sampling.scale_weights", G * T * D * NP * L, E)
    Db, NPb = cfg.bev_depth_num, cfg.num_points_bev
    for x in ("sampling_radar_bev", "sampling_lss_bev"):
        lin(f"{x}.ray_points_offset", Db, E)
        lin(f"{x}.sampling_offset", Db * heads_bev * NPb * 2, E)
        lin(f"{x}.scale_weights", heads_bev * 1 * Db * NPb, E)
        s[f"{x}.positional_encoding.row_embed.weight"] = (cfg.bev_hw[1], 128)
        s[f"{x}.positional_encoding.col_embed.weight"] = (cfg.bev_hw[0], 128)
        lin(f"{x}.attention.bev_queue_weight", T, E)
        lin(f"{x}.attention.value_proj", E, E)
        lin(f"{x}.attention.output_proj", E, E)
    te = "sampling_radar_bev.temporal_encoder"
    conv(f"{te}.convGRU.convGRUCell.gates_conv", 3 * hid, 2 * hid, 3)
    conv(f"{te}.convGRU.convGRUCell.matching_layer", hid, hid, 1)
    conv(f"{te}.temporal_fusion", E, E + hid, 3)
    conv(f"{te}.downsample", hid, E, 3)
    conv(f"{te}.upsample.1", hid, hid, 3)
    in_points, out_points = NP * T * D, 128
    lin("mixing.parameter_generator", G * (C * C + in_points * out_points), E)
    lin("mixing.out_proj", E, C * out_points * G)
    lin("ffn.layers.0.0", 512, E)
    lin("ffn.layers.1", E, 512)
    for n in ("norm1", "norm2", "norm3", "norm_radar_bev", "norm_lss_bev", "norm_fusion"):
        ln(n)
    lin("fusion", E, 3 * E)
    lin("cls_branch.0", E, E); ln("cls_branch.1"); lin("cls_branch.3", E, E); ln("cls_branch.4")
    lin("cls_branch.6", cfg.num_classes, E)
    lin("reg_branch.0", E, E); lin("reg_branch.2", E, E); lin("reg_branch.4", cfg.code_size, E)
    return {"decoder.decoder_layer." + k: v for k, v in s.items()}


def make_state_dict(cfg: RigConfig, seed=0, scheme="tamed_normal"):
    """Name-keyed deterministic weights, identical to ``fill_params`` on a module with the
    same keys."""
    shapes = transformer_param_shapes(cfg)
    _note_fan_in(shapes)
    fn = _param_values if scheme == "tamed_normal" else _param_values_torch_default
    return {k: torch.from_numpy(fn(k, shp, seed).astype(np.float32)) for k, shp in shapes.items()}
