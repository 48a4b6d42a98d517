"""Fused per-layer kernels (rac_sampling4d_fwd, rac_bev_sampling_fwd) against the CPU oracle and
against the op-decomposed GPU path, on identical inputs."""
import numpy as np
import pytest
import torch

import plans
from oracle import restate as R
from racformer_amd import synthetic as syn
from racformer_amd.transformer import RaCFormerTransformer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(cfg, seed, wseed):
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    sd = R._sub({k: v.clone() for k, v in tr.state_dict().items()}, "decoder.decoder_layer.")
    tr = tr.to(DEV)
    qb, qf = syn.make_queries(cfg, seed)
    qf = qf * 5.0            # post-LayerNorm-like magnitude so offsets / logits are not degenerate
    metas = syn.make_img_metas(cfg)
    tr.decoder.stage_metas(metas, cfg.batch, torch.device(DEV))
    return tr, sd, qb, qf, metas


from dataclasses import replace as _replace
MANY_POINTS = _replace(syn.SMALL6, num_points=16, img_depth_num=4)      # P = 64: the tap table of 8 rows would not fit, 4 do


@pytest.mark.parametrize("cfg", [syn.SMALL, syn.SMALL6, syn.F8, syn.F8_3CAM, MANY_POINTS])
def test_sampling4d_fused(cfg):
    tr, sd, qb, qf, metas = _setup(cfg, 21, 22)
    layer = tr.decoder.decoder_layer
    feats_cpu = R.regroup_pyramid(syn.make_pyramid(cfg, 21), cfg.num_cams)
    feats = [f.to(DEV) for f in feats_cpu]
    d_region = cfg.d_region_list[2]
    with torch.no_grad():
        out, loc, w = layer.sampling(qb.to(DEV), qf.to(DEV), feats, metas, d_region=d_region, debug=True)
        out_unf = plans.sampling_reference_ops(layer.sampling, qb.to(DEV), qf.to(DEV), feats, metas, d_region=d_region)
        td = R.time_diff_from_metas(syn.make_img_metas(cfg), cfg.batch, cfg.num_cams)
        l2i = torch.from_numpy(np.asarray([m["lidar2img"] for m in syn.make_img_metas(cfg)]).astype(np.float32))
        pts, sw = R.image_keypoints(sd, qb, qf, td, d_region, cfg)
        B, Q, T, G, P, _ = pts.shape
        oloc, oview, _ = R.project_select(pts.reshape(B, Q, T, G * P, 3), l2i, cfg.image_hw[0], cfg.image_hw[1])
        oloc = oloc.reshape(B, T, Q, G, P, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * T * G, Q, P, 3)
        ref = R.sampling_4d(pts, feats_cpu, sw, l2i, cfg.image_hw[0], cfg.image_hw[1])
    torch.cuda.synchronize()
    loc, w, out, out_unf = loc.cpu(), w.cpu(), out.cpu(), out_unf.cpu()
    N = cfg.num_cams
    same_view = torch.round(loc[..., 2] * (N - 1)) == torch.round(oloc[..., 2] * (N - 1))
    assert same_view.float().mean().item() > 0.9995, same_view.float().mean().item()
    near = same_view & (oloc[..., 0] > 0) & (oloc[..., 0] < 1) & (oloc[..., 1] > 0) & (oloc[..., 1] < 1)  # in-image points
    assert (loc[..., :2] - oloc[..., :2])[near].abs().max().item() < 2e-5
    ow = sw.reshape(B, Q, G, T, P, -1).permute(0, 2, 3, 1, 4, 5).reshape(B * G * T, Q, P, -1)
    assert (w - ow).abs().max().item() < 1e-5
    # outputs: every (b,q,g,t) row whose 12 points chose the same cameras must match the oracle
    row_ok = same_view.reshape(B, T, G, Q, P).all(-1).permute(0, 3, 2, 1)            # [B,Q,G,T]
    err = (out - ref).reshape(B, Q, G, T, P, -1).abs().amax((-1, -2))
    assert err[row_ok].max().item() < 2e-4, err[row_ok].max().item()
    err_u = (out - out_unf).reshape(B, Q, G, T, P, -1).abs().amax((-1, -2))
    assert err_u.median().item() < 1e-4 and (err_u > 5e-4).float().mean().item() < 1e-3


@pytest.mark.parametrize("cfg", [syn.SMALL6, syn.F8, syn.F8_3CAM], ids=["small6", "f8", "f8_3cam"])
def test_sampling4d_variants_agree_and_follow_the_rigs_coverage(cfg):
    """The kernel's two variants (points without any tap set aside / plain) give the same bits on every rig, and the decoder
    picks between them from the rig's measured coverage, not from the number of cameras: a 6-camera rig with three failed
    cameras (zero projection matrices) is treated like the 3-camera rig."""
    from racformer_amd.fused import sampling4d_fused
    from racformer_amd.transformer import compact_variant, rig_coverage
    tr, sd, qb, qf, metas = _setup(cfg, 41, 42)
    smp = tr.decoder.decoder_layer.sampling
    feats = [f.to(DEV) for f in R.regroup_pyramid(syn.make_pyramid(cfg, 41), cfg.num_cams)]
    with torch.no_grad():
        lin = (smp.sampling_offset(qf.to(DEV)), smp.ray_points_offset(qf.to(DEV)), smp.scale_weights(qf.to(DEV)))
        outs = [sampling4d_fused(feats, qb.to(DEV), *lin, metas[0]["time_diff"], metas[0]["lidar2img"], cfg.num_frames, cfg.num_groups,
                                 cfg.num_points, cfg.img_depth_num, list(cfg.pc_range), cfg.d_region_list[1], cfg.image_hw[0],
                                 cfg.image_hw[1], compact=c) for c in (False, True, None)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    cov = metas[0]["_rac_coverage"]
    assert compact_variant(cov) == (cfg.num_cams == 3), cov
    if cfg.num_cams == 6:
        l2i = np.asarray(syn.make_img_metas(cfg)[0]["lidar2img"]).copy()
        l2i[[1, 3, 5]] = 0.0
        assert compact_variant(rig_coverage(l2i, 6, cfg.image_hw, cfg.pc_range)) is True


def test_sampling4d_four_samples_per_forward_level0_beyond_2gib():
    """B = 4 at f8 shapes: level 0 of the grouped pyramid holds 2.2 GB, more than a 31-bit tap offset reaches -- the descriptors
    are per slot since round 3, so the last sample's slots (beyond 2 GiB) are sampled like the first one's: batch element 3 against
    the ORACLE's sampling_4d on that element's own maps, and the whole batch against its own elements run one by one."""
    from dataclasses import replace
    from racformer_amd.fused import sampling4d_fused
    cfg = replace(syn.F8, batch=4)
    one = syn.F8
    tr, sd, qb1, qf1, metas1 = _setup(one, 51, 52)
    smp = tr.decoder.decoder_layer.sampling
    g = torch.Generator().manual_seed(9)
    qb = qb1.repeat(4, 1, 1)
    qb[..., 0:2] = (qb[..., 0:2] + 0.013 * torch.arange(4).view(4, 1, 1)) % 1.0
    qf = qf1.repeat(4, 1, 1) + 0.5 * torch.randn(4, one.num_query, 256, generator=g)
    S1 = one.num_frames * one.num_groups
    feats = [torch.randn(4 * S1, one.num_cams, h, w, 64, generator=g).to(DEV) for (h, w) in one.fpn_hw]
    assert feats[0].numel() * 4 > 2 ** 31
    td = metas1[0]["time_diff"].repeat(4, 1).contiguous()
    l2i = metas1[0]["lidar2img"].repeat(4, 1, 1, 1).contiguous()
    args = (one.num_frames, one.num_groups, one.num_points, one.img_depth_num, list(one.pc_range), one.d_region_list[1], one.image_hw[0],
            one.image_hw[1])
    with torch.no_grad():
        lin = [m(qf.to(DEV)) for m in (smp.sampling_offset, smp.ray_points_offset, smp.scale_weights)]
        out = sampling4d_fused(feats, qb.to(DEV), *lin, td, l2i, *args)
        for b in range(4):
            ob = sampling4d_fused([f[b * S1:(b + 1) * S1] for f in feats], qb[b:b + 1].to(DEV), *[x[b:b + 1].contiguous() for x in lin],
                                  td[b:b + 1], l2i[b:b + 1], *args)
            assert torch.equal(out[b:b + 1], ob), b
        # oracle on the last element
        b = 3
        tdc = R.time_diff_from_metas(syn.make_img_metas(one), 1, one.num_cams)
        l2c = torch.from_numpy(np.asarray([m["lidar2img"] for m in syn.make_img_metas(one)]).astype(np.float32))
        pts, sw = R.image_keypoints(sd, qb[b:b + 1], qf[b:b + 1], tdc, one.d_region_list[1], one)
        ref = R.sampling_4d(pts, [f[b * S1:(b + 1) * S1].cpu() for f in feats], sw, l2c, one.image_hw[0], one.image_hw[1])
    err = (out[b:b + 1].cpu() - ref).abs().reshape(one.num_query, -1).amax(-1)
    # (white-noise maps, O(1) per pixel: an ulp of a pixel coordinate is 1e-5 of a pixel and moves a sum of 16 taps by ~1e-4; a
    #  query one of whose points picked another camera differs by O(1))
    assert err.median().item() < 5e-4 and (err > 5e-3).float().mean().item() < 0.01, (err.median().item(), (err > 5e-3).float().mean().item())


@pytest.mark.parametrize("cfg", [syn.SMALL, syn.SMALL6, syn.F8, syn.F8_3CAM, _replace(syn.SMALL6, batch=2)],
                         ids=["small", "small6", "f8", "f8_3cam", "small6_b2"])
def test_mixing_sampled_equals_sampling_then_mixing(cfg):
    """rac_mixing_sampled_fwd (the AdaptiveMixing kernel gathering its own sampled features) against rac_sampling4d_fwd followed by
    rac_mixing_fwd on what that wrote: the out_proj operand image, the reported locations / camera choices and level weights are
    the same BITS -- free-running and with camera choices imposed (view_in), on 3- and 6-camera rigs (points without any tap) and
    with two samples per forward (the slot arithmetic of the second sample)."""
    from racformer_amd.fused import mixing_fused, mixing_sampled_fused, sampling4d_fused
    tr, sd, qb, qf, metas = _setup(cfg, 61, 62)
    smp = tr.decoder.decoder_layer.sampling
    feats = [f.to(DEV) for f in R.regroup_pyramid(syn.make_pyramid(cfg, 61), cfg.num_cams)]
    B, Q, T, G = cfg.batch, cfg.num_query, cfg.num_frames, cfg.num_groups
    P = cfg.num_points * cfg.img_depth_num
    g = torch.Generator().manual_seed(7)
    params = (torch.randn(B, Q, G * (64 * 64 + 128 * T * P), generator=g) * 0.2).to(DEV)
    args = (T, G, cfg.num_points, cfg.img_depth_num, list(cfg.pc_range), cfg.d_region_list[3], cfg.image_hw[0], cfg.image_hw[1])
    S = B * T * G
    forced = torch.randint(0, cfg.num_cams, (S, Q, P), generator=g, dtype=torch.uint8).to(DEV)
    with torch.no_grad():
        lin = (smp.sampling_offset(qf.to(DEV)), smp.ray_points_offset(qf.to(DEV)), smp.scale_weights(qf.to(DEV)))
        for view_in in (None, forced):
            x, loc, w = sampling4d_fused(feats, qb.to(DEV), *lin, metas[0]["time_diff"], metas[0]["lidar2img"], *args, debug=True,
                                         view_in=view_in)
            want = mixing_fused(x, params, T * P, G, 128, split=True, f16x3=True)
            got, loc2, w2 = mixing_sampled_fused(feats, qb.to(DEV), *lin, metas[0]["time_diff"], metas[0]["lidar2img"], *args, params,
                                                 debug=True, view_in=view_in)
            torch.cuda.synchronize()
            assert torch.equal(loc, loc2) and torch.equal(w, w2)
            assert got.shape == want.shape and torch.equal(got, want), (got.float() - want.float()).abs().max().item()
    assert float(want.float().abs().max()) > 0.1            # (not a comparison of zeros)


@pytest.mark.parametrize("cfg", [syn.SMALL, syn.F8])
def test_bev_sampling_fused(cfg):
    tr, sd, qb, qf, metas = _setup(cfg, 31, 32)
    layer = tr.decoder.decoder_layer
    d_region = cfg.d_region_list[1]
    lss = syn.make_bev(cfg, 31, 0)
    td = R.time_diff_from_metas(syn.make_img_metas(cfg), cfg.batch, cfg.num_cams)
    for name, temp in (("sampling_lss_bev", False), ("sampling_radar_bev", True)):
        mod = getattr(layer, name)
        with torch.no_grad():
            value, hw = mod.prepare_value(lss.to(DEV))
            got = mod.attend_prepared(qb.to(DEV), qf.to(DEV), value, hw, metas[0]["time_diff"], d_region)
            unf = plans.bev_attend_reference_ops(mod, qb.to(DEV), qf.to(DEV), value, hw, metas[0]["time_diff"], d_region)
            ref = R.bev_sampling(sd, name, qb, qf, lss, td, d_region, cfg, temp)
        torch.cuda.synchronize()
        assert (got.cpu() - ref).abs().max().item() < 2e-4, name
        assert (got - unf).abs().max().item() < 1e-4, name


@pytest.mark.parametrize("cfg", [syn.SMALL, syn.F8])
def test_bev_sampling_int16_block_value_streams(cfg):
    """Opt-in 16-bit block storage of the BEV value streams (csrc/quant.hip, rac_bev_sampling_multi_q16_fwd).  (1) The format: q * scale
    reproduces every value to 2^-15 of its (pixel, head) block's largest one, the scale is the power of two the format defines,
    zero blocks stay zero -- checked against the definition restated here in torch on the host.  (2) The kernel: both streams of a
    layer gathered from the int16 streams equal the ORACLE's deformable attention + frame fusion evaluated on the dequantised
    values (the scale folded into the tap weights changes nothing but the rounding order)."""
    from racformer_amd.fused import bev_sampling_multi_fused, box_prep, quantize_values_i16
    tr, sd, qb, qf, metas = _setup(cfg, 33, 34)
    layer = tr.decoder.decoder_layer
    d_region = cfg.d_region_list[2]
    T, heads, B, Q = cfg.num_frames, 4, cfg.batch, cfg.num_query
    H, W = cfg.bev_hw
    g = torch.Generator().manual_seed(5)
    td = R.time_diff_from_metas(syn.make_img_metas(cfg), cfg.batch, cfg.num_cams)
    streams, scales, deq, outs_ref = [], [], [], []
    for name in ("sampling_radar_bev", "sampling_lss_bev"):
        v = torch.randn(B * T, H * W, heads, 64, generator=g)
        v = v * torch.exp2(torch.randint(-20, 12, (B * T, H * W, heads, 1), generator=g).float())      # blocks over 32 octaves
        v[0, :7] = 0.0                                                                                 # some all-zero blocks
        q, sc = quantize_values_i16(v.to(DEV))
        torch.cuda.synchronize()
        m = v.abs().amax(-1)
        want_sc = torch.exp2(torch.frexp(m.clamp_min(1e-30))[1].float() - 1.0 - 14.0)      # m = mant * 2^e, mant in [0.5, 1): floor(log2 m) = e - 1
        assert torch.equal(sc.cpu()[m > 0], want_sc[m > 0]), name
        dq = q.cpu().float() * sc.cpu()[..., None]
        assert (dq - v).abs().max().item() == 0.0 or bool(((dq - v).abs() <= m[..., None] * 2.0 ** -15 * 1.0001).all()), name
        assert float(dq[0, :7].abs().max()) == 0.0 and int(q.abs().max()) <= 32767
        mod = getattr(layer, name)
        with torch.no_grad():
            lin = (mod.sampling_offset(qf.to(DEV)), mod.ray_points_offset(qf.to(DEV)), mod.scale_weights(qf.to(DEV)),
                   mod.attention.bev_queue_weight(qf.to(DEV)))
        streams.append((q,) + tuple(x.contiguous() for x in lin))
        scales.append(sc)
        # the oracle on the dequantised stream: keypoints -> per-frame deformable attention -> frame softmax fusion
        loc, sw = R.bev_keypoints(sd, name, qb, qf, td, d_region, cfg)
        P = loc.shape[-2]
        loc7 = loc.view(B, Q, heads, T, 1, P, 2).permute(3, 0, 1, 2, 4, 5, 6).reshape(B * T, Q, heads, 1, P, 2)
        aw = sw.view(B, Q, heads, T, 1, P).permute(3, 0, 1, 2, 4, 5).reshape(B * T, Q, heads, 1, P)
        o = R.msda(dq.contiguous(), [[H, W]], [0], loc7.contiguous(), aw.contiguous()).permute(1, 2, 0).reshape(Q, 256, B, T)
        qw = R._lin(sd, name + ".attention.bev_queue_weight", qf).permute(1, 0, 2).reshape(Q, 1, B, T)
        outs_ref.append(torch.sum(o * torch.softmax(qw, dim=-1), dim=-1).permute(2, 0, 1))
    out = torch.empty(2, B, Q, 256, device=DEV)
    qbd = qb.to(DEV).contiguous()
    rb = layer.sampling_radar_bev
    with torch.no_grad():
        bev_sampling_multi_fused(streams, (H, W), qbd, metas[0]["time_diff"], rb.num_frames, rb.num_heads, rb.num_points, rb.depth_num,
                                 rb.pc_range, d_region, box_prep(qbd, list(cfg.pc_range)), out, value_scales=scales)
    torch.cuda.synchronize()
    for i in range(2):
        scale_ = float(outs_ref[i].abs().max())
        # (white-noise blocks over 32 octaves: an ulp of a pixel coordinate moves a 640-tap sum by ~3e-5 of its magnitude -- the two
        #  sides compute the coordinates with different libm; the quantisation itself does not enter: both sides read q * scale)
        assert (out[i].cpu() - outs_ref[i]).abs().max().item() < 1e-4 * max(scale_, 1.0), (i, (out[i].cpu() - outs_ref[i]).abs().max().item(), scale_)


@pytest.mark.parametrize("cfg", [syn.SMALL, syn.F8])
def test_sasa_fused(cfg):
    tr, sd, qb, qf, metas = _setup(cfg, 41, 42)
    sa = tr.decoder.decoder_layer.self_attn
    with torch.no_grad():
        got = sa(qb.to(DEV), qf.to(DEV), None)
        unf = sa.forward_unfused(qb.to(DEV), qf.to(DEV), None)
        ref = R.sasa(sd, qb, qf, cfg.pc_range)
    torch.cuda.synchronize()
    assert (got.cpu() - ref).abs().max().item() < 2e-5
    assert (got - unf).abs().max().item() < 2e-5


@pytest.mark.parametrize("P", [96, 36, 24, 7])
def test_mixing_fused(P):
    """rac_mixing_fwd (exact-fp32 MFMA) vs the torch formulation and the CPU oracle; P=96 is f8,
    36/24 the reduced configs (padded tiles), 7 an odd size (unaligned S rows)."""
    from racformer_amd.transformer import AdaptiveMixing
    torch.manual_seed(P)
    mix = AdaptiveMixing(in_dim=256, in_points=P, n_groups=4, out_points=128).eval()
    Q = 37
    x = torch.randn(1, Q, 4, P, 64)
    q = torch.randn(1, Q, 256)
    sd = {"mixing." + k: v.detach().clone() for k, v in mix.state_dict().items()}
    with torch.no_grad():
        ref = R.adaptive_mixing(sd, x, q)
        mg = mix.to(DEV)
        got = mg(x.to(DEV), q.to(DEV), mg.split_out_proj())
        unf = mg(x.to(DEV), q.to(DEV))
        # the two arithmetic modes of the kernel itself on identical parameters (large-magnitude features included:
        # the x operand of RAC_MIX_F16X3 is split into bf16 terms, so it keeps the fp32 exponent range)
        from racformer_amd.fused import mixing_fused
        xs = x.to(DEV) * torch.tensor([1.0, 3.0e5, 1.0e-6, 40.0], device=DEV).view(1, 1, 4, 1, 1)
        params = mg.parameter_generator(q.to(DEV))
        z32 = mixing_fused(xs, params, P, 4, 128)
        z16 = mixing_fused(xs, params, P, 4, 128, f16x3=True)
        zs = mixing_fused(xs, params * 4.0, P, 4, 128, f16x3=True, param_scale=0.25)
    torch.cuda.synchronize()
    assert (got - unf).abs().max().item() < 2e-4
    assert (got.cpu() - ref).abs().max().item() < 2e-4
    assert torch.isfinite(z16).all() and (z16 - z32).abs().max().item() < 5e-5, (z16 - z32).abs().max().item()
    assert torch.equal(zs, z16)          # power-of-two parameter scale is exact


def test_refine_and_add_ln_kernels():
    from racformer_amd.fused import add_ln, refine_fused
    torch.manual_seed(3)
    B, Q, T = 2, 37, 3
    prop = torch.rand(B, Q, 10)
    prop[0, 0, 1:3] = torch.tensor([0.0, 1.0])       # inverse_sigmoid clamps
    delta = torch.randn(B, Q, 10)
    td = torch.tensor([[0.0, 0.5, 1.0], [0.0, 1.0, 1.5]])
    td_safe = td.clone()
    td_safe[td_safe < 1e-5] = 1.0
    ref = R.refine_bbox(prop, delta, 150)
    ref = torch.cat([ref[..., :8], ref[..., 8:] / td_safe[:, 1:2, None]], dim=-1)
    pred, xy = refine_fused(prop.to(DEV), delta.to(DEV), td_safe.to(DEV), 150)
    assert (pred.cpu() - ref).abs().max().item() < 1e-5
    assert (xy.cpu() - R.theta_d2xy(ref)).abs().max().item() < 1e-5
    for dim, S, relu, scale in ((256, 1, False, 1.0), (256, 32, True, 0.25), (512, 3, False, 1.0), (1024, 1, True, 2.0)):
        ln = torch.nn.LayerNorm(dim)
        torch.nn.init.normal_(ln.weight)
        torch.nn.init.normal_(ln.bias)
        a, res, bias = torch.randn(S, 5, 41, dim), torch.randn(5, 41, dim), torch.randn(dim)
        want = ln(scale * a.sum(0) + res + bias)
        want = torch.relu(want) if relu else want
        lg = ln.to(DEV)
        got = add_ln(a.to(DEV) if S > 1 else a[0].to(DEV), lg, residual=res.to(DEV), bias=bias.to(DEV), relu=relu,
                     num_partials=S, a_scale=scale)
        assert (got.cpu() - want.detach()).abs().max().item() < 2e-5, (dim, S)
        ln.cpu()


@pytest.mark.parametrize("Q", [1, 15, 16, 17, 64, 900, 1024, 1100])
def test_sasa_kernels_all_sizes(Q):
    """MFMA kernel (Q <= 1024) and the LDS-tiled VALU kernel (Q > 1024) against a float64 reference, with and
    without the box table; ragged query counts around the 16-row tile."""
    from racformer_amd.fused import box_prep, sasa_fused
    rng = np.random.default_rng(Q)
    H, d = 8, 32
    qkv = torch.from_numpy(rng.standard_normal((1, Q, 3 * H * d), dtype=np.float32))
    tau = torch.from_numpy(rng.random((1, Q, H), dtype=np.float32) * 2)
    qb = torch.from_numpy(rng.random((1, Q, 10), dtype=np.float32))
    q, k, v = (t_.view(1, Q, H, d).permute(0, 2, 1, 3).double() for t_ in qkv.split(H * d, dim=-1))
    c = R.decode_bbox(R.theta_d2xy(qb), syn.PC_RANGE)[..., :2].double()
    dist = -(c[:, :, None] - c[:, None]).norm(dim=-1)
    logits = (q / np.sqrt(d)) @ k.transpose(-1, -2) + dist[:, None] * tau.double().permute(0, 2, 1)[..., None]
    want = (torch.softmax(logits, -1) @ v).permute(0, 2, 1, 3).reshape(1, Q, H * d)
    g = [x.to(DEV) for x in (qkv, tau, qb)]
    for table in (None, box_prep(g[2], syn.PC_RANGE)):
        got = sasa_fused(g[0], g[1], g[2], H, syn.PC_RANGE, box_table=table)
        assert (got.cpu().double() - want).abs().max().item() < 5e-5, (Q, table is None)   # fp32 vs float64, logits O(10)


def test_split_precision_operands_and_gemms():
    """The f16 hi/lo images written by rac_add_ln_fwd / rac_mixing_fwd reproduce the fp32 values to 2^-21, and the
    3-product GEMMs built on them (library GEMM over K-concatenated operands for the generator, hand-written rac_outproj_fwd
    for out_proj) match a float64 GEMM as closely as the fp32 GEMM does."""
    from racformer_amd.fused import SPLIT_ACT_SCALE, SPLIT_BIAS_PAD, add_ln, mixing_fused, outproj_fused, split_weight_f16
    from racformer_amd.transformer import AdaptiveMixing
    torch.manual_seed(5)
    ln = torch.nn.LayerNorm(256).to(DEV)
    a, res = torch.randn(1, 900, 256, device=DEV), torch.randn(1, 900, 256, device=DEV)
    out, img = add_ln(a, ln, residual=res, split=True)
    assert img.dtype == torch.float16 and tuple(img.shape) == (900, 768 + SPLIT_BIAS_PAD)
    hi, hi2, lo, pad = img.float().split([256, 256, 256, SPLIT_BIAS_PAD], dim=1)
    assert torch.equal(pad, torch.tensor([SPLIT_ACT_SCALE] * 2 + [0.0] * (SPLIT_BIAS_PAD - 2), device=DEV).expand(900, SPLIT_BIAS_PAD))
    assert torch.equal(hi, hi2)
    rec = (hi.double() + lo.double()) / SPLIT_ACT_SCALE
    assert (rec - out.view(900, 256).double()).abs().max().item() <= 2.0 ** -21 * out.abs().max().item()
    # generator-shaped GEMM: split path vs fp32 rocBLAS vs float64
    lin = torch.nn.Linear(256, 4096).to(DEV)
    w3, alpha = split_weight_f16(lin.weight, lin.bias)
    got = torch.mm(img, w3.t(), out_dtype=torch.float32) * alpha
    want = out.view(900, 256).double() @ lin.weight.double().t() + lin.bias.double()
    e_split = (got.double() - want).abs().max().item()
    e_fp32 = (lin(out.view(900, 256)).double() - want).abs().max().item()
    assert e_split < 4 * e_fp32 + 1e-6, (e_split, e_fp32)
    # mixing kernel: split image of the output vs its fp32 output; then out_proj both ways
    mix = AdaptiveMixing(in_dim=256, in_points=96, n_groups=4, out_points=128).eval().to(DEV)
    x, q = torch.randn(1, 64, 4, 96, 64, device=DEV), torch.randn(1, 64, 256, device=DEV)
    with torch.no_grad():
        params = mix.parameter_generator(q)
        z = mixing_fused(x, params, 96, 4, 128)
        z16 = mixing_fused(x, params, 96, 4, 128, split=True)
        assert tuple(z16.shape) == (64, 4 * 256, 64)                      # line image: [hi 32 | lo 32] per 32 values of K
        zh, zl = z16.float().view(64, 1024, 2, 32).unbind(2)
        rec = ((zh.double() + zl.double()) / SPLIT_ACT_SCALE).reshape(64, -1)
        assert (rec - z.view(64, -1).double()).abs().max().item() <= 2.0 ** -21 * z.abs().max().item()
        packs = mix.split_packs()
        wh, wl = packs["out_w"].float().view(256, 1024, 2, 32).unbind(2)   # the packed weight image holds W * 2^s as hi + lo
        wrec = (wh.double() + wl.double()).reshape(256, -1) * (packs["out_alpha"] * SPLIT_ACT_SCALE)
        assert (wrec - mix.out_proj.weight.double()).abs().max().item() <= 2.0 ** -21 * mix.out_proj.weight.abs().max().item()
        part = outproj_fused(z16, packs["out_w"], packs["out_slices"]) * packs["out_alpha"]
        assert tuple(part.shape) == (32, 64, 256)
        want = z.view(64, -1).double() @ mix.out_proj.weight.double().t()
        e_split = (part.sum(0).double() - want).abs().max().item()
        e_fp32 = ((z.view(64, -1) @ mix.out_proj.weight.t()).double() - want).abs().max().item()
        assert e_split < 4 * e_fp32 + 1e-6, (e_split, e_fp32)
    assert mix.split_packs(act_bound=1e5) == {}     # operands outside the f16 range: the caller keeps the fp32 GEMMs


@pytest.mark.parametrize("M,N,K,S", [(900, 256, 32768, 32), (900, 256, 32768, 16), (131, 256, 4096, 2), (7, 100, 1024, 4),
                                     (128, 128, 64, 1), (128, 128, 32, 1), (200, 130, 96, 1), (300, 700, 64, 2)])
def test_outproj_kernel_vs_float64(M, N, K, S):
    """rac_outproj_fwd (hand-written split-K GEMM, LDS-DMA staging, source-side swizzle) against a float64 GEMM of the values
    its operand images hold: asymmetric integer-free random operands, ragged M / N (tile edges), every slice checked."""
    from racformer_amd.fused import outproj_fused
    g = torch.Generator().manual_seed(M + K)
    z = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)

    def image(t):       # [R, K] f32 -> line image [R, K/32, hi 32 | lo 32] f16 and the f64 values it represents
        hi = t.to(torch.float16)
        lo = (t - hi.float()).to(torch.float16)
        img = torch.stack([hi.view(-1, K // 32, 32), lo.view(-1, K // 32, 32)], dim=2).reshape(-1, K // 32, 64).contiguous()
        return img, hi.double() + lo.double()

    zi, zv = image(z * 16.0)
    wi, wv = image(w * 1024.0)
    part = outproj_fused(zi, wi, S)
    assert tuple(part.shape) == (S, M, N)
    ks = K // S
    for s_ in range(S):
        want = zv[:, s_ * ks:(s_ + 1) * ks] @ wv[:, s_ * ks:(s_ + 1) * ks].t()
        err = (part[s_].double() - want).abs().max().item()
        # dropped lo*lo term (2^-22 relative to the hi*hi terms) + fp32 accumulation of ks products
        assert err <= 3e-6 * float(want.abs().max()) + 1e-6 * float((zv.abs().max() * wv.abs().max())) * ks ** 0.5, (s_, err)


@pytest.mark.parametrize("M,N,K", [(900, 4096, 256), (900, 65536, 256), (37, 520, 64), (129, 256, 32), (1, 4, 96),
                                   (900, 2189, 256), (70, 777, 256), (33, 30, 256), (4000, 2189, 256), (200, 33000, 256)])
def test_generator_kernel_vs_float64(M, N, K):
    """rac_generator_fwd (the same kernel, persistent over the row tiles of a feature tile, affine epilogue) against float64:
    ragged rows (tiles of unequal height, skipped MFMA tiles), ragged features (N % 256 != 0, N % 4 != 0 with a padded row
    stride: the 2189 outputs of the sampling Linears, rows cut into chunks), bias, alpha; more than three row stages per
    workgroup together with a partial last feature block (M = 4000 / N = 2189, M = 200 / N = 33000: the waves that store
    nothing there must not run ahead of the X stages -- their counted waits cover the pieces alone); and the X line image written by
    rac_rowgemm_fwd's prologue (K = 256, N % 4 == 0) or rac_add_ln_fwd (N % 4 != 0)."""
    from racformer_amd.fused import SPLIT_ACT_SCALE, generator_fused, pack_gemm_split_weight, row_gemm, row_seg, rowgemm_launch
    g = torch.Generator().manual_seed(M + N)
    lin = torch.nn.Linear(K, N)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) * 0.05)
        lin.bias.copy_(torch.randn(N, generator=g))
    lin = lin.to(DEV)
    x = torch.randn(M, K, generator=g).to(DEV)
    w_img, alpha = pack_gemm_split_weight(lin.weight)
    if K == 256:
        # the image as the decoder produces it: finished rows of a rowgemm prologue (here: plain copy of x), written beside a
        # throw-away 16-column GEMM
        img = torch.empty(M, 512, device=DEV, dtype=torch.float16)
        if N % 4 == 0:
            dummy_w, dummy_out = torch.zeros(16, 256, device=DEV), torch.empty(M, 16, device=DEV)
            rowgemm_launch([row_gemm([row_seg(x, split_out=img, split_lines=True)], dummy_w, None, dummy_out)], M)
        else:
            # the image as rac_add_ln_fwd writes it: x := LN(a) with the layer's own affine, the GEMM then runs on that x
            from racformer_amd.fused import add_ln
            ln = torch.nn.LayerNorm(256).to(DEV)
            with torch.no_grad():
                ln.weight.copy_(torch.rand(256, generator=g) + 0.5)
                ln.bias.copy_(torch.randn(256, generator=g) * 0.1)
            x, _ = add_ln(x, ln, split=True, split_lines=True, split_out=img)
    else:
        xs = x * SPLIT_ACT_SCALE
        hi = xs.to(torch.float16)
        lo = (xs - hi.float()).to(torch.float16)
        img = torch.stack([hi.view(M, K // 32, 32), lo.view(M, K // 32, 32)], dim=2).reshape(M, K // 32 * 64).contiguous()
    got = generator_fused(img, w_img, lin.bias, alpha, ld_out=(N + 3) // 4 * 4)[:, :N]
    want = x.double() @ lin.weight.double().t() + lin.bias.double()
    e_split = (got.double() - want).abs().max().item()
    e_fp32 = (lin(x).double() - want).abs().max().item()
    assert tuple(got.shape) == (M, N) and e_split < 4 * e_fp32 + 1e-6, (e_split, e_fp32)


@pytest.mark.parametrize("shape", [(8, 128, 128, 256, 64), (3, 16, 16, 256, 64), (2, 32, 64, 32, 32)])
def test_conv3x3_fused_matches_fp32_conv(shape):
    """rac_absmax/conv_pack/conv3x3 (implicit GEMM, f16 MFMA on hi/lo-split operands) vs a float64 convolution on
    the CPU (small) or vs MIOpen's fp32 convolution (full size), incl. huge / tiny magnitudes (device-side scale)."""
    from racformer_amd.fused import conv3x3_fused, pack_conv3x3_weight
    N, H, W, C1, C2 = shape
    torch.manual_seed(N)
    conv = torch.nn.Conv2d(C1 + C2, 256, 3, padding=1)
    for mag in (1.0, 3.0e7, 1.0e-9):
        x1, x2 = torch.randn(N, C1, H, W) * mag, torch.randn(N, C2, H, W) * mag * 0.5
        ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV))
        got = conv3x3_fused([x1.to(DEV), x2.to(DEV)], ws, alpha, conv.bias.detach().to(DEV))
        assert tuple(got.shape) == (N, H, W, 256)
        with torch.no_grad():
            if N * H * W <= 4096:
                want = torch.nn.functional.conv2d(torch.cat([x1, x2], 1).double(), conv.weight.double(), conv.bias.double(),
                                                  padding=1).permute(0, 2, 3, 1)
                tol = 4e-6
            else:
                want = conv.to(DEV)(torch.cat([x1, x2], 1).to(DEV)).permute(0, 2, 3, 1).double().cpu()
                conv.cpu()
                tol = 2e-5      # two fp32-grade results against each other
        scale = want.abs().max().item()
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= tol * scale, (shape, mag, err, scale)


def test_temporal_encoder_fused_pieces():
    """rac_gru_gate_fwd / rac_upsample2x_fwd against the torch primitives they replace, and the whole RadarBEVTemporalEncoder
    (fused convolution + fused pieces, channel-last output) against the oracle's restatement of the reference module
    (oracle/restate.py::temporal_encoder, itself pinned by the decoder fixtures) on CPU."""
    from racformer_amd.fused import gru_gate_fused, pack_conv3x3_weight, upsample2x_fused
    from racformer_amd.transformer import ConvGRUCell, RadarBEVTemporalEncoder
    torch.manual_seed(9)
    x = torch.randn(3, 8, 20, 12, device=DEV)
    want = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    assert (upsample2x_fused(x) - want).abs().max().item() < 1e-5      # 4-term blends of O(1) values, fma contraction differs
    cell = ConvGRUCell(16, 16, 3).to(DEV)
    xs, h0 = torch.randn(2, 16, 8, 8, device=DEV), torch.randn(2, 3, 16, 8, 8, device=DEV)
    with torch.no_grad():
        want = cell(xs, h0[:, 1])
        out = torch.zeros(2, 3, 16, 8, 8, device=DEV)
        gru_gate_fused(cell.gates(xs, h0[:, 1]), h0[:, 1], out[:, 2])
    assert (out[:, 2] - want).abs().max().item() < 2e-6 and float(out[:, :2].abs().max()) == 0.0
    def oracle_te(mod, x):
        sd = {"te." + k: v.detach().clone() for k, v in mod.state_dict().items()}
        return R.temporal_encoder(sd, "te", x)                            # models/racformer_transformer.py:645-656 restated

    enc = RadarBEVTemporalEncoder(256, 64, 8).eval()
    bev = torch.randn(1, 8, 256, 16, 16) * 0.5
    with torch.no_grad():
        ref = oracle_te(enc, bev)
        eg = enc.to(DEV)
        ws, alpha = pack_conv3x3_weight(eg.temporal_fusion.weight)
        got = eg.forward_channel_last(bev.to(DEV), dict(ws=ws, alpha=alpha, bound=eg.hidden_bound(), **eg.downsample_pack(16, 16)))
        # round 5: with the per-pixel maps in the pack the ConvGRU branch runs on rac_conv_direct_fwd and the fusion convolution
        # skips the constant hidden half of the frames past the live ones (rac_conv3x3_temporal_fwd)
        pack_own = dict(ws=ws, alpha=alpha, bound=eg.hidden_bound(), **eg.temporal_bias_maps(16, 16), **eg.downsample_pack(16, 16))
        assert "gx_ws" in pack_own and "pixel_bias_dead" in pack_own
        got_own = eg.forward_channel_last(bev.to(DEV), pack_own)
    assert (got.permute(0, 3, 1, 2).cpu() - ref[0]).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    assert (got_own.permute(0, 3, 1, 2).cpu() - ref[0]).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    # 32 x 32 maps: the stride-2 kernel and the launch-lean ConvGRU (composed matching layer, bias map) are on the path
    enc2 = RadarBEVTemporalEncoder(256, 64, 8).eval()
    for mod in (enc2.convGRU.convGRUCell.gates_conv, enc2.convGRU.convGRUCell.matching_layer, enc2.downsample):
        torch.nn.init.normal_(mod.bias, std=0.2)
    bev2 = torch.randn(1, 8, 256, 32, 32) * 0.5
    with torch.no_grad():
        ref2 = oracle_te(enc2, bev2)
        eg2 = enc2.to(DEV)
        ws2, alpha2 = pack_conv3x3_weight(eg2.temporal_fusion.weight)
        pack2 = dict(ws=ws2, alpha=alpha2, bound=eg2.hidden_bound(), **eg2.downsample_pack(32, 32))
        assert "gru_w" in pack2 and "down_ws" in pack2
        got2 = eg2.forward_channel_last(bev2.to(DEV), pack2)
        got2_own = eg2.forward_channel_last(bev2.to(DEV), dict(pack2, **eg2.temporal_bias_maps(32, 32)))
    assert (got2.permute(0, 3, 1, 2).cpu() - ref2[0]).abs().max().item() < 2e-5 * ref2.abs().max().item() + 1e-5
    assert (got2_own.permute(0, 3, 1, 2).cpu() - ref2[0]).abs().max().item() < 2e-5 * ref2.abs().max().item() + 1e-5


@pytest.mark.parametrize("rows", [900, 37])
def test_rowgemm_kernel(rows):
    """rac_rowgemm_fwd against torch: plain / LayerNorm / split-K-sum prologues, ReLU placement, K = 256..768,
    N = 10 .. 2189, batched launches, strided sources, side outputs (x_out, f16 split image)."""
    from racformer_amd.fused import SPLIT_ACT_SCALE, SPLIT_BIAS_PAD, row_gemm, row_seg, rowgemm_launch
    torch.manual_seed(rows)
    g = lambda *s: torch.randn(*s, device=DEV)   # noqa: E731
    ln = [torch.nn.LayerNorm(256).to(DEV) for _ in range(3)]
    for m in ln:
        torch.nn.init.normal_(m.weight)
        torch.nn.init.normal_(m.bias)
    # 1) three LN segments, one of them a scaled split-K sum with bias and residual; K = 768, N = 256
    parts, res, b0 = g(16, rows, 256), g(rows, 256), g(256)
    p1, p2 = g(rows, 256), g(rows, 512)
    W, b, out = g(256, 768) * 0.05, g(256), torch.empty(rows, 256, device=DEV)
    xo = torch.empty(rows, 256, device=DEV)
    rowgemm_launch([row_gemm([row_seg(parts, num_partials=16, a_scale=0.25, bias0=b0, residual=res, norm=ln[0], x_out=xo),
                              row_seg(p1, residual=res, norm=ln[1], relu=True), row_seg(p2[:, 256:], norm=ln[2], post=res)],
                             W, b, out)], rows)
    s0 = ln[0](0.25 * parts.sum(0) + b0 + res)
    s1 = torch.relu(ln[1](p1 + res))
    s2 = ln[2](p2[:, 256:]) + res
    want = torch.cat([s0, s1, s2], 1).double() @ W.double().t() + b.double()
    assert (xo - s0).abs().max().item() < 2e-5
    assert (out.double() - want).abs().max().item() < 2e-4 * want.abs().max().item()
    # 2) batched: N = 10 with LN+ReLU prologue, N = 2189 plain with ReLU from column 100, and the split image
    W1, b1, o1 = g(10, 256), g(10), torch.empty(1, rows, 10, device=DEV)
    W2, b2, o2 = g(2189, 256) * 0.1, g(2189), torch.empty(rows, 2189, device=DEV)
    img = torch.empty(rows, 768 + SPLIT_BIAS_PAD, device=DEV, dtype=torch.float16)
    rowgemm_launch([row_gemm([row_seg(p1, norm=ln[0], relu=True)], W1, b1, o1),
                    row_gemm([row_seg(p2[:, :256], split_out=img)], W2, b2, o2, relu_from=100)], rows)
    w1 = torch.relu(ln[0](p1)).double() @ W1.double().t() + b1.double()
    w2 = p2[:, :256].double() @ W2.double().t() + b2.double()
    w2[:, 100:] = torch.relu(w2[:, 100:])
    assert (o1[0].double() - w1).abs().max().item() < 1e-4 * w1.abs().max().item()
    assert (o2.double() - w2).abs().max().item() < 1e-4 * w2.abs().max().item()
    hi, hi2, lo, pad = img.float().split([256, 256, 256, SPLIT_BIAS_PAD], dim=1)
    assert torch.equal(hi, hi2) and float(pad[:, 2:].abs().max()) == 0.0 and torch.all(pad[:, :2] == SPLIT_ACT_SCALE)
    assert ((hi.double() + lo.double()) / SPLIT_ACT_SCALE - p2[:, :256].double()).abs().max().item() < 2.0 ** -20 * 6
    # 3) K = 512 as two strided segments
    W3, o3 = g(256, 512) * 0.05, torch.empty(rows, 256, device=DEV)
    rowgemm_launch([row_gemm([row_seg(p2[:, :256]), row_seg(p2[:, 256:])], W3, None, o3, relu_from=0)], rows)
    w3 = torch.relu(p2.double() @ W3.double().t())
    assert (o3.double() - w3).abs().max().item() < 1e-4 * w3.abs().max().item()


def test_composed_radar_value_stream():
    """value_proj composed into the temporal-fusion convolution (BEVSampling.composed_value_pack, per-pixel bias in the
    convolution's epilogue) against the sequential evaluation temporal_encoder -> + pos -> value_proj."""
    from racformer_amd.transformer import BEVSampling
    torch.manual_seed(13)
    bs = BEVSampling(256, num_frames=8, num_points=4, num_heads=4, num_levels=1, pc_range=list(syn.PC_RANGE), depth_num=5,
                     spatial_shapes=(16, 16), temp_radar=True).eval()
    torch.nn.init.normal_(bs.temporal_encoder.temporal_fusion.bias, std=0.1)
    torch.nn.init.normal_(bs.attention.value_proj.bias, std=0.1)
    bev = torch.randn(1, 8, 256, 16, 16) * 0.7
    with torch.no_grad():
        want, _ = bs.prepare_value(bev)                       # CPU: MIOpen-free sequential reference
        bg = bs.to(DEV)
        got, hw = bg.prepare_value(bev.to(DEV), bg.composed_value_pack(16, 16))
    assert hw == (16, 16) and tuple(got.shape) == tuple(want.shape)
    assert (got.cpu() - want).abs().max().item() < 2e-5 * want.abs().max().item() + 1e-5


@pytest.mark.parametrize("shape", [(2, 32, 32, 320), (8, 128, 128, 320)])
def test_conv3x3_stride2_kernel(shape):
    """rac_conv3x3s2_fwd (the temporal encoder's downsample convolution on the image's first 256 channels) vs torch."""
    from racformer_amd.fused import ConvImage, pack_conv3x3_weight
    N, H, W, C = shape
    torch.manual_seed(H)
    conv = torch.nn.Conv2d(256, 64, 3, stride=2, padding=1)
    x, rest = torch.randn(N, 256, H, W) * 2.0, torch.randn(N, C - 256, H, W)
    img = ConvImage(N, H, W, C, torch.device(DEV))
    xg, rg = x.to(DEV), rest.to(DEV)
    img.begin([xg, rg]).pack(xg, 0).pack(rg, 256)
    ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV), cout=64)
    got = img.conv_s2(ws, alpha, conv.bias.detach().to(DEV), 256)
    assert tuple(got.shape) == (N, 64, H // 2, W // 2)
    with torch.no_grad():
        if N * H * W <= 4096:
            want, tol = conv.double()(x.double()), 4e-6
        else:
            want, tol = conv.to(DEV)(xg).double().cpu(), 2e-5
    assert (got.double().cpu() - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.parametrize("F_,H,W", [(8, 128, 128), (3, 16, 16), (1, 8, 4), (2, 96, 100), (5, 24, 40)])  # 300 blocks: some workgroups take two; 30: fewer than CUs
def test_value_proj_kernel_vs_float64(F_, H, W):
    """rac_value_proj_fwd (transpose + per-pixel hi / lo split + split-precision GEMM + additive term in one kernel) against
    float64, with channels and pixels of very different magnitude (the activation scale is chosen per pixel), an all-zero
    pixel, and against the fp32 library formulation it replaces."""
    from racformer_amd.fused import SPLIT_ACT_SCALE, pack_gemm_split_weight, value_proj_fused
    g = torch.Generator().manual_seed(F_ * H + W)
    lin = torch.nn.Linear(256, 256)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(256, 256, generator=g) * 0.06)
        lin.bias.copy_(torch.randn(256, generator=g))
    lin = lin.to(DEV)
    x = torch.randn(F_, 256, H, W, generator=g)
    x *= torch.exp(torch.randn(F_, 1, H, W, generator=g) * 3.0)          # pixels spanning orders of magnitude
    x[:, ::7] *= 1e-3                                                    # weak channels next to strong ones
    x[0, :, 0, 0] = 0.0
    x = x.to(DEV)
    add = torch.randn(H * W, 256, generator=g).to(DEV)
    w_img, alpha = pack_gemm_split_weight(lin.weight)
    got = value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, add=add)
    a = x.reshape(F_, 256, H * W).transpose(1, 2)
    want = a.double() @ lin.weight.double().t() + add.double()
    ref32 = torch.baddbmm(add.unsqueeze(0).expand(F_, H * W, 256), a, lin.weight.t().unsqueeze(0).expand(F_, 256, 256))
    # error relative to each pixel's own magnitude (rows differ by orders of magnitude)
    scale = a.abs().amax(-1, keepdim=True).double().clamp_min(1e-30)
    e_split = ((got.double() - want).abs() / scale).max().item()
    e_fp32 = ((ref32.double() - want).abs() / scale).max().item()
    assert tuple(got.shape) == (F_, H * W, 256) and e_split < 4 * e_fp32 + 1e-7, (e_split, e_fp32)
    assert torch.equal(got[0, 0], add[0])                                # the all-zero pixel: exactly the additive term
    got_b = value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, bias=lin.bias)
    assert ((got_b.double() - (want - add.double() + lin.bias.double())).abs() / scale).max().item() < 4 * e_fp32 + 1e-6


@pytest.mark.parametrize("N,H,W,pixel_bias", [(8, 128, 128, True), (3, 16, 16, False), (2, 32, 64, True)])
def test_conv3x3_q16_epilogue_is_the_quantiser_of_the_fp32_output(N, H, W, pixel_bias):
    """rac_conv3x3_q16_fwd (the temporal-fusion convolution writing the int16 block storage of the radar value stream in its
    epilogue) against rac_quant_i16_fwd applied to rac_conv3x3_fwd's fp32 output on the same image: mantissas and scales bit for
    bit, incl. blocks spanning many octaves (per-pixel bias map of mixed magnitude) and an all-zero block."""
    from racformer_amd.fused import ConvImage, pack_conv3x3_weight, quantize_values_i16
    g = torch.Generator().manual_seed(N * H + W)
    conv = torch.nn.Conv2d(96, 256, 3, padding=1)
    with torch.no_grad():
        conv.weight[64:128] *= 1e-4                # a head (64 output channels) of tiny values next to ordinary ones
        conv.weight[128:192] = 0.0                 # and one that is exactly the bias
        conv.bias[128:192] = 0.0
    x = (torch.randn(N, 96, H, W, generator=g) * torch.exp(torch.randn(N, 1, H, W, generator=g) * 2.0)).to(DEV)
    ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV))
    pb = None
    if pixel_bias:
        pb = torch.randn(H * W, 256, generator=g) * torch.exp(torch.randn(H * W, 1, generator=g) * 3.0)
        pb[:, 128:192] = 0.0
        pb = pb.to(DEV).contiguous()
    bias = None if pixel_bias else conv.bias.detach().to(DEV)
    img = ConvImage(N, H, W, 96, x.device).begin([x]).pack(x, 0)
    f32 = img.conv(ws, alpha, bias, pb)
    q, sc = img.conv(ws, alpha, bias, pb, q16=True)
    wq, wsc = quantize_values_i16(f32.view(N, H * W, 4, 64))
    torch.cuda.synchronize()
    assert tuple(q.shape) == (N, H * W, 4, 64) and q.dtype == torch.int16 and tuple(sc.shape) == (N, H * W, 4)
    assert torch.equal(sc, wsc) and torch.equal(q, wq)
    assert int(q[:, :, 2].abs().max()) == 0 and int(q[:, :, 0].abs().max()) >= 16384      # the zero head; full-range mantissas elsewhere


@pytest.mark.parametrize("F_,H,W", [(8, 128, 128), (3, 16, 16), (1, 8, 4), (2, 96, 100), (5, 24, 40)])  # 300 blocks: some workgroups take two; 30: fewer than CUs
def test_value_proj_q16_epilogue_is_the_quantiser_of_the_fp32_output(F_, H, W):
    """rac_value_proj_q16_fwd against rac_quant_i16_fwd(rac_value_proj_fwd(.)): bit for bit (the block maximum of a head is
    combined across the two waves that hold its 64 features)."""
    from racformer_amd.fused import SPLIT_ACT_SCALE, pack_gemm_split_weight, quantize_values_i16, value_proj_fused
    g = torch.Generator().manual_seed(F_ * H + W + 1)
    w = torch.randn(256, 256, generator=g) * 0.06
    w[64:128] *= 1e-5
    w[192:256] = 0.0
    x = torch.randn(F_, 256, H, W, generator=g) * torch.exp(torch.randn(F_, 1, H, W, generator=g) * 3.0)
    x[0, :, 0, 0] = 0.0
    add = torch.randn(H * W, 256, generator=g)
    add[:, 192:256] = 0.0
    add[0] = 0.0
    x, add = x.to(DEV), add.to(DEV).contiguous()
    w_img, alpha = pack_gemm_split_weight(w.to(DEV))
    f32 = value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, add=add)
    q, sc = value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, add=add, q16=True)
    wq, wsc = quantize_values_i16(f32.view(F_, H * W, 4, 64))
    torch.cuda.synchronize()
    assert torch.equal(sc, wsc) and torch.equal(q, wq)
    assert int(q[:, :, 3].abs().max()) == 0 and int(q[0, 0].abs().max()) == 0           # the zero head, the all-zero pixel
    b = torch.randn(256, generator=g).to(DEV)
    q2, sc2 = value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, bias=b, q16=True)
    wq2, wsc2 = quantize_values_i16(value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, bias=b).view(F_, H * W, 4, 64))
    assert torch.equal(sc2, wsc2) and torch.equal(q2, wq2)


def test_quantiser_propagates_non_finite_blocks():
    """rac_quant_i16_fwd: a (pixel, head) block that holds an inf or a NaN gets a NaN scale, so every value of that block reads
    back as NaN -- as it would travel through the fp32 stream -- instead of saturating to a large finite number; the
    blocks beside it are untouched, and finite blocks keep |x - q * scale| <= scale / 2."""
    from racformer_amd.fused import quantize_values_i16
    g = torch.Generator().manual_seed(3)
    v = torch.randn(2, 5, 4, 64, generator=g) * torch.exp(torch.randn(2, 5, 4, 1, generator=g) * 4.0)
    v[0, 1, 2, 7] = float("nan")
    v[1, 0, 0, 63] = float("inf")
    v[1, 4, 3, 0] = float("-inf")
    v[0, 3, 1] = 0.0
    q, sc = quantize_values_i16(v.to(DEV))
    back = (q.float() * sc.unsqueeze(-1)).cpu()
    bad = torch.zeros(2, 5, 4, dtype=torch.bool)
    bad[0, 1, 2] = bad[1, 0, 0] = bad[1, 4, 3] = True
    assert torch.isnan(sc.cpu())[bad].all() and torch.isfinite(sc.cpu())[~bad].all()
    assert torch.isnan(back[bad]).all()
    ok = ~bad
    assert ((back[ok] - v[ok]).abs() <= 0.5 * sc.cpu()[ok].unsqueeze(-1) + 0.0).all()
    assert int(q.cpu()[0, 3, 1].abs().max()) == 0


def test_head_finish_kernel_matches_torch():
    """rac_head_finish_fwd against the reference's own element-wise tail (nan_to_num of both stacked outputs,
    racformer_transformer.py:58; centre scaling + column order, racformer_head.py:124-131), incl. NaN and +-inf entries:
    bit-exact (the kernel rounds the multiply and the add separately, as the two tensor operations do)."""
    from racformer_amd.fused import head_finish_fused
    g = torch.Generator().manual_seed(5)
    cls = torch.randn(6, 2, 37, 10, generator=g)
    xy = torch.rand(6, 2, 37, 10, generator=g)
    cls[0, 0, 0, 0], cls[1, 1, 3, 2], cls[2, 0, 5, 9] = float("nan"), float("inf"), float("-inf")
    xy[0, 0, 0, 0], xy[0, 0, 1, 2], xy[3, 1, 2, 4], xy[5, 1, 36, 9] = float("nan"), float("inf"), float("-inf"), float("nan")
    pc = syn.PC_RANGE
    want_cls, b = torch.nan_to_num(cls), torch.nan_to_num(xy)
    lo, span = torch.tensor(pc[0:3]), torch.tensor([pc[3] - pc[0], pc[4] - pc[1], pc[5] - pc[2]])
    xyz = b[..., 0:3] * span + lo
    want_box = torch.cat([xyz[..., 0:2], b[..., 3:5], xyz[..., 2:3], b[..., 5:10]], dim=-1)
    got_cls, got_box = head_finish_fused(cls.to(DEV), xy.to(DEV), pc)
    assert torch.equal(got_cls.cpu(), want_cls)
    assert torch.equal(got_box.cpu(), want_box)


def test_conv_pack_live_matches_explicit_assembly():
    """rac_conv_pack_bias_fwd (bias added before the split, frames past the live ones synthesised from the bias) writes the same
    image, bit for bit, as packing the explicitly assembled tensor: hid[:, :Tv] = src + bias, hid[:, Tv:] = bias."""
    from racformer_amd.fused import ConvImage
    g = torch.Generator().manual_seed(11)
    B, T, Tv, C, H, W, Cx = 2, 4, 3, 64, 16, 16, 32
    src = torch.randn(B * Tv, C, H, W, generator=g).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    hid = torch.empty(B, T, C, H, W, device=DEV)
    hid[:, :Tv] = (src + bias.view(1, C, 1, 1)).view(B, Tv, C, H, W)
    hid[:, Tv:] = bias.view(1, 1, C, 1, 1)
    hid = hid.flatten(0, 1).contiguous()
    a = ConvImage(B * T, H, W, Cx + C, torch.device(DEV))
    a.begin([hid]).pack(hid, Cx)
    want = a.xs[..., Cx // 32:, :, :].clone()
    a.xs[..., Cx // 32:, :, :].zero_()
    a.pack_live(src, bias, Cx, T)
    assert torch.equal(a.xs[..., Cx // 32:, :, :], want)
