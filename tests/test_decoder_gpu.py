"""Decoder-level parity on the GPU: product modules (HIP ops) against the reference goldens and,
stage by stage, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from parity import decoder_parity, oracle_decoder, run_with_reference_views
from racformer_amd import synthetic as syn
from racformer_amd.transformer import RaCFormerTransformer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STAGES = ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling", "mixing", "ffn")


def run_gpu(cfg, seed, wseed, stages=None, fused=True, force_views=None):
    """-> (cls, box, views [layers,S,Q,P]); ``force_views`` imposes the camera choices (tests/parity.py)."""
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    tr.decoder.decoder_layer.fused = fused
    syn.fill_params(tr, wseed)
    tr = tr.to(DEV)
    smp = tr.decoder.decoder_layer.sampling
    taps = smp.capture_loc = []
    if force_views is not None:
        smp.force_views = [torch.as_tensor(np.asarray(v)).to(DEV).contiguous() for v in force_views]
    if stages is not None:
        del stages[:]
    qb, qf = syn.make_queries(cfg, seed)
    with torch.no_grad():
        cls, box = tr(qb.to(DEV), qf.to(DEV), [f.to(DEV) for f in syn.make_pyramid(cfg, seed)],
                      syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV), None,
                      syn.make_img_metas(cfg), stages_per_layer=stages)
    torch.cuda.synchronize()
    return cls.cpu(), box.cpu(), torch.stack([R.views_of(l.cpu(), cfg.num_cams) for l in taps])


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name,cfg", [("decoder_small.npz", syn.SMALL), ("decoder_small6.npz", syn.SMALL6)])
def test_decoder_small_vs_reference_golden(golden_dir, name, cfg, fused):
    g = np.load(os.path.join(golden_dir, name))
    stages = []
    (cls, box, _), _ = run_with_reference_views(
        lambda force: run_gpu(cfg, int(g["seed"]), int(g["weight_seed"]), stages, fused, force), g["views"], name)
    last = cfg.num_layers - 1
    for s in STAGES:
        err = (stages[0][s].cpu() - torch.from_numpy(g[f"{s}_L0"])).abs().max().item()
        assert err < 1e-4, (s, err)
        ref = torch.from_numpy(g[f"{s}_L{last}"])
        err5 = ((stages[last][s].cpu() - ref).abs() / (1 + ref.abs())).max().item()
        assert err5 < 1e-3, (s, "last layer", err5)
    decoder_parity(cls, box, g["cls"], g["box"], what=name, tail_budget=None)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name,cfg", [("decoder_f8.npz", syn.F8), ("decoder_f8_3cam.npz", syn.F8_3CAM)])
def test_decoder_f8_vs_reference_golden(golden_dir, name, cfg, fused):
    """BASELINE configs 3/5 shapes in fp32, fused plan and the reference's op decomposition: free-running criterion of
    tests/parity.py + the fixture's stage outputs of layer 0 and layer 5 (first 16 queries)."""
    g = np.load(os.path.join(golden_dir, name))
    stages = []
    (cls, box, _), _ = run_with_reference_views(
        lambda force: run_gpu(cfg, int(g["seed"]), int(g["weight_seed"]), stages, fused, force), g["views"], name)
    last = cfg.num_layers - 1
    for s in STAGES:
        got = stages[0][s][:, :16].cpu()
        err = (got - torch.from_numpy(g[f"{s}_L0_head"])).abs().max().item()
        assert err < 2e-4, (s, err)
        # layer 5 of the free-running stack: five layers of amplified rounding upstream (tests/parity.py), relative
        ref = torch.from_numpy(g[f"{s}_L{last}_head"])
        err5 = ((stages[last][s][:, :16].cpu() - ref).abs() / (1 + ref.abs())).flatten(2).amax(-1)
        assert err5.median().item() < 1e-4 and err5.max().item() < 5e-2, (s, "layer 5", err5.median().item(), err5.max().item())
    decoder_parity(cls, box, g["cls"], g["box"], what=name)


def test_decoder_small_vs_oracle_stagewise():
    """Same seeded inputs through the oracle (CPU) and the HIP path; every stage of every layer."""
    cfg, seed, wseed = syn.SMALL6, 5, 6
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    ost = []
    ocls, obox, oviews = oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                        syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg, ost)
    gst = []
    (cls, box, _), _ = run_with_reference_views(lambda force: run_gpu(cfg, seed, wseed, gst, True, force), oviews, "small6")
    for li in range(cfg.num_layers):
        for s in STAGES:
            err = (gst[li][s].cpu() - ost[li][s]).abs().max().item()
            assert err < 1e-3, (li, s, err)
    decoder_parity(cls, box, ocls, obox, what="small6 vs oracle", tail_budget=None)


def _writer(cfg, seed):
    from racformer_amd.fpn_writer import FPNOutputWriter
    wr = FPNOutputWriter(num_levels=cfg.num_levels, num_cams=cfg.num_cams)
    with torch.no_grad():
        for i, m in enumerate(wr.fpn_convs):
            m.conv.weight.copy_(torch.from_numpy(syn.rng_normal(seed * 100 + i, tuple(m.conv.weight.shape), 1.0 / 48.0)))
            m.conv.bias.copy_(torch.from_numpy(syn.rng_normal(seed * 100 + 50 + i, (256,), 0.1)))
    return wr


@pytest.mark.parametrize("images,hw", [(12, (16, 44)), (12, (8, 22)), (6, (4, 11)), (6, (2, 6)), (48, (16, 44)), (48, (8, 22)),
                                       (6, (32, 88))])
def test_fpn_output_writer_vs_float64_conv_and_oracle_regroup(images, hw):
    """rac_fpn_conv_fwd (row f2: the neck's per-level 3x3 output convolution writing [B*T*G, N, H, W, 64]) against a float64
    convolution on the CPU followed by the ORACLE's regroup (models/racformer_transformer.py:112-124): every level shape of
    the reduced and the f8 rigs that does not tile into 256 pixels (ragged last tile, W % 4 != 0, W > 128 below), 6 cameras."""
    cfg = syn.SMALL6
    wr = _writer(cfg, 3)
    g = torch.Generator().manual_seed(images + hw[0])
    x = torch.randn(images, 256, *hw, generator=g)
    x[0, :, 0, 0] = 0.0
    conv = wr.fpn_convs[1].conv
    want = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    fp32 = torch.nn.functional.conv2d(x, conv.weight, conv.bias, padding=1)
    want_g = R.regroup_pyramid([want.view(1, images, 256, *hw)], cfg.num_cams)[0]
    wr = wr.to(DEV)
    got = wr.forward([x.to(DEV)] * 2)[1].cpu()       # (level index 1's weights)
    assert tuple(got.shape) == tuple(want_g.shape) == (images // cfg.num_cams * 4, cfg.num_cams, hw[0], hw[1], 64)
    e_kernel = (got.double() - want_g).abs().max().item()
    e_fp32 = (fp32.double() - want).abs().max().item()
    # fp32-convolution accuracy: within a few fp32 roundings of the 2304-term sums (a CPU fp32 convolution is the yardstick)
    assert e_kernel < max(4 * e_fp32, 3e-6 * float(want.abs().max())) + 1e-6, (e_kernel, e_fp32)


def test_fpn_output_writer_wide_rows_f8_level0():
    """Level 0 of the f8 rig (64 x 176: rows wider than one 128-column piece of the pack kernel, 44 full tiles per image) on two
    (batch, frame) groups of 3 cameras, against MIOpen-free float64 on a strided probe (the full float64 convolution of this
    level on the CPU takes minutes) and the oracle's regroup."""
    cfg = syn.F8_3CAM
    wr = _writer(cfg, 4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 256, 64, 176, generator=g)
    conv = wr.fpn_convs[0].conv
    rows = [0, 1, 31, 62, 63]
    # float64 reference on a few output rows: convolve the padded row bands
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    want_rows = torch.stack([torch.nn.functional.conv2d(xp[:, :, r:r + 3], conv.weight.double(), conv.bias.double())[:, :, 0]
                             for r in rows], dim=2)                                         # [6,256,len(rows),176]
    got = wr.to(DEV).forward([x.to(DEV)])[0].cpu()                                         # [2*4, 3, 64, 176, 64]
    full = torch.zeros(1, 6, 256, 64, 176, dtype=torch.float64)
    full[0, :, :, rows] = want_rows
    want_g = R.regroup_pyramid([full], cfg.num_cams)[0][:, :, rows]
    err = (got[:, :, rows].double() - want_g).abs().max().item()
    assert err < 2e-5 * float(want_rows.abs().max()) + 1e-6, err


def test_decoder_on_writer_output_vs_oracle():
    """Row f2 end to end, against the oracle (not against the product's own regroup path): the neck's output convolutions
    run on the CPU in fp32 and feed the ORACLE's decoder in the reference layout [B, T*N, 256, H, W]; the product's
    FPNOutputWriter writes the grouped channel-last pyramid from the same laterals and the ``pregrouped`` decoder consumes it
    without rac_regroup_fwd.  Literal criterion on all six layers; a pyramid in the reference layout is refused."""
    cfg, seed, wseed = syn.SMALL6, 7, 8
    wr = _writer(cfg, 9)
    laterals = [f[0] for f in syn.make_pyramid(cfg, seed)]                                  # [T*N, 256, H, W] per level
    with torch.no_grad():
        pyr = [wr.fpn_convs[i].conv(x)[None] for i, x in enumerate(laterals)]
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    lss, radar = syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1)
    ocls, obox, oviews = oracle_decoder(R, sd, qb, qf, pyr, lss, radar, syn.make_img_metas(cfg), cfg)
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    tr, wr = tr.to(DEV), wr.to(DEV)
    tr.decoder.pregrouped = True
    layer = tr.decoder.decoder_layer

    def run(force):
        layer.sampling.capture_loc = []
        layer.sampling.force_views = [v.to(DEV).contiguous() for v in force] if force is not None else None
        with torch.no_grad():
            grouped = wr([x.to(DEV) for x in laterals])
            cls, box = tr(qb.to(DEV), qf.to(DEV), grouped, lss.to(DEV), radar.to(DEV), None, syn.make_img_metas(cfg))
        torch.cuda.synchronize()
        return cls.cpu(), box.cpu(), torch.stack([R.views_of(l.cpu(), cfg.num_cams) for l in layer.sampling.capture_loc])

    (cls, box, _), _ = run_with_reference_views(run, oviews, "writer -> pregrouped decoder")
    decoder_parity(cls, box, ocls, obox, what="writer -> pregrouped decoder vs oracle", tail_budget=None)
    with torch.no_grad(), pytest.raises(RuntimeError, match="pregrouped"):
        tr(qb.to(DEV), qf.to(DEV), [p.to(DEV) for p in pyr], lss.to(DEV), radar.to(DEV), None, syn.make_img_metas(cfg))
