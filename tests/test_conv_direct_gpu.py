"""rac_conv_direct_fwd / rac_upsample2x_image_fwd / rac_conv3x3_temporal_fwd (round 5: the ConvGRU branch of
RadarBEVTemporalEncoder, models/racformer_transformer.py:645-656, 674-720, on own kernels) against float64 convolutions and the
torch modules they replace.  Every mode of the direct convolution, both strides, compile-time and run-time chunk counts, ragged
pixel tiles, frame maps; the GRU update against ConvGRUCell; the fold of the constant hidden half against the explicit image."""
import math

import pytest
import torch
import torch.nn.functional as F

from racformer_amd import _lib
from racformer_amd import synthetic as syn  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def act_scale(bound):
    """host copy of rac_act_scale (csrc/rac_common.h)"""
    if not (1e-30 < bound < 3e38):
        return 1.0
    return 2.0 ** (14 - math.frexp(bound)[1])


def image_values(img, scale):
    """activation image [frames, H+2, W+2, chunks, 2, 32] f16 -> ([frames, C, H, W] float64 interior values, border max |.|)"""
    v = (img[..., 0, :].double() + img[..., 1, :].double()) / scale          # [f, H+2, W+2, chunks, 32]
    f, hp, wp, ch, _ = v.shape
    v = v.reshape(f, hp, wp, ch * 32)
    inner = v[:, 1:-1, 1:-1].permute(0, 3, 1, 2).contiguous()
    border = torch.cat([v[:, 0].abs().flatten(), v[:, -1].abs().flatten(), v[:, :, 0].abs().flatten(), v[:, :, -1].abs().flatten()])
    return inner.cpu(), float(border.max())


@pytest.mark.parametrize("N,T,Tv,H,W,cin", [(8, 8, 4, 32, 32, 256), (4, 4, 2, 16, 24, 64), (3, 3, 3, 12, 20, 96)])
def test_stride2_image_mode_vs_float64(N, T, Tv, H, W, cin):
    """The downsample convolution out of the fusion image's x half (of the LIVE frames of each group) into an activation image whose
    scale comes from the weights' bound; cin = 256 takes the straight-line K loop (8 chunks), the others the run-time loop; 12 x 20
    maps leave a ragged last pixel tile (60 output pixels)."""
    from racformer_amd.fused import ConvImage, act_image, conv_direct, pack_conv3x3_weight
    g = torch.Generator().manual_seed(N * H + cin)
    conv = torch.nn.Conv2d(cin, 64, 3, stride=2, padding=1)
    torch.nn.init.normal_(conv.bias, std=0.3)
    x = torch.randn(N, cin, H, W, generator=g) * 2.0
    hidden = 64
    img = ConvImage(N, H, W, cin + hidden, torch.device(DEV))
    xg = x.to(DEV)
    img.begin([xg], 1.5).pack(xg, 0)
    ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV), cout=64)
    l1, bmax = float(conv.weight.detach().abs().sum(dim=(1, 2, 3)).max()), float(conv.bias.detach().abs().max())
    groups = N // T
    live = groups * Tv
    out = act_image("t_s2", live, H // 2, W // 2, 64, torch.device(DEV))
    out.zero_()
    conv_direct(_lib.CD_IMAGE, live, H, W, img.xs, (cin + hidden) // 32, cin // 32, ws, alpha, 64, (img.amax, 1.0, 0.0), conv_stride=2,
                in_frames=(Tv, T, 0), bias=conv.bias.detach().to(DEV), out_img=out, out_chunks_total=2, out_scale=(img.amax, l1, bmax))
    torch.cuda.synchronize()
    amax = float(img.amax)
    assert amax >= float(x.abs().max()) - 1e-6
    got, border = image_values(out, act_scale(l1 * amax + bmax))
    sel = [gidx * T + t for gidx in range(groups) for t in range(Tv)]
    want = F.conv2d(x[sel].double(), conv.weight.double(), conv.bias.double(), stride=2, padding=1)
    assert border == 0.0
    assert (got - want).abs().max().item() <= 4e-6 * want.abs().max().item()


@pytest.mark.parametrize("frames,H,W,cin,cout", [(4, 16, 16, 64, 192), (2, 12, 20, 32, 64), (3, 8, 8, 128, 128)])
def test_f32_mode_vs_float64(frames, H, W, cin, cout):
    """Channel-last fp32 output with bias and per-pixel map (the gates convolution's x half + the composed bias map)."""
    from racformer_amd.fused import ConvImage, conv_direct, pack_conv3x3_weight
    g = torch.Generator().manual_seed(frames + H + cin)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    x = torch.randn(frames, cin, H, W, generator=g)
    pmap = torch.randn(H * W, cout, generator=g)
    img = ConvImage(frames, H, W, cin, torch.device(DEV))
    xg = x.to(DEV)
    img.begin([xg]).pack(xg, 0)
    ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV), cout=cout)
    out = torch.empty(frames, H * W, cout, device=DEV)
    conv_direct(_lib.CD_F32, frames, H, W, img.xs, cin // 32, cin // 32, ws, alpha, cout, (img.amax, 1.0, 0.0),
                bias=conv.bias.detach().to(DEV), out_f32=out, pixel_map=pmap.to(DEV))
    want = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1).reshape(frames, H * W, cout) \
        + pmap.double()
    assert (out.double().cpu() - want).abs().max().item() <= 4e-6 * want.abs().max().item()


@pytest.mark.parametrize("B,Tv,H,W", [(1, 4, 16, 16), (2, 3, 8, 12)])
def test_gru_mode_vs_convgru_cell(B, Tv, H, W):
    """The recurrence: h half of the gates convolution (matching layer composed in) + x part + the GRU update in the epilogue,
    Tv steps from h = 0, against ConvGRU / ConvGRUCell (the product's own torch restatement of racformer_transformer.py:696-720,
    evaluated in float64 on the CPU); B = 2 exercises the frame maps (frame = b * Tv + t)."""
    from racformer_amd.fused import act_image, conv_direct, pack_conv3x3_weight
    from racformer_amd.transformer import ConvGRU, convgru_fused_pack
    torch.manual_seed(B * 7 + H)
    gru = ConvGRU(64, 64, 3)
    cell = gru.convGRUCell
    for m in (cell.gates_conv, cell.matching_layer):
        torch.nn.init.normal_(m.bias, std=0.2)
    x = torch.randn(B, Tv, 64, H, W)
    with torch.no_grad():
        want = gru.double()(x.double())[:, :Tv]                                    # [B, Tv, 64, H, W]
        gru.float()
        gw, gmap = convgru_fused_pack(gru, H, W)
    gh, gh_a = pack_conv3x3_weight(gw[:, 64:].contiguous().to(DEV), cout=192)
    # x part in float64 on the host (its kernel is test_f32_mode's subject): conv(x, W_gx) + bias map, channel-last
    xpart = (F.conv2d(x.flatten(0, 1).double(), gw[:, :64].double(), None, padding=1) + gmap.double()[None]).permute(0, 2, 3, 1) \
        .reshape(B * Tv, H * W, 192).float().to(DEV).contiguous()
    h_img = act_image("t_gru_h", B * Tv, H, W, 64, torch.device(DEV))
    hs = torch.zeros(B * Tv, H * W, 64, device=DEV)
    one = (None, 0.0, 1.0)
    for t in range(Tv):
        conv_direct(_lib.CD_GRU, B, H, W, h_img, 2, 0 if t == 0 else 2, gh, gh_a, 192, one, in_frames=(1, Tv, max(t - 1, 0)),
                    out_img=h_img, out_chunks_total=2, out_frames=(1, Tv, t), out_scale=one, xpart=xpart, xpart_frames=(1, Tv, t),
                    h_prev=hs if t else None, h_prev_frames=(1, Tv, max(t - 1, 0)), h_out=hs, h_out_frames=(1, Tv, t))
    torch.cuda.synchronize()
    got = hs.view(B, Tv, H, W, 64).permute(0, 1, 4, 2, 3).double().cpu()
    assert (got - want).abs().max().item() < 5e-6
    img_vals, border = image_values(h_img, act_scale(1.0))
    assert border == 0.0 and (img_vals.view(B, Tv, 64, H, W) - got).abs().max().item() < 2.0 ** -21


def test_upsample_image_vs_torch():
    from racformer_amd.fused import act_image, upsample2x_image
    torch.manual_seed(2)
    x = torch.rand(3, 10, 6, 64) * 2 - 1                                          # channel-last, |.| <= 1
    img = act_image("t_up", 3, 20, 12, 64, torch.device(DEV))
    upsample2x_image(x.to(DEV), img, 1.0)
    torch.cuda.synchronize()
    want = F.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=True)
    got, border = image_values(img, act_scale(1.0))
    assert border == 0.0 and (got - want).abs().max().item() < 2e-6


@pytest.mark.parametrize("q16", [False, True])
def test_temporal_conv_folds_the_constant_hidden_half(q16):
    """rac_conv3x3_temporal_fwd: frames past the live ones skip their (constant) hidden chunks and add the map that holds the
    constant's contribution through the zero padding -- against rac_conv3x3_fwd on the image with the constant packed explicitly
    (fp32-rounding apart: the fold is exact algebra), live frames bit for bit."""
    from racformer_amd.fused import ConvImage, pack_conv3x3_weight, quantize_values_i16
    from racformer_amd.transformer import dead_frame_bias_map
    torch.manual_seed(4)
    B, T, Tv, H, W, Cx, Ch = 2, 4, 2, 16, 16, 64, 64
    conv = torch.nn.Conv2d(Cx + Ch, 256, 3, padding=1)
    x = torch.randn(B * T, Cx, H, W).to(DEV)
    hv = torch.randn(B * Tv, Ch, H, W).to(DEV) * 0.5
    b_up = torch.randn(Ch) * 0.3
    pmap = torch.randn(H * W, 256)
    dead = (pmap.double() + dead_frame_bias_map(conv.weight.detach().double()[:, Cx:], b_up.double(), H, W)).float()
    ws, alpha = pack_conv3x3_weight(conv.weight.to(DEV))
    img = ConvImage(B * T, H, W, Cx + Ch, torch.device(DEV))
    img.begin([x, hv], float(b_up.abs().max()) + float(hv.abs().max())).pack(x, 0).pack_live(hv, b_up.to(DEV), Cx, T)
    want = img.conv(ws, alpha, None, pmap.to(DEV))
    if q16:
        gq, gs = img.conv_temporal(ws, alpha, pmap.to(DEV), dead.to(DEV), Cx, T, Tv, q16=True)
        got = gq.float() * gs.unsqueeze(-1)
        wq, wsc = quantize_values_i16(want.view(B * T, H * W, 4, 64))
        want_q = (wq.float() * wsc.unsqueeze(-1)).view(B, T, H * W, 256)
        got = got.view(B, T, H * W, 256)
        assert torch.equal(got[:, :Tv], want_q[:, :Tv])
        assert (got[:, Tv:] - want_q[:, Tv:]).abs().max().item() <= 2.0 ** -13 * want.abs().max().item()
        return
    got = img.conv_temporal(ws, alpha, pmap.to(DEV), dead.to(DEV), Cx, T, Tv).view(B, T, H, W, 256)
    want = want.view(B, T, H, W, 256)
    assert torch.equal(got[:, :Tv], want[:, :Tv])
    assert (got[:, Tv:] - want[:, Tv:]).abs().max().item() <= 4e-6 * want.abs().max().item()
