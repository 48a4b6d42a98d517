#!/usr/bin/env python3
"""Experiment: the launches of the ConvGRU branch (rac_conv_direct_fwd) in isolation at f8 shapes, HIP-event timed, each alone and the
whole chain back to back.  RAC_CD_DEPTH selects the ring depth of an experiment build."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib  # noqa: E402
from racformer_amd.fused import ConvImage, act_image, conv_direct, pack_conv3x3_weight, upsample2x_image  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, Tv, C, H, W, hd = 1, 8, 4, 256, 128, 128, 64
h, w = H // 2, W // 2
x = torch.randn(B * T, C, H, W, device=dev)
img = ConvImage(B * T, H, W, C + hd, dev)
img.begin([x], 2.0).pack(x, 0)
down = torch.nn.Conv2d(C, hd, 3, stride=2, padding=1).to(dev)
gates = torch.nn.Conv2d(2 * hd, 3 * hd, 3, padding=1).to(dev)
up = torch.nn.Conv2d(hd, hd, 3, padding=1).to(dev)
dws, dwa = pack_conv3x3_weight(down.weight, cout=hd)
gx, gxa = pack_conv3x3_weight(gates.weight[:, :hd].contiguous(), cout=3 * hd)
gh, gha = pack_conv3x3_weight(gates.weight[:, hd:].contiguous(), cout=3 * hd)
uw, uwa = pack_conv3x3_weight(up.weight, cout=hd)
bmap = torch.randn(h * w, 3 * hd, device=dev)
live = B * Tv
fus, one = (img.amax, 1.0, 0.0), (None, 0.0, 1.0)
dsc = (img.amax, float(down.weight.detach().abs().sum(dim=(1, 2, 3)).max()), float(down.bias.detach().abs().max()))
down_img, h_img, up_img = (act_image(t, live, *s, hd, dev) for t, s in (("e_down", (h, w)), ("e_h", (h, w)), ("e_up", (H, W))))
xpart = torch.empty(live, h * w, 3 * hd, device=dev)
hs = torch.empty(live, h, w, hd, device=dev)


def k_s2():
    conv_direct(_lib.CD_IMAGE, live, H, W, img.xs, (C + hd) // 32, C // 32, dws, dwa, hd, fus, conv_stride=2, in_frames=(Tv, T, 0),
                bias=down.bias, out_img=down_img, out_chunks_total=2, out_scale=dsc)


def k_xpart():
    conv_direct(_lib.CD_F32, live, h, w, down_img, 2, 2, gx, gxa, 3 * hd, dsc, out_f32=xpart, pixel_map=bmap)


def k_step(t):
    conv_direct(_lib.CD_GRU, B, h, w, h_img, 2, 0 if t == 0 else 2, gh, gha, 3 * hd, one, in_frames=(1, Tv, max(t - 1, 0)), out_img=h_img,
                out_chunks_total=2, out_frames=(1, Tv, t), out_scale=one, xpart=xpart, xpart_frames=(1, Tv, t),
                h_prev=hs if t else None, h_prev_frames=(1, Tv, max(t - 1, 0)), h_out=hs, h_out_frames=(1, Tv, t))


def k_ups():
    upsample2x_image(hs, up_img, 1.0)


def k_upconv():
    conv_direct(_lib.CD_IMAGE, live, H, W, up_img, 2, 2, uw, uwa, hd, one, bias=up.bias, out_img=img.xs, out_chunks_total=(C + hd) // 32,
                out_chunk0=C // 32, out_frames=(Tv, T, 0), out_scale=fus)


def chain():
    k_s2(); k_xpart()
    for t in range(Tv):
        k_step(t)
    k_ups(); k_upconv()


def timed(fn, reps=10, inner=20):
    """median over `reps` of (`inner` launches in a row between one event pair) / inner: an event pair costs a 5-10 us bubble"""
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        for _ in range(inner):
            fn()
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3 / inner


chain()
torch.cuda.synchronize()
print("depth", os.environ.get("RAC_CD_DEPTH", "3"))
for name, fn in (("s2", k_s2), ("xpart", k_xpart), ("step0", lambda: k_step(0)), ("step1", lambda: k_step(1)), ("upsample", k_ups),
                 ("upconv", k_upconv), ("chain", chain)):
    print(f"{name:10s} {timed(fn):8.1f} us")
