"""Which of the product's kernels deviate beside a looping mixing (or conv) kernel?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from racformer_amd import synthetic as syn
from racformer_amd.fused import (box_prep, generator_fused, outproj_fused, mixing_fused, pack_conv3x3_weight, conv3x3_fused,
                                 value_proj_fused, pack_gemm_split_weight, sasa_fused, add_ln, row_gemm, row_seg, rowgemm_launch)
from racformer_amd.transformer import regroup_pyramid
DEV = "cuda:0"
cfg = syn.F8
gen = torch.Generator().manual_seed(3)
Q = cfg.num_query
feats = [f.to(DEV) for f in syn.make_pyramid(cfg, 4)]
qb = syn.make_queries(cfg, 4)[0].to(DEV)
table = box_prep(qb, list(cfg.pc_range))
x_img = (torch.randn(900, 512, generator=gen) * 100).to(torch.float16).to(DEV)
w_img = (torch.randn(65536, 8, 64, generator=gen) * 100).to(torch.float16).to(DEV)
bias = torch.randn(65536, generator=gen).to(DEV)
z_img = (torch.randn(900, 1024, 64, generator=gen) * 100).to(torch.float16).to(DEV)
wo_img = (torch.randn(256, 1024, 64, generator=gen) * 100).to(torch.float16).to(DEV)
mx = torch.randn(1, 900, 4, 96, 64, generator=gen).to(DEV)
mp = (torch.randn(1, 900, 65536, generator=gen) * 0.1).to(DEV)
qkv = torch.randn(1, 900, 768, generator=gen).to(DEV); tau = torch.rand(1, 900, 8, generator=gen).to(DEV)
conv_w = torch.randn(256, 320, 3, 3, generator=gen).to(DEV) * 0.02
ws, alpha = pack_conv3x3_weight(conv_w)
cx = torch.randn(8, 320, 128, 128, generator=gen).to(DEV)
vpw = torch.randn(256, 256, generator=gen).to(DEV) * 0.05
vimg, valpha = pack_gemm_split_weight(vpw)
vx = torch.randn(8, 256, 128, 128, generator=gen).to(DEV)
ln = torch.nn.LayerNorm(256).to(DEV)
xa = torch.randn(900, 256, generator=gen).to(DEV)
lw = torch.randn(512, 256, generator=gen).to(DEV) * 0.05
lb = torch.randn(512, generator=gen).to(DEV)
def rowg():
    out = torch.empty(900, 512, device=DEV)
    rowgemm_launch([row_gemm([row_seg(xa, norm=ln, relu=True)], lw, lb, out)], 900)
    return out
K = {
    "generator": lambda: generator_fused(x_img, w_img, bias, 1e-6, timer_name=None),
    "outproj": lambda: outproj_fused(z_img, wo_img, 32),
    "mixing f16x3": lambda: mixing_fused(mx, mp, 96, 4, split=True, f16x3=True),
    "mixing f32": lambda: mixing_fused(mx, mp, 96, 4, split=True, f16x3=False),
    "regroup": lambda: regroup_pyramid(feats, cfg.num_cams)[0],
    "conv3x3": lambda: conv3x3_fused([cx], ws, alpha, None),
    "value_proj": lambda: value_proj_fused(vx, vimg, valpha),
    "sasa": lambda: sasa_fused(qkv, tau, qb, 8, list(cfg.pc_range), box_table=table),
    "add_ln": lambda: add_ln(xa, ln),
    "rowgemm": rowg,
}
sb = torch.cuda.Stream()
for nname in sys.argv[1:] or ["mixing f16x3"]:
    noise = K[nname]
    for name, fn in K.items():
        want = fn().clone(); torch.cuda.synchronize()
        bad = 0
        for it in range(5):
            with torch.cuda.stream(sb):
                for _ in range(30):
                    noise()
            outs = [fn() for _ in range(12)]
            torch.cuda.synchronize()
            bad += sum(not torch.equal(o, want) for o in outs)
        print("noise = %-14s victim = %-14s deviating launches: %3d of 60" % (nname, name, bad), flush=True)
