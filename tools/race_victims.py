#!/usr/bin/env python3
"""Which kernels produce other bits when they run beside another stream's MFMA kernels?  (GPU box.)

usage:  [RACFORMER_HIP_LIB=build/lib_<variant>.so] python3 tools/race_victims.py [out.json]

Victims: the product's stand-alone gather operator (rac_msmv_fwd of whatever library build is loaded: the in-tree one, or a
diagnostic build -- packed-FP32 accumulation, packed with every tap load issued before the first FMA, ...), and LIBRARY kernels
that are captured in the same graphs and are not rebuilt by csrc/Makefile's -packed-fp32-ops switch: MIOpen's Winograd
convolution of the ConvGRU gates (F.conv2d, [1,128,64,64] -> 192), torch's index_select over 2 M rows, a torch elementwise
addcmul.  Aggressors, looping on a second stream: mixing_c64_f16x3_kernel and conv3x3_f16x3_kernel (the two kernels beside
which round 3 saw the deviations).  Every victim is first run alone (reference bits), then 5 x 12 times beside the aggressor;
a launch deviates if its output is not bit-identical to the reference."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib, synthetic as syn  # noqa: E402
from racformer_amd.fused import conv3x3_fused, mixing_fused, pack_conv3x3_weight  # noqa: E402
from racformer_amd.msmv import msmv_forward  # noqa: E402

DEV = "cuda:0"


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    cfg = syn.F8
    g = torch.Generator().manual_seed(3)
    S, N, Q, C = cfg.num_frames * cfg.num_groups, cfg.num_cams, cfg.num_query, cfg.channels
    P, L = cfg.num_points * cfg.img_depth_num, cfg.num_levels
    feats = [torch.randn(S, N, h, w, C, generator=g).to(DEV) for (h, w) in cfg.fpn_hw]
    loc = torch.rand(S, Q, P, 3, generator=g) * 0.98 + 0.01             # all in range: no tap outside a map
    loc[..., 2] = torch.randint(0, N, (S, Q, P), generator=g).float() / (N - 1)
    wts = torch.softmax(torch.randn(S, Q, P, L, generator=g), dim=-1)
    loc, wts = loc.to(DEV), wts.to(DEV)
    mx = torch.randn(1, 900, 4, 96, 64, generator=g).to(DEV)
    mp = (torch.randn(1, 900, 65536, generator=g) * 0.1).to(DEV)
    conv_w = torch.randn(256, 320, 3, 3, generator=g).to(DEV) * 0.02
    ws, alpha = pack_conv3x3_weight(conv_w)
    cx = torch.randn(8, 320, 128, 128, generator=g).to(DEV)
    gx = torch.randn(1, 128, 64, 64, generator=g).to(DEV)
    gw = torch.randn(192, 128, 3, 3, generator=g).to(DEV) * 0.05
    table = torch.randn(2_000_000, 16, generator=g).to(DEV)
    idx = torch.randint(0, 2_000_000, (2_000_000,), generator=g).to(DEV)
    ea, eb, ec = (torch.randn(8_000_000, generator=g).to(DEV) for _ in range(3))
    victims = {
        "rac_msmv_fwd (this library build)": lambda: msmv_forward(feats, loc, wts, out_layout=_lib.OUT_BQGTPC, num_frames=cfg.num_frames,
                                                                  num_groups=cfg.num_groups),
        "MIOpen conv2d 128->192 3x3 on 64x64 (ConvGRU gates)": lambda: F.conv2d(gx, gw, None, padding=1),
        "torch index_select, 2M rows of 64 B": lambda: table.index_select(0, idx),
        "torch addcmul, 8M elements": lambda: torch.addcmul(ea, eb, ec),
    }
    aggressors = {
        "mixing_c64_f16x3_kernel": lambda: mixing_fused(mx, mp, 96, 4, split=True, f16x3=True),
        "conv3x3_f16x3_kernel": lambda: conv3x3_fused([cx], ws, alpha, None),
    }
    side = torch.cuda.Stream()
    rows = []
    with torch.no_grad():
        for an, noise in aggressors.items():
            noise()
            for vn, fn in victims.items():
                fn()
                torch.cuda.synchronize()
                want = fn().clone()
                torch.cuda.synchronize()
                bad = total = 0
                for _ in range(5):
                    with torch.cuda.stream(side):
                        for _ in range(30):
                            noise()
                    outs = [fn() for _ in range(12)]
                    torch.cuda.synchronize()
                    bad += sum(not torch.equal(o, want) for o in outs)
                    total += len(outs)
                rows.append({"aggressor": an, "victim": vn, "deviating_launches": bad, "launches": total})
                print(f"beside {an:26s} {vn:55s} deviating launches: {bad:3d} of {total}", flush=True)
    res = {"library": os.environ.get("RACFORMER_HIP_LIB", "in-tree libracformer_hip.so"), "results": rows}
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
