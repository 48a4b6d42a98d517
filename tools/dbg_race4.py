"""rac_msmv_fwd (build under test: RACFORMER_HIP_LIB) beside a looping mixing kernel: deviating launches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import synthetic as syn
from racformer_amd.fused import mixing_fused
from racformer_amd.msmv import msmv_forward
DEV = "cuda:0"
cfg = syn.F8
gen = torch.Generator().manual_seed(3)
T, G, NP, D, Q, L = cfg.num_frames, cfg.num_groups, cfg.num_points, cfg.img_depth_num, cfg.num_query, cfg.num_levels
P, S, N, C = NP * D, 32, cfg.num_cams, 64
feats = [torch.randn(S, N, h, w, C, generator=gen).to(DEV) for (h, w) in cfg.fpn_hw]
mode = sys.argv[1] if len(sys.argv) > 1 else "stress"
loc = torch.rand(S, Q, P, 3, generator=gen) * (1.1 if mode == "stress" else 0.9) - (0.05 if mode == "stress" else -0.05)
loc[..., 2] = torch.randint(0, N, (S, Q, P), generator=gen).float() / (N - 1)
wts = torch.softmax(torch.randn(S, Q, P, L, generator=gen), dim=-1)
loc, wts = loc.to(DEV), wts.to(DEV)
mx = torch.randn(1, 900, 4, 96, 64, generator=gen).to(DEV)
mp = (torch.randn(1, 900, 65536, generator=gen) * 0.1).to(DEV)
probe = lambda: msmv_forward(feats, loc, wts, out_layout=1, num_frames=T, num_groups=G)
want = probe().clone(); torch.cuda.synchronize()
sb = torch.cuda.Stream()
bad = 0
for it in range(8):
    with torch.cuda.stream(sb):
        for _ in range(40):
            mixing_fused(mx, mp, 96, 4, split=True, f16x3=True)
    outs = [probe() for _ in range(15)]
    torch.cuda.synchronize()
    bad += sum(not torch.equal(o, want) for o in outs)
print(os.path.basename(os.environ.get("RACFORMER_HIP_LIB", "default")), "locations:", mode, "-> deviating launches", bad, "of 120", flush=True)
