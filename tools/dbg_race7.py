import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACFORMER_HIP_LIB"] = os.path.join(ROOT, "build/lib_msmv_check.so")
from racformer_amd import synthetic as syn, _lib
from racformer_amd.fused import mixing_fused
from racformer_amd.msmv import msmv_forward
DEV = "cuda:0"
cfg = syn.F8
gen = torch.Generator().manual_seed(3)
T, G, NP, D, Q, L = cfg.num_frames, cfg.num_groups, cfg.num_points, cfg.img_depth_num, cfg.num_query, cfg.num_levels
P, S, N, C = NP * D, 32, cfg.num_cams, 64
feats = [torch.randn(S, N, h, w, C, generator=gen).to(DEV) for (h, w) in cfg.fpn_hw]
loc = torch.rand(S, Q, P, 3, generator=gen) * 0.9 + 0.05
loc[..., 2] = torch.randint(0, N, (S, Q, P), generator=gen).float() / (N - 1)
wts = torch.softmax(torch.randn(S, Q, P, L, generator=gen), dim=-1)
loc, wts = loc.to(DEV), wts.to(DEV)
mx = torch.randn(1, 900, 4, 96, 64, generator=gen).to(DEV)
mp = (torch.randn(1, 900, 65536, generator=gen) * 0.1).to(DEV)
probe = lambda: msmv_forward(feats, loc, wts, out_layout=1, num_frames=T, num_groups=G)
h = _lib.lib()
h.rac_dbg_read.restype = ctypes.c_int
cnt = (ctypes.c_uint * 8)()
want = probe().clone(); torch.cuda.synchronize()
h.rac_dbg_read(cnt, 1); print("solo: counters", list(cnt)[:4])
for _ in range(10):
    probe()
h.rac_dbg_read(cnt, 1); print("solo x10: counters", list(cnt)[:4])
sb = torch.cuda.Stream()
bad = 0
for it in range(6):
    with torch.cuda.stream(sb):
        for _ in range(40):
            mixing_fused(mx, mp, 96, 4, split=True, f16x3=True)
    outs = [probe() for _ in range(15)]
    torch.cuda.synchronize()
    bad += sum(not torch.equal(o, want) for o in outs)
h.rac_dbg_read(cnt, 1)
print("beside mixing: deviating launches", bad, "of 90; counters [lds offsets differ, lds weights differ, tap data differ, wave-steps]", list(cnt)[:4])
