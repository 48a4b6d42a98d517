"""What are the deviating values?  rac_msmv_fwd into a buffer pre-filled with a sentinel, beside a looping mixing kernel."""
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
loc = torch.rand(S, Q, P, 3, generator=gen) * 0.9 + 0.05
loc[..., 2] = torch.randint(0, N, (S, Q, P), generator=gen).float() / (N - 1)
wts = torch.softmax(torch.randn(S, Q, P, L, generator=gen), dim=-1)
loc, wts = loc.to(DEV), wts.to(DEV)
mx = torch.randn(1, 900, 4, 96, 64, generator=gen).to(DEV)
mp = (torch.randn(1, 900, 65536, generator=gen) * 0.1).to(DEV)
SENT = 12345.0
bufs = [torch.empty(1, Q, G, T * P, C, device=DEV) for _ in range(10)]
def probe(i):
    bufs[i].fill_(SENT)
    return msmv_forward(feats, loc, wts, out_layout=1, num_frames=T, num_groups=G, out=bufs[i])
want = probe(0).clone(); torch.cuda.synchronize()
# per-level partial results to recognise "one level / tap missing"
parts = []
for l in range(L):
    w1 = torch.zeros_like(wts); w1[..., l] = wts[..., l]
    parts.append(msmv_forward(feats, loc, w1, out_layout=1, num_frames=T, num_groups=G).clone())
torch.cuda.synchronize()
sb = torch.cuda.Stream()
shown = 0
for it in range(8):
    with torch.cuda.stream(sb):
        for _ in range(40):
            mixing_fused(mx, mp, 96, 4, split=True, f16x3=True)
    outs = [probe(i) for i in range(10)]
    torch.cuda.synchronize()
    for o in outs:
        d = o != want
        if d.any() and shown < 8:
            shown += 1
            got, w = o[d], want[d]
            is_sent = (got == SENT).float().mean().item()
            # which single level's contribution is missing / doubled?
            expl = []
            for l in range(L):
                pl = parts[l][d]
                expl.append(((got - (w - pl)).abs() < 1e-5).float().mean().item())
            print("deviating elements %d | == sentinel: %.2f | == result minus level l: %s | got/want sample %s / %s" %
                  (int(d.sum()), is_sent, ["%.2f" % e for e in expl], [round(float(x), 4) for x in got[:4]], [round(float(x), 4) for x in w[:4]]), flush=True)
