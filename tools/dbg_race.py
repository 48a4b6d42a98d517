"""Which stage of the decoder layer deviates when another plan runs beside it?  Eager forward on stream A with stage hooks,
a captured plan replayed on stream B as noise; per stage and layer the first deviation from the solo run is reported."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from racformer_amd import synthetic as syn
from racformer_amd.graph import CapturedStep
from racformer_amd.fused import scratch_namespace
from test_parity_gpu import build_head
DEV = "cuda:0"
cfg = syn.F8
g = np.load(os.path.join(ROOT, "tests/golden/head_f8.npz"))
head = build_head(cfg, g, int(g["seed"]), int(g["weight_seed"]))
seed = int(g["seed"])
feats = [f.to(DEV) for f in syn.make_pyramid(cfg, seed)]
lss, radar = syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV)
metas = syn.make_img_metas(cfg)
tr = head.transformer
qb, qf = (x.to(DEV) for x in syn.make_queries(cfg, seed))
STAGES = ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling", "mixing", "ffn")

import racformer_amd.transformer as T
REC = []
def wrap(name):
    orig = getattr(T, name)
    def f(*a, **k):
        out = orig(*a, **k)
        if name == "generator_fused" and out.shape[-1] == 2192:
            REC.append((name + "(wide)", out[:, :2189]))
            return out
        REC.append((name, out))      # (kept alive: no clone launch that would change the timing much)
        return out
    setattr(T, name, f)
for n in ("generator_fused", "add_ln", "sasa_fused", "regroup_pyramid"):
    wrap(n)
orig_prepare = T.RaCFormerTransformerDecoderLayer.prepare
def prep(self, *a, **k):
    out = orig_prepare(self, *a, **k)
    REC.append(("radar_value", out["radar_value"])); REC.append(("lss_value", out["lss_value"]))
    return out
T.RaCFormerTransformerDecoderLayer.prepare = prep

def fwd():
    st = []
    del REC[:]
    with torch.no_grad(), scratch_namespace("eager_probe"):
        cls, box = tr(qb, qf, list(feats), lss, radar, None, [dict(m) for m in metas], stages_per_layer=st)
    return st, cls, box, list(REC)

def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []

solo, scls, sbox, srec = fwd(); torch.cuda.synchronize()
solo = [{k: v.clone() for k, v in s.items()} for s in solo]
srec = [(n, [t.clone() for t in flat(o)]) for n, o in srec]
st2, c2, b2, _ = fwd(); torch.cuda.synchronize()
print("eager repeat equal:", torch.equal(c2, scls) and torch.equal(b2, sbox))
noise = CapturedStep(head, feats, lss, radar, metas, own_scratch=True)
sb = torch.cuda.Stream()
first = {}
for it in range(40):
    with torch.cuda.stream(sb):
        for _ in range(3):
            noise.replay()
    st, cls, box, rec = fwd()
    torch.cuda.synchronize()
    for j, ((n, o), (n2, so)) in enumerate(zip(rec, srec)):
        bad = [i for i, (a, b) in enumerate(zip(flat(o), so)) if a.shape == b.shape and not torch.equal(a, b)]
        if bad:
            a, b = flat(o)[bad[0]], so[bad[0]]
            print("iter", it, "first deviating recorded op: #%d %s (tensor %d, shape %s) max diff %.2e, %d elements differ" %
                  (j, n, bad[0], tuple(a.shape), (a.float() - b.float()).abs().max().item(), int((a != b).sum())))
            break
    for l in range(6):
        hit = None
        for s in STAGES:
            if not torch.equal(st[l][s], solo[l][s]):
                hit = (l, s, (st[l][s] - solo[l][s]).abs().amax().item())
                break
        if hit:
            first[hit[:2]] = first.get(hit[:2], 0) + 1
            print("iter", it, "first deviation: layer", hit[0], "stage", hit[1], "max diff %.2e" % hit[2])
            break
print("first-deviation histogram:", first)
