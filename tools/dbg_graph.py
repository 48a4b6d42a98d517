import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from racformer_amd import synthetic as syn
from racformer_amd.graph import CapturedStep
from test_parity_gpu import build_head
DEV = "cuda:0"
cfg = syn.F8
g = np.load(os.path.join(ROOT, "tests/golden/head_f8.npz"))
head = build_head(cfg, g, int(g["seed"]), int(g["weight_seed"]))
seed = int(g["seed"])
feats = [f.to(DEV) for f in syn.make_pyramid(cfg, seed)]
lss, radar = syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV)
metas = syn.make_img_metas(cfg)
other = [dict(m) for m in metas]
for m in other:
    m["img_timestamp"] = [10.0 - 0.4 * (i // cfg.num_cams) + 0.001 * (i % cfg.num_cams) for i in range(len(m["img_timestamp"]))]
def eager(ms):
    with torch.no_grad():
        p = head(list(feats), lss, radar, [dict(m) for m in ms])
    return p["all_bbox_preds"].clone()
e1, e2 = eager(metas), eager(other)
e1b = eager(metas)
print("eager repeat equal:", torch.equal(e1, e1b), "eager metas vs other max diff", (e1 - e2).abs().amax().item())
cap = CapturedStep(head, feats, lss, radar, metas)
p, _ = cap.replay(); torch.cuda.synchronize()
print("replay same metas == eager:", torch.equal(p["all_bbox_preds"], e1))
p, _ = cap.replay(img_metas=metas); torch.cuda.synchronize()
print("replay restaged same metas == eager:", torch.equal(p["all_bbox_preds"], e1), (p["all_bbox_preds"] - e1).abs().amax().item())
p, _ = cap.replay(img_metas=other); torch.cuda.synchronize()
d = (p["all_bbox_preds"] - e2).abs()
print("replay other metas vs eager other: max", d.amax().item(), "per layer", [d[l].amax().item() for l in range(6)], "per column L0", d[0].amax(dim=(0, 1)).tolist())
m0 = cap.metas[0]
fresh = [dict(m) for m in other]
head.transformer.decoder.stage_metas(fresh, 1, torch.device(DEV))
for k in ("time_diff", "time_diff_safe", "lidar2img"):
    print(k, "staged equal:", torch.equal(m0[k], fresh[0][k]), m0[k].flatten()[:4].tolist(), fresh[0][k].flatten()[:4].tolist())
e2b = eager(other)
print("eager other repeat equal:", torch.equal(e2, e2b))
p, _ = cap.replay(img_metas=metas); torch.cuda.synchronize()
print("replay back to metas == eager:", torch.equal(p["all_bbox_preds"], e1), (p["all_bbox_preds"] - e1).abs().amax().item())
outs = []
for i in range(4):
    p, _ = cap.replay(); torch.cuda.synchronize()
    outs.append(p["all_bbox_preds"].clone())
print("consecutive replays equal:", [torch.equal(outs[i], outs[i + 1]) for i in range(3)], "vs eager", [(o - e1).abs().amax().item() for o in outs])
