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
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
layer = head.transformer.decoder.decoder_layer
if mode == "nomiopen_gru":
    layer.sampling_radar_bev.temporal_encoder.fused_conv = False
single = CapturedStep(head, feats, lss, radar, metas)
p, d = single.replay(); torch.cuda.synchronize()
want_c, want_b = p["all_cls_scores"].clone(), p["all_bbox_preds"].clone()
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lanes = [(CapturedStep(head, feats, lss, radar, metas, own_scratch=True), torch.cuda.Stream()) for _ in range(nl)]
main = torch.cuda.current_stream()
bad = 0
for rnd in range(30):
    got = []
    for i, (cap, st) in enumerate(lanes):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            p, d = cap.replay(img_metas=metas if mode != "nostage" else None)
        got.append(p)
        if mode == "serial":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for i, p in enumerate(got):
        if not torch.equal(p["all_bbox_preds"], want_b):
            bad += 1
            e = [(p["all_bbox_preds"][l] - want_b[l]).abs().amax().item() for l in range(6)]
            print("round", rnd, "lane", i, "per-layer max diff", ["%.2e" % x for x in e], "nan", bool(torch.isnan(p["all_bbox_preds"]).any()))
print(mode, "mismatching (round, lane) pairs:", bad, "of", 30 * nl)
