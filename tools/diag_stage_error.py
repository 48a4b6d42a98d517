#!/usr/bin/env python3
"""Diagnostic (GPU box): per-stage error of one teacher-forced decoder layer against the reference's fixture
(tests/golden/decoder_f8_tf.npz) for each execution plan, beside the oracle's own error -- shows which stage injects how
much rounding noise.  usage: python tools/diag_stage_error.py [layers...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plans  # noqa: E402,F401  (registers the alternate execution plans)
from oracle import restate as R  # noqa: E402
from racformer_amd import synthetic as syn  # noqa: E402
from racformer_amd.transformer import RaCFormerTransformer, regroup_pyramid  # noqa: E402

DEV = "cuda:0"
STAGES = ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling", "mixing", "ffn")
g = np.load(os.path.join(ROOT, "tests", "golden", "decoder_f8_tf.npz"))
cfg = syn.F8
seed, wseed = int(g["seed"]), int(g["weight_seed"])
layers = [int(a) for a in sys.argv[1:]] or [0, 3, 5]
t = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731
pq, ps = t(g["probe_q"]), t(g["probe_q_sampling"])


def report(tag, l, feat, cls, box, st):
    ref_feat = t(g["in_feat"][l + 1]) if l + 1 < 6 else t(g["out_feat_last"])
    parts = []
    for s in STAGES:
        sel = ps if s == "sampling" else pq
        e = (st[s].cpu()[:, sel] - t(g["stage_" + s][l])).abs()
        parts.append(f"{s[:12]} {e.max():.1e}/{e.median():.1e}")
    e1, e2, e3 = (feat.cpu() - ref_feat).abs(), (cls.cpu() - t(g["out_cls"][l])).abs(), (box.cpu() - t(g["out_box"][l])).abs()
    print(f"[{tag:28s}] L{l} " + " | ".join(parts) + f" || feat {e1.max():.1e}/{e1.median():.1e} cls {e2.max():.1e} box {e3.max():.1e}",
          flush=True)


# oracle
torch.set_num_threads(16)
lsd = R._sub(syn.make_state_dict(cfg, wseed), "decoder.decoder_layer.")
metas = syn.make_img_metas(cfg)
td = R.time_diff_from_metas(metas, 1, cfg.num_cams)
l2i = torch.from_numpy(np.asarray([m["lidar2img"] for m in metas]).astype(np.float32))
feats_cl = R.regroup_pyramid(syn.make_pyramid(cfg, seed), cfg.num_cams, cfg.num_groups)
lss, radar = syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1)
for l in layers:
    st = {}
    with torch.no_grad():
        feat, cls, box = R.decoder_layer(lsd, t(g["in_bbox"][l]), t(g["in_feat"][l]), feats_cl, lss, radar, td, l2i, cfg, l, st)
    report("oracle (CPU fp32)", l, feat, cls, box, st)

plans = [("default", {}), ("rowgemm=False", dict(rowgemm=False)), ("split_gemm=False", dict(split_gemm=False)),
         ("split_gemm=False,rowgemm=False", dict(split_gemm=False, rowgemm=False)), ("fused=False", dict(fused=False))]
for tag, flags in plans:
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    layer = tr.decoder.decoder_layer
    for k, v in flags.items():
        setattr(layer, k, v)
    tr = tr.to(DEV)
    metas = syn.make_img_metas(cfg)
    tr.decoder.stage_metas(metas, 1, torch.device(DEV))
    feats = regroup_pyramid([f.to(DEV) for f in syn.make_pyramid(cfg, seed)], cfg.num_cams)
    lg, rg = lss.to(DEV), radar.to(DEV)
    with torch.no_grad():
        prepared = layer.prepare(lg, rg)
        for l in layers:
            st = {}
            layer._carry = None
            feat, cls, box = layer(t(g["in_bbox"][l]).to(DEV), t(g["in_feat"][l]).to(DEV), feats, lg, rg, None, metas, layer=l,
                                   prepared=prepared, stages=st)
            torch.cuda.synchronize()
            report(tag, l, feat, cls, box, st)
    del tr, feats
