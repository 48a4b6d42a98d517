#!/usr/bin/env python3
"""How far two fp32 CPU implementations of the same arithmetic drift apart over the six free-running layers: the oracle
(oracle/restate.py) against the reference's own CPU forward (the committed fixtures), per fixture and layer.  This is the
measurement the tail budget of tests/parity.py is derived from (run in the build container: ~2 minutes):

    python tools/measure_cpu_vs_cpu.py > profiles/r03_cpu_vs_cpu_drift.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import restate as R                      # noqa: E402
from parity import oracle_decoder, flipped_points, load_rig_state_dict   # noqa: E402
from racformer_amd import synthetic as syn           # noqa: E402

FIX = [("decoder_f8.npz", syn.F8), ("decoder_f8_s1.npz", syn.F8), ("decoder_f8_s2.npz", syn.F8), ("decoder_f8_s3.npz", syn.F8),
       ("decoder_f8_3cam.npz", syn.F8_3CAM), ("decoder_f8_3cam_s1.npz", syn.F8_3CAM),
       ("decoder_f8_init.npz", syn.F8), ("decoder_f8_3cam_init.npz", syn.F8_3CAM)]


def main():
    torch.set_num_threads(os.cpu_count())
    out = {}
    for name, cfg in FIX:
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        seed, wseed = int(g["seed"]), int(g["weight_seed"])
        sd = load_rig_state_dict(cfg, g, os.path.join(ROOT, "tests", "golden"))
        qb, qf = syn.make_queries(cfg, seed)
        args = (R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1),
                syn.make_img_metas(cfg), cfg)
        cls, box, views = oracle_decoder(*args)
        nflip = flipped_points(views, g["views"])
        if sum(nflip):
            cls, box, views = oracle_decoder(*args, force_views=g["views"])
        gc, gb = torch.from_numpy(g["cls"]).double(), torch.from_numpy(g["box"]).double()
        rows = []
        for l in range(cls.shape[0]):
            eb = (box[l].double() - gb[l]).abs().amax(-1).reshape(-1)
            ec = (cls[l].double() - gc[l]).abs().amax(-1).reshape(-1)
            top2 = gc[l].reshape(-1, gc.shape[-1]).topk(2, -1).values
            margin = (top2[:, 0] - top2[:, 1])
            mism = (cls[l].argmax(-1) != gc[l].argmax(-1)).reshape(-1)
            rows.append(dict(layer=l, box_max=float(eb.max()), box_p50=float(eb.median()), over_1e3=int((eb > 1e-3).sum()),
                             cls_max=float(ec.max()), cls_p50=float(ec.median()), argmax_mismatch=int(mism.sum()),
                             margin_of_mismatches=[float(m) for m in margin[mism]], min_margin=float(margin.min())))
        out[name] = dict(differing_views_free_run=nflip, layers=rows)
        print(name, [(r["layer"], f"{r['box_max']:.1e}", r["over_1e3"], f"{r['cls_max']:.1e}", r["argmax_mismatch"]) for r in rows],
              file=sys.stderr)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
