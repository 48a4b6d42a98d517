#!/usr/bin/env python3
"""Route B of VERDICT r4 item 4, sized before any kernel is written (CPU, the oracle's own BEV keypoints).

Could the BEV gather serve its taps from an LDS tile instead of the texture path?  A (query, head) gathers 8 frames x 20 points x
4 taps of 256 bytes from EIGHT different maps (one per frame), so a staged tile is per (query, head, frame).  For the bench rig
(random-everything weights, seed 0) and the reference-initialised rig, all six layers, both streams, this measures
  * how compact a (query, head, frame)'s 20 points are: extent of their bounding box, and the share of the taps inside a
    (16+1)^2 and a (8+1)^2 pixel tile centred on the points' median,
  * the bytes: tile staging against the taps themselves (a 17 x 17 tile is 74 KB for 20 KB of taps),
  * how many DISTINCT pixels the 80 taps of a (query, head, frame) touch (what a perfect per-item cache would have to fetch).
Writes profiles/r05_bev_tile_locality.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import restate as R  # noqa: E402
from racformer_amd import synthetic as syn  # noqa: E402


def run(cfg, sd, seed):
    rec = []
    orig = R.bev_keypoints

    def spy(sd_, prefix, *a, **k):
        loc, sw = orig(sd_, prefix, *a, **k)
        rec.append((prefix, loc.clone()))
        return loc, sw
    R.bev_keypoints = spy
    try:
        qb, qf = syn.make_queries(cfg, seed)
        with torch.no_grad():
            R.transformer_forward(sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1),
                                  syn.make_img_metas(cfg), cfg)
    finally:
        R.bev_keypoints = orig
    return rec


def analyse(rec, H=128, W=128):
    out = []
    for i, (prefix, loc) in enumerate(rec):          # loc [B,Q,heads,T,P,2] in [0,1]
        layer = i // 2
        px = (loc[..., 0] * W - 0.5).numpy()[0]      # [Q,heads,T,P]
        py = (loc[..., 1] * H - 0.5).numpy()[0]
        x0, y0 = np.floor(px), np.floor(py)
        ext = np.maximum(px.max(-1) - px.min(-1), py.max(-1) - py.min(-1))           # per (q, head, frame)
        row = dict(layer=layer, stream=prefix.split("_")[1], extent_px_p50=float(np.median(ext)), extent_px_p90=float(np.percentile(ext, 90)),
                   extent_px_max=float(ext.max()))
        for half, name in ((8, "17x17"), (4, "9x9")):
            cx, cy = np.round(np.median(px, -1, keepdims=True)), np.round(np.median(py, -1, keepdims=True))
            ins = 0.0
            for dx in (0, 1):
                for dy in (0, 1):
                    ins += ((np.abs(x0 + dx - cx) <= half) & (np.abs(y0 + dy - cy) <= half)).mean()
            row[f"taps_inside_{name}_tile_share"] = float(ins / 4)
            full = ((np.abs(x0 - cx) <= half) & (np.abs(x0 + 1 - cx) <= half) & (np.abs(y0 - cy) <= half) & (np.abs(y0 + 1 - cy) <= half)).all(-1)
            row[f"items_entirely_inside_{name}_share"] = float(full.mean())
        # distinct pixels per (query, head, frame): the 80 taps of its 20 points
        keys = np.stack([(y0 + dy) * 4096 + (x0 + dx) for dx in (0, 1) for dy in (0, 1)], -1).reshape(*x0.shape[:-1], -1)   # [Q,h,T,80]
        srt = np.sort(keys, -1)
        distinct = 1 + (np.diff(srt, axis=-1) != 0).sum(-1)
        row["distinct_pixels_per_item_mean"] = float(distinct.mean())
        out.append(row)
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = syn.F8
    res = {"_what": __doc__.split("Writes")[0].strip(),
           "bytes": {"taps_per_item": 20 * 4 * 256, "tile_17x17": 17 * 17 * 256, "tile_9x9": 9 * 9 * 256,
                     "items_per_launch (900 queries x 4 heads x 8 frames x 2 streams)": 57600,
                     "taps_per_launch_MB": 57600 * 20 * 4 * 256 / 1e6, "tiles_17x17_per_launch_MB": 57600 * 17 * 17 * 256 / 1e6,
                     "tiles_9x9_per_launch_MB": 57600 * 9 * 9 * 256 / 1e6}}
    rigs = {"bench rig (random-everything weights, seed 0)": syn.make_state_dict(cfg, 0)}
    g = np.load(os.path.join(ROOT, "tests", "golden", "decoder_f8_init.npz"))
    from parity import load_rig_state_dict
    rigs["reference-initialised rig (decoder_f8_init)"] = load_rig_state_dict(cfg, g, os.path.join(ROOT, "tests", "golden"))
    for name, sd in rigs.items():
        rows = analyse(run(cfg, sd, 0))
        res[name] = rows
        for r in rows:
            print(name[:12], r["layer"], r["stream"], "extent p50/p90/max %.1f/%.1f/%.1f" % (r["extent_px_p50"], r["extent_px_p90"], r["extent_px_max"]),
                  "inside17 %.3f all-in %.3f | inside9 %.3f all-in %.3f | distinct px %.1f" % (
                      r["taps_inside_17x17_tile_share"], r["items_entirely_inside_17x17_share"], r["taps_inside_9x9_tile_share"],
                      r["items_entirely_inside_9x9_share"], r["distinct_pixels_per_item_mean"]), flush=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r05_bev_tile_locality.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
