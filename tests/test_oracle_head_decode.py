"""Pins the end of the path in the CPU oracle -- RaCFormer_head.forward (inference branch), _init_layers /
generate_points, NMSFreeCoder.decode_single, get_bboxes -- and more seeds of the full f8 decoder against fixtures made by
the reference's own files (tests/golden/gen_golden.py: decode_cases.npz, head_*.npz, decoder_f8_s*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from parity import (decode_parity, decoder_parity, detections_parity, head_boxes_normalised, oracle_decoder,
                    run_with_reference_views, teacher_forced_layer_check)
from racformer_amd import synthetic as syn

POST_RANGE = (-61.2, -61.2, -10.0, 61.2, 61.2, 10.0)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("case,tag,thr", [("A", "thr", 0.05), ("A", "none", None), ("A", "zero", 0.0), ("B", "thr", 0.05),
                                          ("C", "thr", 0.05)])
def test_nms_free_decode_cases(golden_dir, case, tag, thr):
    g = load(golden_dir, "decode_cases.npz")
    cls, box, K = t(g[f"{case}_cls"]), t(g[f"{case}_box"]), int(g[f"{case}_K"])
    for z_bottom, (kb, ks, kl) in ((False, ("bboxes", "scores", "labels")), (True, ("get_bboxes", "get_scores", "get_labels"))):
        got = R.nms_free_decode(cls, box, K, cls.shape[1], thr, POST_RANGE, z_bottom=z_bottom)
        ref = dict(bboxes=g[f"{case}_{tag}_{kb}"], scores=g[f"{case}_{tag}_{ks}"], labels=g[f"{case}_{tag}_{kl}"])
        info = decode_parity(got, ref, cls, box, K, cls.shape[1], z_bottom=z_bottom, what=f"{case}/{tag}")
        assert info["n"] == len(ref["scores"])
    if case == "A":
        n = {"thr": 7, "none": None, "zero": None}[tag]      # 10 clear ranks - 2 masked centres - ... (sanity of the fixture)
        assert n is None or len(ref["scores"]) <= 10


def test_decode_fixture_exercises_the_masks(golden_dir):
    """The crafted case really contains what it claims: rows removed by the centre range, by the score threshold, a tie
    group inside the top-K and the 0.0-threshold quirk (mask computed, not applied)."""
    g = load(golden_dir, "decode_cases.npz")
    assert len(g["A_none_scores"]) > len(g["A_thr_scores"])            # the threshold removes rows ...
    assert len(g["A_zero_scores"]) == len(g["A_none_scores"])          # ... but 0.0 does not (truthiness, :69)
    assert len(g["A_none_scores"]) < int(g["A_K"])                     # centres outside the range were dropped
    s = g["A_thr_scores"]
    assert int((s[1:] == s[:-1]).sum()) >= 2                           # three-way tie inside the top-K
    assert np.any(g["A_thr_get_bboxes"][:, 0] == np.float32(61.2))     # centre exactly on the limit is kept
    sb = g["B_thr_scores"]
    assert int((sb == 1.0).sum()) >= 2 and int((sb[1:] == sb[:-1]).sum()) > 50


def test_head_init_query(golden_dir):
    for name, cfg in (("head_small6.npz", syn.SMALL6), ("head_f8.npz", syn.F8)):
        g = load(golden_dir, name)
        w = R.head_init_query(cfg.num_query, cfg.num_clusters)
        cols = g["init_query_cols"].tolist()
        assert torch.equal(w[:, cols], t(g["init_query_fixed"])), name
        assert torch.equal(w[:, :2], t(g["generate_points"])), name


def run_head(cfg, g, what):
    """oracle head_forward with the fixture's camera choices -> (outputs, head state dict)."""
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr_sd = syn.make_state_dict(cfg, wseed)
    head_sd = {"init_query_bbox.weight": syn.make_queries(cfg, seed)[0][0], "label_enc.weight": t(g["label_enc"])}

    def run(force):
        R.LOC_TAP = []
        R.VIEW_FORCE = [np.asarray(v) for v in force] if force is not None else None
        try:
            with torch.no_grad():
                out = R.head_forward(head_sd, tr_sd, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                     syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg)
            views = torch.stack([R.views_of(l, cfg.num_cams) for l in R.LOC_TAP])
        finally:
            R.LOC_TAP = R.VIEW_FORCE = None
        return out, views

    (out, _), _ = run_with_reference_views(run, g["views"], what)
    return out


def check_head(cfg, g, out, what):
    """Decoder outputs (normalised space, camera choices equal), strict decode on the reference's own head outputs, and the
    end-to-end detections."""
    ref_box_n = head_boxes_normalised(g["all_bbox_preds"], cfg.pc_range)
    rows = decoder_parity(out["all_cls_scores"], head_boxes_normalised(out["all_bbox_preds"], cfg.pc_range),
                          g["all_cls_scores"], ref_box_n, what=what)
    ref_det = dict(bboxes=g["det_boxes"], scores=g["det_scores"], labels=g["det_labels"])
    rcls, rbox = t(g["all_cls_scores"])[-1, 0], t(g["all_bbox_preds"])[-1, 0]
    strict = R.nms_free_decode(rcls, rbox, 300, cfg.num_classes, 0.05, POST_RANGE)
    decode_parity(strict, ref_det, rcls, rbox, 300, cfg.num_classes, what=what + " decode(ref outputs)")
    own = R.nms_free_decode(out["all_cls_scores"][-1, 0], out["all_bbox_preds"][-1, 0], 300, cfg.num_classes, 0.05, POST_RANGE)
    info = detections_parity(own, ref_det, what=what + " end-to-end", allowed_unmatched=3 * rows[-1]["failing"])
    assert info["matched"] >= 0.97 * len(ref_det["scores"]), info


def test_head_forward_small6(golden_dir):
    g = load(golden_dir, "head_small6.npz")
    check_head(syn.SMALL6, g, run_head(syn.SMALL6, g, "head small6"), "head small6")


def test_head_forward_f8(golden_dir):
    g = load(golden_dir, "head_f8.npz")
    torch.set_num_threads(min(16, os.cpu_count()))
    check_head(syn.F8, g, run_head(syn.F8, g, "head f8"), "head f8")


@pytest.mark.parametrize("name,cfg", [("decoder_f8_s1.npz", syn.F8), ("decoder_f8_s2.npz", syn.F8), ("decoder_f8_s3.npz", syn.F8),
                                      ("decoder_f8_3cam_s1.npz", syn.F8_3CAM)])
def test_decoder_f8_more_seeds(golden_dir, name, cfg):
    """More seeds of the full-size decoder under the free-running criterion of tests/parity.py (camera choices equal)."""
    g = load(golden_dir, name)
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    (cls, box, _), _ = run_with_reference_views(
        lambda force: oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                     syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg, None, force), g["views"], name)
    decoder_parity(cls, box, g["cls"], g["box"], what=name)


def test_decoder_f8_teacher_forced(golden_dir):
    """Every layer fed the reference's own inputs (no amplification through the stack): all 900 queries within 1e-4 on
    the layer outputs, probes of every stage within 1e-4; only shown view flips are exempt."""
    g = load(golden_dir, "decoder_f8_tf.npz")
    cfg = syn.F8
    torch.set_num_threads(min(16, os.cpu_count()))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    lsd = R._sub(syn.make_state_dict(cfg, wseed), "decoder.decoder_layer.")
    metas = syn.make_img_metas(cfg)
    time_diff = R.time_diff_from_metas(metas, 1, cfg.num_cams)
    lidar2img = torch.from_numpy(np.asarray([m["lidar2img"] for m in metas]).astype(np.float32))
    feats_cl = R.regroup_pyramid(syn.make_pyramid(cfg, seed), cfg.num_cams, cfg.num_groups)
    lss, radar = syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1)
    nflips = []
    for l in range(cfg.num_layers):
        st = {}
        R.LOC_TAP = []
        try:
            with torch.no_grad():
                feat, cls, box = R.decoder_layer(lsd, t(g["in_bbox"][l]), t(g["in_feat"][l]), feats_cl, lss, radar, time_diff,
                                                 lidar2img, cfg, l, st)
            views = R.views_of(R.LOC_TAP[0], cfg.num_cams)
        finally:
            R.LOC_TAP = None
        nflips.append(teacher_forced_layer_check(l, g, cfg, feat, cls, box, st, views, what="oracle"))
    print("teacher-forced: flipped queries per layer", nflips)
    assert sum(nflips) <= 12
