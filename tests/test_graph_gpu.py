"""The captured plan (racformer_amd/graph.py: regroup + prologue + six layers [+ decode] as one HIP-graph submission) against
the reference's fixtures and the oracle -- the same criteria as the eager plan, through the replayed graph."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from parity import ARGMAX_MARGIN_INIT_RIG, MAX_FLIPPED_POINTS, decoder_parity, detections_parity, fill_rig_module, flipped_points, head_boxes_normalised, kept_rows
from racformer_amd import synthetic as syn
from racformer_amd.graph import CapturedForward, CapturedStep
from racformer_amd.transformer import RaCFormerTransformer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,cfg,literal", [("decoder_f8_init.npz", syn.F8, True), ("decoder_f8_3cam_init.npz", syn.F8_3CAM, True),
                                               ("decoder_f8.npz", syn.F8, False)])
def test_captured_decoder_vs_reference(golden_dir, name, cfg, literal):
    """Six free-running layers replayed from the graph (twice: the second replay runs on the first one's buffers) against the
    reference's CPU forward; the camera choices come out of the replayed kernels' own loc_out."""
    g = np.load(os.path.join(golden_dir, name))
    seed = int(g["seed"])
    tr = fill_rig_module(RaCFormerTransformer(**cfg.transformer_kwargs()).eval(), cfg, g, golden_dir).to(DEV)
    layer = tr.decoder.decoder_layer
    qb, qf = (x.to(DEV) for x in syn.make_queries(cfg, seed))
    feats = [f.to(DEV) for f in syn.make_pyramid(cfg, seed)]
    lss, radar = syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV)
    metas = syn.make_img_metas(cfg)
    tr.decoder.stage_metas(metas, 1, torch.device(DEV))
    layer.sampling.force_views = [t(v).to(DEV).contiguous() for v in g["views"]]      # equal discrete choices on both sides,
    layer.sampling.force_views_cyclic = True                                          # in the warm-up forwards and the captured one
    taps = []

    def fn():
        layer.sampling.capture_loc = taps
        del taps[:]
        return tr(qb, qf, list(feats), lss, radar, None, metas)

    cap = CapturedForward(fn, torch.device(DEV))
    for _ in range(2):
        cls, box = cap.replay()
    torch.cuda.synchronize()
    assert len(taps) == cfg.num_layers
    views = torch.stack([R.views_of(l.cpu(), cfg.num_cams) for l in taps])
    nflip = flipped_points(views, g["views"])
    print(name, "captured plan: own camera choices differing from the reference's, per layer:", nflip)
    assert sum(nflip) <= MAX_FLIPPED_POINTS
    decoder_parity(cls.cpu(), box.cpu(), g["cls"], g["box"], what=name + " (captured)", **(dict(tail_budget=None, argmax_margin=ARGMAX_MARGIN_INIT_RIG) if literal else {}))


def test_captured_step_new_metas_and_detections(golden_dir):
    """CapturedStep (head + fixed-shape decode): replayed with the metas of the sample it was captured on it reproduces the
    eager step bit for bit; replayed with ANOTHER sample's timestamps (staged in front of the graph) it follows an eager
    forward on those metas -- the per-sample host arithmetic is outside the graph, not frozen into it -- and the end-to-end
    detection list matches the reference head's (head_f8.npz)."""
    from test_parity_gpu import build_head
    cfg = syn.F8
    g = np.load(os.path.join(golden_dir, "head_f8.npz"))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    head = build_head(cfg, g, seed, wseed)
    feats = [f.to(DEV) for f in syn.make_pyramid(cfg, seed)]
    lss, radar = syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV)
    metas = syn.make_img_metas(cfg)
    with torch.no_grad():
        eager = head(list(feats), lss, radar, [dict(m) for m in metas])
        eager_det = head.get_detections_fixed(eager).clone()
        eager_cls, eager_box = eager["all_cls_scores"].clone(), eager["all_bbox_preds"].clone()
    cap = CapturedStep(head, feats, lss, radar, metas)
    preds, det = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(preds["all_cls_scores"], eager_cls) and torch.equal(preds["all_bbox_preds"], eager_box)
    assert torch.equal(det, eager_det)
    # against the reference head: a second capture with the reference's camera choices imposed (the comparison of
    # tests/test_parity_gpu.py::test_head_forward_and_detections_vs_reference, through the replayed graph)
    smp = head.transformer.decoder.decoder_layer.sampling
    smp.force_views, smp.force_views_cyclic = [t(v).to(DEV).contiguous() for v in g["views"]], True
    preds, det = CapturedStep(head, feats, lss, radar, metas).replay()
    torch.cuda.synchronize()
    smp.force_views, smp.force_views_cyclic = None, False
    ref_det = dict(bboxes=g["det_boxes"], scores=g["det_scores"], labels=g["det_labels"])
    ref_n = head_boxes_normalised(g["all_bbox_preds"], cfg.pc_range)
    rows = decoder_parity(preds["all_cls_scores"].cpu(), head_boxes_normalised(preds["all_bbox_preds"].cpu(), cfg.pc_range),
                          g["all_cls_scores"], ref_n, what="head_f8 (captured)")
    detections_parity(kept_rows(det[0].cpu()), ref_det, what="head_f8 captured end-to-end", allowed_unmatched=3 * rows[-1]["failing"])
    # another sample's metas (first capture, free-running): frame spacing 0.4 s instead of 0.5 s
    other = [dict(m) for m in metas]
    for m in other:
        m["img_timestamp"] = [10.0 - 0.4 * (i // cfg.num_cams) + 0.001 * (i % cfg.num_cams) for i in range(len(m["img_timestamp"]))]
    preds2, det2 = cap.replay(img_metas=other)
    torch.cuda.synchronize()
    with torch.no_grad():
        eager2 = head(list(feats), lss, radar, [dict(m) for m in other])
    assert not torch.equal(preds2["all_bbox_preds"], eager_box)
    assert torch.equal(preds2["all_cls_scores"], eager2["all_cls_scores"]) and torch.equal(preds2["all_bbox_preds"], eager2["all_bbox_preds"])


def test_plans_in_flight_side_by_side_match_the_single_plan(golden_dir):
    """Several samples in flight (bench.py --in-flight N): captured plans with scratch AND INPUTS of their own -- every lane holds
    a different sample: pyramid, BEV stacks, metas (round 4; round 3's lanes all read one sample's buffers) -- replayed on streams
    of their own, interleaved and overlapping.  Each lane reproduces, bit for bit, what ONE plan alone computes on that lane's
    sample: the eager step with nothing else on the GPU (a shared scratch buffer, a lane reading another lane's inputs or any
    other cross-plan state would show here)."""
    from test_parity_gpu import build_head
    cfg = syn.F8
    g = np.load(os.path.join(golden_dir, "head_f8.npz"))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    head = build_head(cfg, g, seed, wseed)
    n_lanes = 3
    samples = []
    for i in range(n_lanes):
        feats = [f.to(DEV) for f in syn.make_pyramid(cfg, seed + i)]
        lss, radar = syn.make_bev(cfg, seed + i, 0).to(DEV), syn.make_bev(cfg, seed + i, 1).to(DEV)
        metas = syn.make_img_metas(cfg, sample=i)
        other = syn.make_img_metas(cfg, sample=i + n_lanes)         # a second set of metas for the same buffers (restaged per replay)
        samples.append((feats, lss, radar, (metas, other)))
    want = []
    with torch.no_grad():
        for feats, lss, radar, both in samples:
            row = []
            for ms in both:
                p = head(list(feats), lss, radar, [dict(m) for m in ms])
                d = head.get_detections_fixed(p)
                torch.cuda.synchronize()
                row.append((p["all_bbox_preds"].clone(), d.clone()))
            assert not torch.equal(row[0][0], row[1][0])
            want.append(row)
    assert not torch.equal(want[0][0][0], want[1][0][0]) and not torch.equal(want[1][0][0], want[2][0][0])
    lanes = [(CapturedStep(head, f, l, r, both[0], own_scratch=True), torch.cuda.Stream()) for f, l, r, both in samples]
    main = torch.cuda.current_stream()
    for rnd in range(4):
        got = []
        for i, (cap, st) in enumerate(lanes):
            st.wait_stream(main)
            which = (i + rnd) % 2
            with torch.cuda.stream(st):
                p, d = cap.replay(img_metas=samples[i][3][which])
            got.append((i, which, p, d))
        torch.cuda.synchronize()
        for i, which, p, d in got:
            assert torch.equal(p["all_bbox_preds"], want[i][which][0]) and torch.equal(d, want[i][which][1]), (rnd, i, which)
    for cap, _ in lanes:
        cap.close()
