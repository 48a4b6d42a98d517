"""Shared parity criteria (north_star: box regressions within 1e-3, class argmax bit-exact against the reference CPU path).

What is compared, and how strictly
----------------------------------
1. TEACHER-FORCED, per decoder layer (``teacher_forced_layer_check``): layer l is fed the reference's own
   (query_bbox, query_feat) of layer l (fixture decoder_f8_tf.npz).  No error enters from earlier layers, so the criterion
   is literal and ten times tighter than north_star's: ALL queries within 1e-4 on the layer's outputs, probes of every
   stage within 1e-4, no allowance.
2. FREE-RUNNING, six layers (``decoder_parity``).  Two things stand between two independent float32 implementations and a
   literal 1e-3 on every query here, and both are measured, not assumed:
   * the path has ONE discontinuous step, first-valid-view selection (models/sparsebev_sampling.py:89-110): a sampling
     point whose projection lies within rounding of an image border is assigned to another camera by any implementation
     whose libm differs in the last bit (the reference's own CUDA and CPU paths included).  Measured: 0-3 of the 2.07 M
     points of a forward differ; ONE such point moves its query by ~1.5e-2 and its self-attention neighbours by ~2e-3 a
     layer later.  The tests therefore (a) count the differing points -- the kernel reports its choices (``loc_out``), the
     fixtures hold the reference's, read at its msmv operator boundary -- and bound them, and (b) re-run the comparison
     with the REFERENCE'S choices imposed on the product (``view_in`` of rac_sampling4d_fwd: the only thing it changes is
     which camera those few points are sampled in).  With the discrete choices equal, the decoder is a continuous
     function of its inputs and the criterion below applies to every query: nothing is "attributed" by heuristics.
   * the random-everything rig (every Linear N(0, 1/fan_in), offsets / mixing parameters / tau driven by the features --
     SURVEY 8d's first rig) amplifies float32 rounding 4-5x per layer.  Two fp32 CPU implementations of the same
     arithmetic -- this repository's oracle and the reference's own files, equal camera choices -- drift apart over the
     six layers by what tools/measure_cpu_vs_cpu.py measures (profiles/r03_cpu_vs_cpu_drift.json, all eight f8
     fixtures): box error at most 2.8e-6, 1.7e-5, 7.8e-5, 9.4e-5, 1.2e-3, 2.3e-3 and class-logit error at most 4.8e-6,
     2.7e-5, 8.8e-5, 2.3e-4, 3.9e-3, 1.2e-2 in layers 0..5: even CPU against CPU, 1-2 of 900 queries miss 1e-3 in the last
     two layers.  The criterion on THAT rig therefore is, separately for the two halves of north_star's statement:
       - class argmax: identical for every query, except a query whose two leading REFERENCE logits are closer together
         than ``ARGMAX_MARGIN[layer]`` = the CPU-vs-CPU logit drift measured for that layer (a tie at the resolution of the
         arithmetic: neither implementation's argmax is "the" answer there; since round 5 at most 1.5e-3 in layers 4-5, ten times
         the largest margin ever used); any other mismatch fails the test;
       - boxes: layers 0-2 within 1e-3 for every query; in layers 3, 4, 5 at most ``TAIL_QUERIES`` = 2, 3, 11 of the 900
         queries may miss 1e-3, none by more than ``TAIL_TOL`` = 2e-2; p50 <= 1e-4 everywhere.
   * SURVEY 8d's SECOND rig -- weights as torch's constructors draw them, then the reference's own init_weights()
     (zero offset / generator / tau weights, xavier value / output / fusion Linears; fixtures decoder_f8_init.npz,
     decoder_f8_3cam_init.npz) -- is the model as the reference initialises it and does not amplify (CPU vs CPU: 2.7e-5 /
     4.2e-5 in layer 5).  There the criterion is north_star's, LITERALLY: every query of every one of the six
     free-running layers within 1e-3, argmax identical (subject only to the tie rule above), no tail.  Reduced
     configurations (30 queries) and smoke() are literal too.
   Every comparison records what it actually used (differing / imposed camera choices, box misses, argmax mismatches and
   their margins, per layer) in ``USED``; the GPU session writes it to gpurun_out/parity_budget_used.json, and the committed
   copy is profiles/r04_parity_budget_used.json (keyed by session kind: a CPU session writes its own file).
3. The NMS-free decode is positional and exact (``decode_parity``); the end-to-end detection list is matched one to one
   (``detections_parity``).
Boxes are compared in the decoder's normalised output space (xyz / pc_range span, log sizes, sin, cos, velocity).
"""
import numpy as np
import torch

# FROZEN at their round-4 values since round 5 (no longer re-derived from the product's own measurements).  What justifies a tail on
# this rig is now an INDEPENDENT arbiter: the oracle evaluated in float64 (tools/fp64_arbiter.py, profiles/r05_fp64_arbiter.json).
# Against it the reference's own fp32 CPU forward misses 1e-3 on 1 / 5, 0 / 3 and 1 / 10 queries in layers 4 / 5 of the three chaotic
# seeds (worst query 5.9e-3), the GPU on 0 / 4, 0 / 3 and 2 / 10 (worst 9.0e-3); the GPU's median error is 0.87-1.05 x the reference's
# on every fixture and layer (tests/test_fp64_arbiter_gpu.py asserts <= 1.5 x).  11 = the reference's own 10 + 1.
TAIL_QUERIES = (0, 0, 0, 2, 3, 11)                    # random-everything rig, per layer: queries (of 900) that may miss 1e-3 on the box
                                                      # (round 4: measured maximum over every comparison of the GPU session + 1 -- 1 / 2 / 10,
                                                      #  profiles/r04_parity_budget_used.json; round 3 allowed 2 / 4 / 12)
TAIL_BUDGET = tuple(q / 900.0 for q in TAIL_QUERIES)  # ... as a fraction of the queries
TAIL_TOL = 2e-2                                       # ... and by how much at most (box space; measured maximum 9.9e-3)
# measured CPU-vs-CPU logit drift per layer (see above) for layers 0-3; layers 4-5 were 3.9e-3 / 1.2e-2 (that drift's maximum) until
# round 4 and are now 1.5e-3 = 10 x the largest margin any comparison has ever used (1.5e-4, profiles/r04_parity_budget_used.json)
ARGMAX_MARGIN = (4.8e-6, 2.7e-5, 8.8e-5, 2.3e-4, 1.5e-3, 1.5e-3)
# The reference-initialised rig (decoder_f8*_init.npz): its class logits are nearly constant over the queries (the reference's own
# init leaves the generator / offset / tau weights at zero), so a "tie" has to be judged at THAT rig's arithmetic resolution: the
# CPU-vs-CPU logit drift measured there is 4.2e-5 in layer 5 (profiles/r03_cpu_vs_cpu_drift.json); x2 as the margin, every layer.
ARGMAX_MARGIN_INIT_RIG = (1e-4,) * 6
MAX_FLIPPED_POINTS = 8                                # differing camera choices per forward on the equalised trajectory
                                                      # (of ~2.07 M points at f8; measured maximum 6, + 2: the library convolutions of the
                                                      #  ConvGRU branch may pick another algorithm on another box)
USED = []                                             # one record per comparison: what of the allowances it actually used


def _rows(cls, box, gcls, gbox):
    cls, box = torch.as_tensor(cls).double().cpu(), torch.as_tensor(box).double().cpu()
    gcls, gbox = torch.as_tensor(gcls).double(), torch.as_tensor(gbox).double()
    assert cls.shape == gcls.shape and box.shape == gbox.shape
    rows = []
    for l in range(cls.shape[0]):
        eb = (box[l] - gbox[l]).abs().amax(-1).reshape(-1)
        ec = (cls[l] - gcls[l]).abs().amax(-1).reshape(-1)
        mism = (cls[l].argmax(-1) != gcls[l].argmax(-1)).reshape(-1)
        rows.append(dict(layer=l, eb=eb, ec=ec, mism=mism))
    return rows


def _fmt(what, r, extra=""):
    eb, ec = r["eb"], r["ec"]
    return (f"{what} L{r['layer']}: box p50 {eb.median():.1e} p99 {eb.quantile(0.99):.1e} max {eb.max():.1e} | cls p50 "
            f"{ec.median():.1e} max {ec.max():.1e} | argmax mismatch {int(r['mism'].sum())}{extra}")


def flipped_points(views_a, views_b):
    """views_* [layers, S, Q, P] camera indices -> number of sampling points per layer whose selected camera differs."""
    va, vb = torch.as_tensor(np.asarray(views_a)).long(), torch.as_tensor(np.asarray(views_b)).long()
    assert va.shape == vb.shape, (va.shape, vb.shape)
    return (va != vb).flatten(1).sum(1).tolist()


def flipped_queries(views_a, views_b, num_frames, num_groups):
    """views_* [layers, S, Q, P] camera indices (S = B*T*G slots) -> bool [layers, B, Q]: the query has at least one
    sampling point in that layer whose selected camera differs."""
    va, vb = torch.as_tensor(np.asarray(views_a)).long(), torch.as_tensor(np.asarray(views_b)).long()
    assert va.shape == vb.shape, (va.shape, vb.shape)
    L, S, Q, P = va.shape
    B = S // (num_frames * num_groups)
    diff = (va != vb).view(L, B, num_frames * num_groups, Q, P)
    return diff.any(-1).any(2)


def decoder_parity(cls, box, gcls, gbox, box_tol=1e-3, what="", tail_budget=TAIL_BUDGET, tail_tol=TAIL_TOL,
                   argmax_margin=ARGMAX_MARGIN):
    """cls/box [layers,B,Q,.] of a free-running decoder (camera choices equal on both sides) vs the comparand's: module
    docstring, point 2.  ``tail_budget=None``: boxes literal for every query of every layer.  Class argmax is never part of
    the tail: a mismatch passes only as a tie of the comparand's two leading logits (margin < ``argmax_margin[layer]``)."""
    rows = _rows(cls, box, gcls, gbox)
    gc = torch.as_tensor(gcls).double()
    msgs, bad, used = [], [], []
    for r in rows:
        l = r["layer"]
        top2 = gc[l].reshape(-1, gc.shape[-1]).topk(2, -1).values
        margin = top2[:, 0] - top2[:, 1]
        lim = argmax_margin[min(l, len(argmax_margin) - 1)]
        hard = r["mism"] & (margin >= lim)                 # argmax differs although the comparand's decision is clear
        miss = r["eb"] > box_tol
        n = miss.numel()
        frac = 0.0 if tail_budget is None else tail_budget[min(l, len(tail_budget) - 1)]
        allowed = int(round(frac * n))
        beyond = int((r["eb"] > tail_tol).sum())
        r.update(failing=int(miss.sum()), allowed=allowed, beyond=beyond, n=n, argmax_hard=int(hard.sum()),
                 argmax_ties=int((r["mism"] & ~hard).sum()))
        used.append(dict(layer=l, box_misses=r["failing"], box_budget=allowed, box_max=float(r["eb"].max()),
                         box_p50=float(r["eb"].median()), cls_max=float(r["ec"].max()), argmax_mismatches=int(r["mism"].sum()),
                         argmax_mismatch_margins=[float(m) for m in margin[r["mism"]]], argmax_margin_limit=lim))
        msgs.append(_fmt(what, r, f" (of them ties below {lim:g}: {r['argmax_ties']}) | boxes over {box_tol:g}: {r['failing']}/{n} "
                                  f"(budget {allowed}), beyond {tail_tol:g}: {beyond}"))
        if r["failing"] > allowed or beyond or r["argmax_hard"] or float(r["eb"].median()) > box_tol / 10:
            bad.append(l)
    msg = "\n".join(msgs)
    print(msg)
    USED.append(dict(what=what, kind="decoder_parity", literal=tail_budget is None, layers=used, passed=not bad))
    assert not bad, f"decoder parity fails in layers {bad}:\n{msg}"
    return rows


def decode_parity(got, ref, cls_logits, bbox_preds, max_num, num_classes=10, z_bottom=True, box_tol=1e-5,
                  score_tol=1e-6, what=""):
    """NMS-free decode outputs (kept rows in rank order): ``got`` / ``ref`` = dict(bboxes [n,9], scores [n], labels [n]).
    Everything is positional and exact -- labels equal, scores within ``score_tol`` (one sigmoid rounding), boxes within
    ``box_tol`` (abs + rel) -- except where torch leaves the result open: inside a group of exactly tied scores the
    order is implementation-defined, and of a tie group that straddles rank ``max_num`` any members may have been
    returned.  There every row must still be a distinct, real candidate of the group (right label, right box), and if the
    whole group lies inside the top ``max_num`` the two row sets must be equal."""
    from oracle import restate as R
    gb, gs, gl = (torch.as_tensor(np.asarray(got[k])).double() for k in ("bboxes", "scores", "labels"))
    rb, rs, rl = (torch.as_tensor(np.asarray(ref[k])).double() for k in ("bboxes", "scores", "labels"))
    assert len(gs) == len(rs), f"{what}: {len(gs)} detections kept, reference keeps {len(rs)}"
    n = len(rs)
    if n == 0:
        return dict(n=0, tied=0)
    assert float((gs - rs).abs().max()) <= score_tol, f"{what}: score error {(gs - rs).abs().max():.2e}"
    logits = torch.as_tensor(np.asarray(cls_logits)).float()
    sig = torch.sigmoid(logits).reshape(-1).double()
    den = R.denormalize_bbox(torch.as_tensor(np.asarray(bbox_preds)).float()).double()
    if z_bottom:
        den[:, 2] = den[:, 2] - den[:, 5] * 0.5

    def close(a, b):
        return bool(((a - b).abs() <= box_tol + box_tol * b.abs()).all())

    i, tied = 0, 0
    while i < n:
        j = i + 1
        while j < n and rs[j] == rs[i]:
            j += 1
        if j - i == 1 and int((sig == rs[i]).sum()) <= 1:
            assert gl[i] == rl[i], f"{what}: rank {i}: label {int(gl[i])} vs {int(rl[i])}"
            assert close(gb[i], rb[i]), f"{what}: rank {i}: box error {(gb[i] - rb[i]).abs().max():.2e}"
        else:
            cand = (sig == rs[i]).nonzero().reshape(-1).tolist()
            inside = int((sig > rs[i]).sum()) + len(cand) <= max_num
            used = set()
            for r in range(i, j):
                hit = next((c for c in cand if c not in used and c % num_classes == int(gl[r])
                            and close(gb[r], den[c // num_classes])), None)
                assert hit is not None, f"{what}: rank {r} (tie group of {len(cand)}) is not a candidate of its tie group"
                used.add(hit)
            if inside:
                left = list(range(i, j))
                for r in range(i, j):
                    m = next((k for k in left if rl[k] == gl[r] and close(gb[r], rb[k])), None)
                    assert m is not None, f"{what}: rank {r}: tie group inside the top-{max_num} differs from the reference's"
                    left.remove(m)
            tied += j - i
        i = j
    return dict(n=n, tied=tied)


def kept_rows(det):
    """[K,11] fixed-shape detections (score = -1 on masked rows) -> dict of the kept rows in rank order."""
    det = torch.as_tensor(det).detach().cpu()
    keep = det[:, 9] >= 0
    return dict(bboxes=det[keep, :9], scores=det[keep, 9], labels=det[keep, 10])


def head_boxes_normalised(all_bbox_preds, pc_range):
    """RaCFormer_head.forward's boxes (cx, cy, w, l, cz, h, sin, cos, vx, vy with METRIC centres, racformer_head.py:
    102-111) back in the decoder's normalised output space, the space north_star's 1e-3 refers to."""
    b = torch.as_tensor(np.asarray(all_bbox_preds)).double().clone()
    b[..., 0] = (b[..., 0] - pc_range[0]) / (pc_range[3] - pc_range[0])
    b[..., 1] = (b[..., 1] - pc_range[1]) / (pc_range[4] - pc_range[1])
    b[..., 4] = (b[..., 4] - pc_range[2]) / (pc_range[5] - pc_range[2])
    return b


def detections_parity(got, ref, score_tol=1e-3, box_tol=2e-3, what="", allowed_unmatched=0):
    """End-to-end detections (decoder + decode on each side's OWN decoder outputs): rows are matched one to one by label
    and box; matched scores agree within ``score_tol``; ranks may only differ where scores are that close (each list must
    itself be sorted); a row may be missing on one side only if its score is within ``score_tol`` of the kept list's
    lowest score (rank-K / threshold boundary) -- or, up to ``allowed_unmatched`` rows per side, because it belongs to a
    query whose last-layer outputs failed the decoder criterion with a shown view flip (the caller passes that count:
    every class of such a query may enter or leave the list)."""
    gb, gs, gl = (torch.as_tensor(np.asarray(got[k])).double() for k in ("bboxes", "scores", "labels"))
    rb, rs, rl = (torch.as_tensor(np.asarray(ref[k])).double() for k in ("bboxes", "scores", "labels"))
    assert bool((gs[1:] <= gs[:-1]).all()), f"{what}: detections are not sorted by score"
    left = list(range(len(gs)))
    unmatched_ref = []
    for i in range(len(rs)):
        tol = box_tol + box_tol * rb[i].abs()
        m = next((k for k in left if gl[k] == rl[i] and abs(float(gs[k] - rs[i])) <= score_tol
                  and bool(((gb[k] - rb[i]).abs() <= tol).all())), None)
        if m is None:
            unmatched_ref.append(i)
        else:
            left.remove(m)
    floor = min(float(rs.min()) if len(rs) else 1.0, float(gs.min()) if len(gs) else 1.0)
    inner_ref = [i for i in unmatched_ref if float(rs[i]) > floor + score_tol]
    inner_got = [k for k in left if float(gs[k]) > floor + score_tol]
    assert len(inner_ref) <= allowed_unmatched, \
        f"{what}: reference detections {inner_ref} (scores {[round(float(rs[i]), 4) for i in inner_ref]}) have no counterpart"
    assert len(inner_got) <= allowed_unmatched, \
        f"{what}: detections {inner_got} (scores {[round(float(gs[k]), 4) for k in inner_got]}) are not in the reference's list"
    return dict(matched=len(rs) - len(unmatched_ref), unmatched_ref=len(unmatched_ref), unmatched_got=len(left))


TF_STAGES = ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling", "mixing", "ffn")


def teacher_forced_layer_check(l, g, cfg, feat, cls, box, stages, views_l, tol=1e-4, what=""):
    """One decoder layer fed the REFERENCE's own (query_bbox, query_feat) of layer ``l`` (fixture decoder_f8_tf.npz): no
    error is carried in from earlier layers, so every one of the Q queries must agree within ``tol`` (abs + rel) on the
    layer's outputs (features, class logits, refined box) and, for the probe queries the fixture keeps, on every stage.
    The only exemption is a query with a SHOWN first-valid-view flip in this very layer (``views_l`` [S,Q,P] vs the
    fixture's); self-attention precedes the sampling, so neighbours are not exempt.  -> number of flipped queries."""
    t = lambda a: torch.as_tensor(np.asarray(a)).double().cpu()   # noqa: E731
    L = g["in_feat"].shape[0]
    flips = flipped_queries(np.asarray(views_l)[None], g["views"][l][None], cfg.num_frames, cfg.num_groups)[0]   # [B,Q]
    ref_feat = t(g["in_feat"][l + 1]) if l + 1 < L else t(g["out_feat_last"])

    def worst(a, b, keep):
        a, b = t(a), t(b)
        excess = ((a - b).abs() - tol * b.abs()).amax(-1) if a.dim() == 3 else ((a - b).abs() - tol * b.abs()).flatten(2).amax(-1)
        return float(torch.where(keep, excess, torch.zeros_like(excess)).max())

    keep = ~flips
    for name, a, b in (("query_feat", feat, ref_feat), ("cls", cls, g["out_cls"][l]), ("box", box, g["out_box"][l])):
        w = worst(a, b, keep)
        assert w <= tol, f"{what} layer {l} {name}: max excess error {w:.2e} over {int(keep.sum())} un-flipped queries (tol {tol})"
    if stages is not None:
        pq, ps = torch.as_tensor(g["probe_q"]), torch.as_tensor(g["probe_q_sampling"])
        for s_ in TF_STAGES:
            sel = ps if s_ == "sampling" else pq
            upstream = s_ in ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev")
            k = torch.ones_like(flips[:, sel]) if upstream else ~flips[:, sel]
            w = worst(t(stages[s_])[:, sel], g["stage_" + s_][l], k)
            assert w <= tol, f"{what} layer {l} stage {s_}: max excess error {w:.2e} (tol {tol})"
    return int(flips.sum())


def init_rig_params(g, golden_dir):
    """The parameters the reference's init_weights() wrote on the init_weights rig (fixture key ``init_params`` names the
    file that holds them, bit for bit): {state_dict key: tensor}, empty for the random-everything rigs."""
    import os
    if "init_params" not in getattr(g, "files", g):
        return {}
    z = np.load(os.path.join(golden_dir, str(g["init_params"])))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def fill_scheme(g):
    return str(g["fill_scheme"]) if "fill_scheme" in getattr(g, "files", g) else "tamed_normal"


def load_rig_state_dict(cfg, g, golden_dir):
    """state_dict of the rig a decoder fixture was generated on: the seeded fill, then (init rig) what init_weights() wrote."""
    from racformer_amd import synthetic as syn
    sd = syn.make_state_dict(cfg, int(g["weight_seed"]), scheme=fill_scheme(g))
    for k, v in init_rig_params(g, golden_dir).items():
        assert k in sd and sd[k].shape == v.shape, k
        sd[k] = v.clone()
    return sd


def fill_rig_module(module, cfg, g, golden_dir):
    """The same for a product (or reference) RaCFormerTransformer module, in place."""
    from racformer_amd import synthetic as syn
    syn.fill_params(module, int(g["weight_seed"]), scheme=fill_scheme(g))
    over = init_rig_params(g, golden_dir)
    if over:
        sd = module.state_dict()
        with torch.no_grad():
            for k, v in over.items():
                assert sd[k].shape == v.shape, k
                sd[k].copy_(v)
    return module


def oracle_decoder(R, sd, qb, qf, pyramid, lss, radar, metas, cfg, stages=None, force_views=None):
    """The oracle's decoder forward -> (cls, box, views): views = the camera index it used for every sampling point of
    every layer ([layers,S,Q,P] uint8).  ``force_views`` [layers,S,Q,P]: impose these choices instead of its own."""
    R.LOC_TAP = []
    R.VIEW_FORCE = [np.asarray(v) for v in force_views] if force_views is not None else None
    try:
        with torch.no_grad():
            cls, box = R.transformer_forward(sd, qb, qf, pyramid, lss, radar, metas, cfg, stages)
        views = torch.stack([R.views_of(l, cfg.num_cams) for l in R.LOC_TAP])
    finally:
        R.LOC_TAP = R.VIEW_FORCE = None
    return cls, box, views


def run_with_reference_views(run, ref_views, what=""):
    """``run(force_views) -> (outputs..., views)`` where ``views`` are the implementation's OWN camera choices (also when
    others are imposed).  Runs once freely; if any choice differs from ``ref_views`` (a differing choice moves the query, so
    later layers differ as a consequence), runs again with the reference's choices imposed: on that equalised trajectory the
    implementation's own choices differ from the reference's only at genuinely borderline points, which are counted and
    bounded.  -> (outputs of the comparable run, differing points per layer on the comparable trajectory)."""
    res = run(None)
    nflip = flipped_points(res[-1], ref_views)
    free = list(nflip)
    per_layer = int(np.prod(np.asarray(ref_views).shape[1:]))
    if sum(nflip):
        print(f"{what}: free run: differing camera choices per layer {nflip} (incl. consequences of earlier ones)")
        res = run(ref_views)
        nflip = flipped_points(res[-1], ref_views)
    print(f"{what}: sampling points whose camera choice differs from the comparand's on the same trajectory, per layer: {nflip} "
          f"(of {per_layer} per layer)")
    USED.append(dict(what=what, kind="camera_choices", free_run_differing=free, imposed=bool(sum(free)),
                     differing_on_equalised_trajectory=nflip, points_per_layer=per_layer))
    assert sum(nflip) <= MAX_FLIPPED_POINTS, f"{what}: {nflip} differing camera choices"
    return res, nflip
