"""Decoder-, head- and decode-level parity of the HIP path against the reference's fixtures and the CPU oracle under the
criteria of tests/parity.py: teacher-forced per-layer comparison at 1e-4 for all 900 queries; free-running six layers with
the differing camera choices counted, bounded and then equalised (the reference's choices imposed through the kernel's
view_in), class argmax identical and boxes within 1e-3 with a measured per-layer tail budget; the NMS-free decode
positional and exact.  Nothing here compares the product with itself."""
import os

import numpy as np
import pytest
import torch

import plans  # noqa: F401  (registers the alternate execution plans: decoder_layer.fused / .rowgemm = False)
from oracle import restate as R
from parity import (ARGMAX_MARGIN_INIT_RIG, decode_parity, decoder_parity, detections_parity, fill_rig_module, flipped_points, head_boxes_normalised,
                    kept_rows, oracle_decoder, run_with_reference_views, teacher_forced_layer_check)
from racformer_amd import synthetic as syn
from racformer_amd.head import RaCFormer_head
from racformer_amd.transformer import RaCFormerTransformer, regroup_pyramid

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]


def t(a):
    return torch.from_numpy(np.asarray(a))


def gpu_views(layer, cfg):
    """[layers, S, Q, P] camera indices the sampling kernel selected (its own loc_out, collected by capture_loc)."""
    return torch.stack([R.views_of(l.cpu(), cfg.num_cams) for l in layer.sampling.capture_loc])


def run_decoder_gpu(cfg, seed, wseed, force_views=None, rig=None, **layer_flags):
    """-> (cls, box, views): one forward of the product decoder; ``force_views`` [layers,S,Q,P] imposes the camera choices.
    ``rig`` = (fixture, golden_dir): the weights of the rig that fixture was generated on (init_weights rig) instead of the
    seeded random-everything fill."""
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    if rig is not None:
        fill_rig_module(tr, cfg, *rig)
    else:
        syn.fill_params(tr, wseed)
    layer = tr.decoder.decoder_layer
    for k, v in layer_flags.items():
        assert hasattr(layer, k), k
        setattr(layer, k, v)
    tr = tr.to(DEV)
    layer.sampling.capture_loc = []
    if force_views is not None:
        layer.sampling.force_views = [torch.as_tensor(np.asarray(v)).to(DEV).contiguous() for v in force_views]
    qb, qf = syn.make_queries(cfg, seed)
    with torch.no_grad():
        cls, box = tr(qb.to(DEV), qf.to(DEV), [f.to(DEV) for f in syn.make_pyramid(cfg, seed)],
                      syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV), None, syn.make_img_metas(cfg))
    torch.cuda.synchronize()
    return cls.cpu(), box.cpu(), gpu_views(layer, cfg)


# ------------------------------------------------------------------------------------------------ whole decoder
F8_FIXTURES = [("decoder_f8.npz", syn.F8), ("decoder_f8_s1.npz", syn.F8), ("decoder_f8_s2.npz", syn.F8),
               ("decoder_f8_s3.npz", syn.F8), ("decoder_f8_3cam.npz", syn.F8_3CAM), ("decoder_f8_3cam_s1.npz", syn.F8_3CAM)]


@pytest.mark.parametrize("name,cfg", F8_FIXTURES)
def test_decoder_f8_vs_reference_all_seeds(golden_dir, name, cfg):
    """Four seeds of the 6-cam rig and two of the 3-cam rig at full f8 shapes against the reference's own CPU forward."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force), g["views"], name)
    decoder_parity(cls, box, g["cls"], g["box"], what=name)


@pytest.mark.parametrize("name,cfg", [("decoder_f8_init.npz", syn.F8), ("decoder_f8_3cam_init.npz", syn.F8_3CAM)])
def test_decoder_f8_init_weights_rig_literal(golden_dir, name, cfg):
    """SURVEY 8d's second rig at full f8 shapes: the model as the reference initialises it (torch's constructor distributions,
    then the reference's own init_weights(); what that wrote is in init_params_w7.npz).  north_star's criterion, literally:
    every query of all six free-running layers within 1e-3 on the box, class argmax identical -- no tail budget."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force, rig=(g, golden_dir)),
                                                g["views"], name)
    # (the tie rule at THIS rig's resolution: its CPU-vs-CPU logit drift is 4.2e-5, not the random rig's 1.2e-2)
    decoder_parity(cls, box, g["cls"], g["box"], what=name, tail_budget=None, argmax_margin=ARGMAX_MARGIN_INIT_RIG)


@pytest.mark.parametrize("name,cfg", [("decoder_small.npz", syn.SMALL), ("decoder_small6.npz", syn.SMALL6)])
def test_decoder_small_vs_reference_literal(golden_dir, name, cfg):
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force), g["views"], name)
    decoder_parity(cls, box, g["cls"], g["box"], what=name, tail_budget=None)


def test_decoder_f8_vs_oracle_unseen_seed():
    """A seed no fixture covers, against the oracle run beside it (the oracle's camera choices imposed if any differ)."""
    cfg, seed, wseed = syn.F8, 17, 18
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    ocls, obox, oviews = oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                        syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg)
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force), oviews, "f8 seed 17")
    decoder_parity(cls, box, ocls, obox, what="f8 seed 17 vs oracle")


@pytest.mark.parametrize("name,cfg", [("decoder_f8.npz", syn.F8), ("decoder_f8_3cam.npz", syn.F8_3CAM)])
def test_decoder_bf16_feature_storage_measured_gap(golden_dir, name, cfg):
    """BASELINE configs 3 / 5 name bf16.  The gather kernels take a bf16 pyramid (`decoder.feature_dtype = torch.bfloat16`: half
    the sampling kernel's bytes), and this is what that costs against the reference's fp32 CPU forward, as a test with its
    own stated bound rather than prose: one layer stays inside north_star's tolerance (boxes < 1e-3, argmax identical --
    measured 7.3e-4 / 6.3e-4 at most), six free-running layers do not (measured at layer 5: median 9.6e-4 / 3.1e-4,
    433 / 241 of 900 queries over 1e-3, 3 / 2 argmax flips), which is why the default and the benchmark keep fp32 storage."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    tr = tr.to(DEV)
    tr.decoder.feature_dtype = torch.bfloat16
    qb, qf = syn.make_queries(cfg, seed)
    with torch.no_grad():
        cls, box = tr(qb.to(DEV), qf.to(DEV), [f.to(DEV) for f in syn.make_pyramid(cfg, seed)],
                      syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV), None, syn.make_img_metas(cfg))
    eb = (box.cpu() - t(g["box"])).abs().amax(-1)[:, 0]
    mism = (cls.cpu().argmax(-1) != t(g["cls"]).argmax(-1))[:, 0]
    print("bf16 features:", [f"L{l} p50 {eb[l].median():.1e} max {eb[l].max():.1e} argmax {int(mism[l].sum())}" for l in range(eb.shape[0])])
    assert eb[0].max().item() < 1e-3 and int(mism[0].sum()) == 0                 # one layer: inside the tolerance
    assert eb[-1].median().item() < 2e-3 and int(mism[-1].sum()) <= 9            # six layers: bounded, but outside it
    assert int((eb[-1] > 1e-3).sum()) > 9, "bf16 storage now meets the tolerance: make it the default and update DESIGN"


@pytest.mark.parametrize("levels", [(0,), (0, 1)], ids=["c2", "c2c3"])
def test_decoder_bf16_fine_levels_only_measured_gap(golden_dir, levels):
    """Round-2 verdict, item 9: bf16 storage for the fine pyramid levels only (c2 alone = 75 %, c2 + c3 = 94 % of the pyramid's bytes),
    coordinates, weights and the coarse levels in fp32.  Storage in bf16 is exactly a round-to-nearest of the stored values (the kernels
    widen bf16 to fp32 and accumulate in fp32), so the experiment rounds those levels of the regrouped fp32 pyramid and runs the fp32
    kernels on it -- the accuracy a mixed-dtype kernel would have, before anybody builds one.  Measured against the reference's fp32
    CPU forward (decoder_f8.npz), asserted as measured: one layer stays inside north_star's tolerance, six layers do not."""
    name, cfg = "decoder_f8.npz", syn.F8
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    tr = tr.to(DEV)
    tr.decoder.pregrouped = True
    grouped = regroup_pyramid([f.to(DEV) for f in syn.make_pyramid(cfg, seed)], cfg.num_cams)
    for l in levels:
        grouped[l] = grouped[l].to(torch.bfloat16).to(torch.float32)
    qb, qf = syn.make_queries(cfg, seed)
    with torch.no_grad():
        cls, box = tr(qb.to(DEV), qf.to(DEV), grouped, syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV), None,
                      syn.make_img_metas(cfg))
    eb = (box.cpu() - t(g["box"])).abs().amax(-1)[:, 0]
    mism = (cls.cpu().argmax(-1) != t(g["cls"]).argmax(-1))[:, 0]
    over = [(int((eb[l] > 1e-3).sum())) for l in range(eb.shape[0])]
    print("bf16 levels", levels, [f"L{l} p50 {eb[l].median():.1e} max {eb[l].max():.1e} >1e-3: {over[l]} argmax {int(mism[l].sum())}"
                                for l in range(eb.shape[0])])
    assert eb[0].max().item() < 1e-3 and int(mism[0].sum()) == 0                 # one layer: inside the tolerance
    assert eb[-1].median().item() < 2e-3                                         # six layers: bounded ...
    assert over[-1] > 10, "bf16 storage of the fine levels now meets the tolerance: build the mixed-dtype kernel and update DESIGN"


# ------------------------------------------------------------------------------------------------ teacher forcing
def test_decoder_f8_teacher_forced(golden_dir):
    """Every decoder layer (each d_region) fed the reference's own (query_bbox, query_feat): all 900 queries within 1e-4 on
    the layer's outputs, the fixture's probes within 1e-4 on every stage; only queries with a shown view flip in that very
    layer are exempt.  Also pins the hoisted temporal encoder / value streams, whose outputs every stage reads."""
    g = np.load(os.path.join(golden_dir, "decoder_f8_tf.npz"))
    cfg = syn.F8
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    tr = tr.to(DEV)
    dec, layer = tr.decoder, tr.decoder.decoder_layer
    metas = syn.make_img_metas(cfg)
    dec.stage_metas(metas, 1, torch.device(DEV))
    feats = regroup_pyramid([f.to(DEV) for f in syn.make_pyramid(cfg, seed)], cfg.num_cams)
    lss, radar = syn.make_bev(cfg, seed, 0).to(DEV), syn.make_bev(cfg, seed, 1).to(DEV)
    nflips = []
    with torch.no_grad():
        prepared = layer.prepare(lss, radar)
        for l in range(cfg.num_layers):
            st = {}
            layer._carry, layer.sampling.capture_loc = None, []
            feat, cls, box = layer(t(g["in_bbox"][l]).to(DEV), t(g["in_feat"][l]).to(DEV), feats, lss, radar, None, metas,
                                   layer=l, prepared=prepared, stages=st)
            torch.cuda.synchronize()
            views = R.views_of(layer.sampling.capture_loc[0].cpu(), cfg.num_cams)
            st = {k: v.cpu() for k, v in st.items()}
            nflips.append(teacher_forced_layer_check(l, g, cfg, feat.cpu(), cls.cpu(), box.cpu(), st, views, what="HIP"))
    print("teacher-forced: flipped queries per layer", nflips)
    assert sum(nflips) <= 12, nflips


def test_temporal_encoder_vs_reference_probe(golden_dir):
    """The hoisted radar temporal encoder against the reference module's own output (strided probe kept in
    decoder_f8.npz) and against the oracle's restatement on the full map."""
    g = np.load(os.path.join(golden_dir, "decoder_f8.npz"))
    cfg = syn.F8
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, wseed)
    sd = R._sub({k: v.clone() for k, v in tr.state_dict().items()}, "decoder.decoder_layer.")
    tr = tr.to(DEV)
    rbs = tr.decoder.decoder_layer.sampling_radar_bev
    te = rbs.temporal_encoder
    radar = syn.make_bev(cfg, seed, 1)
    from racformer_amd.fused import pack_conv3x3_weight
    with torch.no_grad():
        ws, alpha = pack_conv3x3_weight(te.temporal_fusion.weight)
        H, W = radar.shape[-2:]
        got = te.forward_channel_last(radar.to(DEV), dict(ws=ws, alpha=alpha, bound=te.hidden_bound(), **te.downsample_pack(H, W)))
        got = got.permute(0, 3, 1, 2).reshape(1, cfg.num_frames, 256, H, W).cpu()      # [B,T,C,H,W]
        want = R.temporal_encoder(sd, "sampling_radar_bev.temporal_encoder", radar)
    scale = float(want.abs().max())
    assert (got - want).abs().max().item() <= 2e-5 * scale + 1e-5, (got - want).abs().max().item()
    probe = got[:, :, ::37, ::9, ::11]
    assert (probe - t(g["temporal_encoder_L0_probe"])).abs().max().item() <= 2e-5 * scale + 1e-5


# ------------------------------------------------------------------------------------------------ execution plans
PLANS = [("default plan", {}), ("library GEMM chain instead of rowgemm", dict(rowgemm=False)),
         ("fp32 library GEMMs for the mixing Linears", dict(split_gemm=False)),
         ("the reference's op decomposition (torch keypoints + msmv / MSDA operators)", dict(fused=False))]


@pytest.mark.parametrize("cfg", [syn.SMALL6, syn.F8], ids=["small6", "f8"])
def test_every_execution_plan_vs_oracle_six_layers(cfg):
    """All six layers of every execution plan the product can run, each against the ORACLE on the same seeded inputs with
    equal camera choices -- not against each other, and not cut to the first layers."""
    seed, wseed = 71, 72
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    ocls, obox, oviews = oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                        syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg)
    for what, flags in PLANS:
        (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force, **flags), oviews, what)
        decoder_parity(cls, box, ocls, obox, what=what, tail_budget=None if cfg is syn.SMALL6 else parity_budget())


def parity_budget():
    from parity import TAIL_BUDGET
    return TAIL_BUDGET


# ------------------------------------------------------------------------------------------------ decode
def product_coder(K, thr, C):
    from racformer_amd.head import NMSFreeCoder
    return NMSFreeCoder(pc_range=list(syn.PC_RANGE), post_center_range=POST_RANGE, max_num=K, score_threshold=thr, num_classes=C)


@pytest.mark.parametrize("case,tag,thr", [("A", "thr", 0.05), ("A", "none", None), ("A", "zero", 0.0), ("B", "thr", 0.05),
                                          ("C", "thr", 0.05)])
def test_decode_kernel_vs_reference_cases(golden_dir, case, tag, thr):
    """rac_decode_fwd and the torch formulation (NMSFreeCoder.decode_single / get_bboxes on device tensors) against the
    reference's decode_single / get_bboxes outputs: ties inside the top-K, a tie group across rank K, centres outside and
    exactly on post_center_range, scores either side of the threshold, the 0.0-threshold quirk, saturated sigmoids."""
    from racformer_amd.fused import decode_fused
    g = np.load(os.path.join(golden_dir, "decode_cases.npz"))
    cls, box, K = t(g[f"{case}_cls"]), t(g[f"{case}_box"]), int(g[f"{case}_K"])
    C = cls.shape[1]
    ref = dict(bboxes=g[f"{case}_{tag}_get_bboxes"], scores=g[f"{case}_{tag}_get_scores"], labels=g[f"{case}_{tag}_get_labels"])
    ref_single = dict(bboxes=g[f"{case}_{tag}_bboxes"], scores=g[f"{case}_{tag}_scores"], labels=g[f"{case}_{tag}_labels"])
    det = decode_fused(cls.to(DEV), box.to(DEV), K, POST_RANGE, thr).cpu()
    assert tuple(det.shape) == (K, 11)
    info = decode_parity(kept_rows(det), ref, cls, box, K, C, what=f"kernel {case}/{tag}")
    assert info["n"] == len(ref["scores"])
    # the oracle on the same inputs agrees with both
    decode_parity(kept_rows(det), R.nms_free_decode(cls, box, K, C, thr, POST_RANGE), cls, box, K, C, what=f"kernel vs oracle {case}/{tag}")
    # torch formulation of the product (host-synchronising drop-in API)
    coder = product_coder(K, thr, C)
    single = coder.decode_single(cls.to(DEV), box.to(DEV))
    decode_parity({k: v.cpu() for k, v in single.items()}, ref_single, cls, box, K, C, z_bottom=False, what=f"decode_single {case}/{tag}")
    head = RaCFormer_head.__new__(RaCFormer_head)          # get_bboxes / get_detections_fixed only read bbox_coder
    torch.nn.Module.__init__(head)
    head.bbox_coder = coder
    preds = dict(all_cls_scores=cls.to(DEV)[None, None], all_bbox_preds=box.to(DEV)[None, None])
    b, s, l = head.get_bboxes(preds, None)[0]
    decode_parity(dict(bboxes=b.cpu(), scores=s.cpu(), labels=l.cpu()), ref, cls, box, K, C, what=f"get_bboxes {case}/{tag}")
    fixed = head.get_detections_fixed(preds)[0].cpu()
    decode_parity(kept_rows(fixed), ref, cls, box, K, C, what=f"get_detections_fixed {case}/{tag}")


@pytest.mark.parametrize("Q,C,K", [(20, 10, 300), (1600, 10, 512), (7, 3, 5), (900, 10, 1)])
def test_decode_kernel_vs_oracle_shapes(Q, C, K):
    """Shapes beside the configured one (K > Q*C, K = 1, Q*C near the kernel's limit) against the oracle."""
    from racformer_amd.fused import decode_fused
    rng = np.random.default_rng(Q + K)
    cls = torch.from_numpy((rng.standard_normal((Q, C)) * 2.0).astype(np.float32))
    box = torch.from_numpy(rng.standard_normal((Q, 10)).astype(np.float32))
    box[:, 0:2] *= 40.0
    det = decode_fused(cls.to(DEV), box.to(DEV), K, POST_RANGE, 0.3).cpu()
    n = min(K, Q * C)
    want = R.nms_free_decode(cls, box, n, C, 0.3, POST_RANGE)
    decode_parity(kept_rows(det), want, cls, box, n, C, what=f"decode {Q}x{C} top-{K}")
    assert bool((det[n:, 9] == -1).all())


# ------------------------------------------------------------------------------------------------ head
def build_head(cfg, g, seed, wseed):
    head = RaCFormer_head(
        num_classes=cfg.num_classes, in_channels=cfg.embed_dims, num_query=cfg.num_query, num_clusters=cfg.num_clusters,
        code_size=cfg.code_size, transformer=dict(type="RaCFormerTransformer", **cfg.transformer_kwargs()),
        bbox_coder=dict(type="NMSFreeCoder", post_center_range=POST_RANGE, pc_range=list(cfg.pc_range), max_num=300,
                        score_threshold=0.05, num_classes=cfg.num_classes))
    # what _init_layers / generate_points put into the query embedding (racformer_head.py:51-79), bit for bit
    cols = g["init_query_cols"].tolist()
    assert torch.equal(head.init_query_bbox.weight.detach()[:, cols], t(g["init_query_fixed"]))
    assert torch.equal(head.generate_points(), t(g["generate_points"]))
    syn.fill_params(head.transformer, wseed)
    with torch.no_grad():
        head.label_enc.weight.copy_(t(g["label_enc"]))
        head.init_query_bbox.weight.copy_(syn.make_queries(cfg, seed)[0][0])
    return head.eval().to(DEV)


@pytest.mark.parametrize("name,cfg", [("head_small6.npz", syn.SMALL6), ("head_f8.npz", syn.F8)])
def test_head_forward_and_detections_vs_reference(golden_dir, name, cfg):
    """RaCFormer_head.forward -> get_detections_fixed (rac_decode_fwd) / get_bboxes against the reference head's outputs:
    decoder outputs under the literal criterion in normalised units, the decode strictly on the reference's own head
    outputs, the end-to-end detection list as a one-to-one match."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    head = build_head(cfg, g, seed, wseed)
    layer = head.transformer.decoder.decoder_layer

    def run(force):
        layer.sampling.capture_loc = []
        layer.sampling.force_views = [t(v).to(DEV).contiguous() for v in force] if force is not None else None
        with torch.no_grad():
            preds = head([f.to(DEV) for f in syn.make_pyramid(cfg, seed)], syn.make_bev(cfg, seed, 0).to(DEV),
                         syn.make_bev(cfg, seed, 1).to(DEV), syn.make_img_metas(cfg))
            fixed = head.get_detections_fixed(preds)[0].cpu()
            bsl = head.get_bboxes(preds, None)[0]
        torch.cuda.synchronize()
        return preds, fixed, bsl, gpu_views(layer, cfg)

    (preds, fixed, (b, s, l), _), _ = run_with_reference_views(run, g["views"], name)
    assert preds["enc_cls_scores"] is None and preds["enc_bbox_preds"] is None
    ref_n = head_boxes_normalised(g["all_bbox_preds"], cfg.pc_range)
    rows = decoder_parity(preds["all_cls_scores"].cpu(), head_boxes_normalised(preds["all_bbox_preds"].cpu(), cfg.pc_range),
                          g["all_cls_scores"], ref_n, what=name, tail_budget=None if cfg is syn.SMALL6 else parity_budget())
    # decode, strictly: the reference's last-layer outputs through the kernel and the torch formulation
    ref_det = dict(bboxes=g["det_boxes"], scores=g["det_scores"], labels=g["det_labels"])
    rcls, rbox = t(g["all_cls_scores"])[-1, 0], t(g["all_bbox_preds"])[-1, 0]
    rp = dict(all_cls_scores=rcls.to(DEV)[None, None], all_bbox_preds=rbox.to(DEV)[None, None])
    decode_parity(kept_rows(head.get_detections_fixed(rp)[0]), ref_det, rcls, rbox, 300, cfg.num_classes, what=name + " kernel decode")
    rb, rs, rl = head.get_bboxes(rp, None)[0]
    decode_parity(dict(bboxes=rb.cpu(), scores=rs.cpu(), labels=rl.cpu()), ref_det, rcls, rbox, 300, cfg.num_classes,
                  what=name + " get_bboxes")
    # end to end
    allow = 3 * rows[-1]["failing"]
    detections_parity(kept_rows(fixed), ref_det, what=name + " end-to-end (kernel)", allowed_unmatched=allow)
    detections_parity(dict(bboxes=b.cpu(), scores=s.cpu(), labels=l.cpu()), ref_det, what=name + " end-to-end (get_bboxes)",
                      allowed_unmatched=allow)


def test_layer_boundary_kernel_vs_oracle():
    """rac_layer_boundary_fwd (refine_bbox + velocity scaling + theta_d2xy + next layer's box table and position-encoder
    head) against the oracle's restatement of racformer_transformer.py:230-236, 265-269, bbox/utils.py:66-90."""
    from racformer_amd.fused import layer_boundary_fused
    rng = np.random.default_rng(17)
    B, Q = 2, 37
    prop = torch.from_numpy(rng.random((B, Q, 10)).astype(np.float32))
    prop[0, 0, 1:3] = torch.tensor([0.0, 1.0])                     # inverse_sigmoid clamps
    delta = torch.from_numpy(rng.standard_normal((B, Q, 10)).astype(np.float32))
    td = torch.tensor([[0.0, 0.5, 1.0], [0.0, 1.0, 1.5]])
    td_safe = td.clone()
    td_safe[td_safe < 1e-5] = 1.0
    lin, ln = torch.nn.Linear(3, 256), torch.nn.LayerNorm(256)
    torch.nn.init.normal_(ln.weight)
    torch.nn.init.normal_(ln.bias)
    want = R.refine_bbox(prop, delta, 150)
    want = torch.cat([want[..., :8], want[..., 8:] / td_safe[:, 1:2, None]], dim=-1)
    want_xy = R.theta_d2xy(want)
    dec = R.decode_bbox(want_xy, syn.PC_RANGE)                    # (cx,cy,cz,w,l,h,yaw,vx,vy)
    with torch.no_grad():
        want_h = torch.relu(ln(lin(want[..., :3])))
        pred, xy, table, h = layer_boundary_fused(prop.to(DEV), delta.to(DEV), td_safe.to(DEV), 150, syn.PC_RANGE, lin.to(DEV), ln.to(DEV))
    assert (pred.cpu() - want).abs().max().item() < 1e-5
    assert (xy.cpu() - want_xy).abs().max().item() < 1e-5
    table = table.cpu()
    assert (table[..., :6] - dec[..., :6]).abs().max().item() < 2e-5 * float(dec[..., :6].abs().max())
    assert (table[..., 6] - torch.cos(dec[..., 6])).abs().max().item() < 1e-5
    assert (table[..., 7] - torch.sin(dec[..., 6])).abs().max().item() < 1e-5
    assert (h.cpu() - want_h).abs().max().item() < 2e-5


def test_decoder_small_batch2_vs_oracle():
    """Two samples per forward (reduced 6-cam shapes): the B > 1 paths of the kernels -- the slot order of the image sampling
    (quirk Q1), the frame / batch pairing of the BEV attention (quirk Q2: the keypoint chain per (t', b') instead of the hoisted
    base points), per-batch time_diff and lidar2img -- against the oracle's restatement of the reference, literal for every
    query.  (No reference fixture exists at B = 2: this pins the kernels to the oracle, whose B = 1 behaviour is pinned.)"""
    from dataclasses import replace
    cfg, seed, wseed = replace(syn.SMALL6, batch=2), 23, 24
    sd = syn.make_state_dict(cfg, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    ocls, obox, oviews = oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0),
                                        syn.make_bev(cfg, seed, 1), syn.make_img_metas(cfg), cfg)
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force), oviews, "small6 batch 2")
    assert tuple(cls.shape[:2]) == (6, 2)
    decoder_parity(cls, box, ocls, obox, what="small6 batch 2 vs oracle", tail_budget=None)


@pytest.mark.parametrize("cfg", [syn.F8, syn.F8_3CAM], ids=["f8", "f8_3cam"])
def test_decoder_sampling_inside_the_mixing_kernel_same_bits(cfg):
    """`fuse_sampling_mixing = True` (rac_mixing_sampled_fwd: the mixing workgroup gathers its own sampled features) against the default
    plan (rac_sampling4d_fwd -> rac_mixing_fwd): six free-running layers, the same bits -- class scores, boxes, camera choices.  (The
    fused launch is not the default because it is not faster; every parity statement about the default plan therefore
    holds for it as well.)"""
    a = run_decoder_gpu(cfg, 27, 28, fuse_sampling_mixing=False)
    b = run_decoder_gpu(cfg, 27, 28, fuse_sampling_mixing=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
