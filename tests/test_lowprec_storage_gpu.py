"""16-bit STORAGE of the sampled tensors, measured where it might legitimately pass (round-3 verdict, item 4) and asserted as
measured -- profiles/r04_lowprec_storage.json holds the full table (tools/exp_lowprec.py).

BASELINE configs 3 / 5 name bf16.  Round 3 measured bf16 storage only on the random-everything rig (which amplifies rounding
4-5x per layer).  Here, with the reference's camera choices imposed and fp32 arithmetic throughout:

(i)  bf16 PYRAMID on the rig as the reference initialises it (decoder_f8_init.npz / decoder_f8_3cam_init.npz -- the rig on
     which fp32 is literal with 5e-5 / 1.1e-4 to spare): NOT literal.  6-cam: 2 / 4 of 900 queries over 1e-3 in layers 4 / 5
     (max 3.3e-3) and two argmax changes in layer 1; 3-cam: queries over 1e-3 from layer 1 on, max 3.1e-2.  The fine levels
     alone (c2: 75 % of the bytes) still lose one query in layers 4 and 5.  So bf16 pyramid storage stays an op-level option
     (`decoder.feature_dtype`), is not benchmarked, and no mixed-dtype kernel is built.
(ii) the two hoisted BEV VALUE STREAMS (what `bev_sampling_d64_kernel` gathers: 2 x 134 MB) with an fp32 pyramid:
     bf16 is not literal on the init rig either (1 / 1 / 3 queries in layers 3-5); **f16 (11 significant bits) IS**: literal on
     both init rigs (max 1.4e-4 / 7.0e-4, argmax identical) and on the random rig's decoder_f8.npz (max 7.5e-4, nothing over
     1e-3) -- but on that rig's chaotic seed (decoder_f8_s1.npz, where fp32 itself has 1 / 8 queries over 1e-3 in layers
     4 / 5) it has 8 / 33, beyond the tail budget the fp32 path is held to (3 / 11).  f16 value streams would halve the
     bytes through the CU's texture path that bound the BEV kernel, so the result is recorded precisely; the
     default keeps fp32 because one committed fixture fails with it.
(iii) int16 BLOCK storage of the value streams (one power-of-two scale per (pixel, head) block of 64 channels, 14-15 significant
     bits relative to the block's largest value: csrc/quant.hip) -- the format that is actually built
     (`decoder_layer.value_storage = "i16"`, rac_bev_sampling_multi_q16_fwd): literal on both init rigs and inside the SAME
     tail budget as fp32 on the random rig including its chaotic seed (emulated: 0/0/0/0/1/7 queries over 1e-3 against fp32's
     0/0/0/0/1/8).  `test_decoder_int16_block_value_streams_*` run the REAL kernels under the fp32 path's own criteria."""
import os

import numpy as np
import pytest
import torch

from lowprec import GOLD, rig_inputs, run
from parity import TAIL_QUERIES
from racformer_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _g(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name,cfg", [("decoder_f8_init.npz", syn.F8), ("decoder_f8_3cam_init.npz", syn.F8_3CAM)])
def test_init_rig_bf16_pyramid_and_16bit_value_streams_measured(name, cfg):
    g = _g(name)
    inputs = rig_inputs(cfg, int(g["seed"]))
    fp32 = run(cfg, g, inputs, True)
    # (fp32 storage: literal, with room -- measured 5e-5 ... 2.2e-4 at most over the builds of rounds 3-4: the last digit follows the
    #  compiler's contraction pattern in the footprint arithmetic)
    assert sum(fp32["queries_over_1e-3"]) == 0 and sum(fp32["argmax_mismatches"]) == 0 and max(fp32["box_max"]) < 5e-4
    # (i) bf16 pyramid: one layer inside the tolerance, six free-running layers not literal
    pyr = run(cfg, g, inputs, True, pyramid_dtype=torch.bfloat16)
    print(name, "pyramid bf16:", pyr)
    assert pyr["box_max"][0] < 1e-3 and pyr["argmax_mismatches"][0] == 0
    assert sum(pyr["queries_over_1e-3"]) + sum(pyr["argmax_mismatches"]) > 0, \
        "bf16 pyramid storage is now literal on the init rig: ship it as an opt-in with a second bench line and update DESIGN 3.11"
    assert max(pyr["queries_over_1e-3"]) <= 12 and max(pyr["box_max"]) < 0.1              # ... but bounded: a handful of queries
    c2 = run(cfg, g, inputs, True, round_levels=(0,))
    print(name, "pyramid c2 bf16:", c2)
    assert sum(c2["queries_over_1e-3"][:4]) == 0 and sum(c2["queries_over_1e-3"]) <= 6        # (measured 2 / 1: at the edge of the tolerance)
    # (ii) BEV value streams: bf16 not literal, f16 literal
    vb = run(cfg, g, inputs, True, value_dtype=torch.bfloat16)
    print(name, "values bf16:", vb)
    assert sum(vb["queries_over_1e-3"]) > 0 and max(vb["queries_over_1e-3"]) <= 12
    vh = run(cfg, g, inputs, True, value_dtype=torch.float16)
    print(name, "values f16:", vh)
    assert max(vh["value_abs_max"].values()) < 6.0e4                                         # (inside f16's range on this rig: no scale needed)
    # (boxes literal; the class argmax identical on the 6-cam rig and, depending on the build's rounding trajectory, identical or
    #  flipped for ONE query of the last layer on the 3-cam rig -- measured 0 and 1 in round 4)
    assert sum(vh["queries_over_1e-3"]) == 0 and sum(vh["argmax_mismatches"]) <= 1 and max(vh["box_max"]) < 1e-3


def test_random_rig_f16_value_streams_measured():
    """f16 value streams on the rig that amplifies rounding: inside the tolerance on decoder_f8.npz, outside the fp32 path's
    tail budget on the chaotic seed."""
    cfg = syn.F8
    g = _g("decoder_f8.npz")
    inputs = rig_inputs(cfg, int(g["seed"]))
    vh = run(cfg, g, inputs, False, value_dtype=torch.float16)
    print("decoder_f8.npz values f16:", vh)
    assert sum(vh["queries_over_1e-3"]) == 0 and sum(vh["argmax_mismatches"]) == 0
    vb = run(cfg, g, inputs, False, value_dtype=torch.bfloat16)
    print("decoder_f8.npz values bf16:", vb)
    assert vb["queries_over_1e-3"][5] > TAIL_QUERIES[5]
    del inputs
    torch.cuda.empty_cache()
    g = _g("decoder_f8_s1.npz")
    inputs = rig_inputs(cfg, int(g["seed"]))
    fp32 = run(cfg, g, inputs, False)
    vh = run(cfg, g, inputs, False, value_dtype=torch.float16)
    print("decoder_f8_s1.npz fp32:", fp32, "\nvalues f16:", vh)
    assert all(a <= b for a, b in zip(fp32["queries_over_1e-3"], TAIL_QUERIES))
    assert vh["queries_over_1e-3"][5] > TAIL_QUERIES[5] and max(vh["box_max"]) < 5e-2, \
        "f16 value streams now stay inside the tail budget on the chaotic seed: make them the BEV kernel's storage (DESIGN 3.2)"


# ------------------------------------------------------------------------------------------------ (iii) the built format
from parity import ARGMAX_MARGIN_INIT_RIG, decoder_parity, run_with_reference_views  # noqa: E402
from test_parity_gpu import run_decoder_gpu  # noqa: E402


@pytest.mark.parametrize("name,cfg", [("decoder_f8_init.npz", syn.F8), ("decoder_f8_3cam_init.npz", syn.F8_3CAM)])
def test_decoder_int16_block_value_streams_init_rig_literal(golden_dir, name, cfg):
    """The product decoder with `value_storage = "i16"` (quantiser + int16 BEV kernel) on the reference-initialised rig:
    north_star's criterion literally, as for fp32 storage."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    (cls, box, _), _ = run_with_reference_views(
        lambda force: run_decoder_gpu(cfg, seed, wseed, force, rig=(g, golden_dir), value_storage="i16"), g["views"], name + " (i16 values)")
    decoder_parity(cls, box, g["cls"], g["box"], what=name + " (i16 values)", tail_budget=None, argmax_margin=ARGMAX_MARGIN_INIT_RIG)


@pytest.mark.parametrize("name,cfg", [("decoder_f8.npz", syn.F8), ("decoder_f8_s1.npz", syn.F8), ("decoder_f8_3cam_s1.npz", syn.F8_3CAM)])
def test_decoder_int16_block_value_streams_random_rig_same_budget(golden_dir, name, cfg):
    """... and on the random-everything rig, chaotic seeds included, under the SAME tail budget and tie rule as the fp32 path."""
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    (cls, box, _), _ = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force, value_storage="i16"), g["views"],
                                                name + " (i16 values)")
    decoder_parity(cls, box, g["cls"], g["box"], what=name + " (i16 values)")


def test_decoder_int16_value_streams_producer_epilogues_equal_the_quantiser_launches():
    """`value_storage = "i16"` takes the two streams from their producers' own epilogues (rac_conv3x3_q16_fwd, rac_value_proj_q16_fwd);
    with `fused_q16_producers = False` the fp32 streams are written and quantised by rac_quant_i16_fwd launches.  Same bits out of
    six free-running layers (the epilogues ARE that quantiser applied to the same fp32 values)."""
    cfg, seed, wseed = syn.F8, 23, 24
    a = run_decoder_gpu(cfg, seed, wseed, value_storage="i16")
    b = run_decoder_gpu(cfg, seed, wseed, value_storage="i16", fused_q16_producers=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_head_small6_int16_value_streams_measured_gap(golden_dir):
    """Why int16 block storage of the value streams is opt-in and not the product default: on the reduced 30-query head fixture
    (random-everything rig, criterion LITERAL: no tail) the fp32 streams keep every query of every layer within 1e-3 (measured
    2.4e-4 ... 5.5e-4 at most in the last layer, depending on whether the reference's camera choices are imposed), the int16 streams
    leave the last layer at 8.6e-4 ... 1.1e-3 -- the free-running form of this comparison (test_head_forward_and_detections_vs_reference
    run with int16 streams) FAILS the literal criterion on one query -- and at five times the fp32 path's median error (2.7e-5
    against 5.3e-6).  Recorded as measured, with bounds on both sides."""
    from parity import head_boxes_normalised
    from test_parity_gpu import DEV, build_head
    cfg, name = syn.SMALL6, "head_small6.npz"
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    ref_n = head_boxes_normalised(g["all_bbox_preds"], cfg.pc_range)
    errs = {}
    for storage in ("f32", "i16"):
        head = build_head(cfg, g, seed, wseed)
        layer = head.transformer.decoder.decoder_layer
        layer.value_storage = storage
        layer.sampling.force_views = [torch.as_tensor(np.asarray(v)).to(DEV).contiguous() for v in g["views"]]
        with torch.no_grad():
            preds = head([f.to(DEV) for f in syn.make_pyramid(cfg, seed)], syn.make_bev(cfg, seed, 0).to(DEV),
                         syn.make_bev(cfg, seed, 1).to(DEV), syn.make_img_metas(cfg))
        torch.cuda.synchronize()
        eb = (head_boxes_normalised(preds["all_bbox_preds"].cpu(), cfg.pc_range) - torch.as_tensor(ref_n)).abs().amax(-1).flatten(1)
        errs[storage] = ([float(x) for x in eb.max(1).values], [float(x) for x in eb.median(1).values])
        assert (preds["all_cls_scores"].cpu().argmax(-1) == torch.as_tensor(g["all_cls_scores"]).argmax(-1)).all()
    print("head_small6 box error per layer (max, median):", errs)
    assert max(errs["f32"][0]) < 1e-3
    assert max(errs["i16"][0][:5]) < 1e-3 and errs["i16"][0][5] < 3e-3
    assert errs["i16"][1][5] > errs["f32"][1][5], "int16 value streams now as accurate as fp32 on this fixture: reconsider the default (DESIGN 3.11)"
