"""Pins the CPU oracle (oracle/restate.py + oracle/gather_ref.c) against golden vectors produced
by the reference's own CPU path (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import restate as R
from racformer_amd import synthetic as syn
from parity import ARGMAX_MARGIN_INIT_RIG, decoder_parity, init_rig_params, load_rig_state_dict, oracle_decoder, run_with_reference_views


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def t(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, atol, rtol=0.0, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    assert bool((err <= lim).all()), f"{what}: max err {err.max().item():.3e} (atol {atol}, rtol {rtol})"


def test_geometry(golden_dir):
    g = load(golden_dir, "geom_small.npz")
    q, off, xy, x = t(g["q"]), t(g["off"]), t(g["xy"]), t(g["x"])
    assert_close(R.decode_bbox(q, syn.PC_RANGE), g["decode_bbox"], 1e-6, 1e-6, "decode_bbox")
    assert_close(R.theta_d2xy(q), g["theta_d2xy"], 1e-6, 0, "theta_d2xy")
    assert_close(R.xy2theta_d(xy), g["xy2theta_d"], 1e-6, 0, "xy2theta_d")
    assert_close(R.denormalize_bbox(q), g["denormalize_bbox"], 1e-6, 1e-6, "denormalize")
    assert_close(R.make_sample_points(R.theta_d2xy(q), off, syn.PC_RANGE), g["make_sample_points"],
                 1e-5, 1e-6, "make_sample_points")
    assert_close(R.inverse_sigmoid(x), g["inverse_sigmoid"], 1e-6, 0, "inverse_sigmoid")
    assert_close(R.rotate_z(off, q[..., 6:7]), g["rotation"], 1e-6, 0, "rotation")


@pytest.mark.parametrize("tag,L", [("c2345", 4), ("c45", 2), ("c23456", 5)])
@pytest.mark.parametrize("force_torch", [False, True])
def test_msmv(golden_dir, tag, L, force_torch):
    g = load(golden_dir, "msmv_small.npz")
    feats = [t(g[f"{tag}_feat{i}"]) for i in range(L)]
    out = R.msmv_gather(feats, t(g[f"{tag}_loc"]), t(g[f"{tag}_w"]), force_torch=force_torch)
    # reference fallback interpolates the view axis trilinearly (weight ~1e-7 on the neighbour
    # view): equal within fp32 noise, not bit-identical.
    assert_close(out, g[f"{tag}_out"], 2e-5, 0, f"msmv {tag}")


def test_msmv_c_equals_torch(golden_dir):
    if R._clib() is None:
        pytest.skip("C oracle not built")
    g = load(golden_dir, "msmv_small.npz")
    feats = [t(g[f"c2345_feat{i}"]) for i in range(4)]
    a = R.msmv_gather(feats, t(g["c2345_loc"]), t(g["c2345_w"]))
    b = R.msmv_gather(feats, t(g["c2345_loc"]), t(g["c2345_w"]), force_torch=True)
    assert_close(a, b, 1e-6, 0, "C vs torch msmv")


def test_float64_instance_of_the_c_gathers(golden_dir):
    """The `_f64` instance of oracle/gather_ref.c (the arbiter of tools/fp64_arbiter.py: the same C text compiled for double) against
    the reference's own fp32 outputs: float64 operands select it, the result is float64 and agrees with the golden to the
    golden's own fp32 rounding -- and more closely with a float64 evaluation of the torch restatement than with the fp32 one."""
    if R._clib() is None:
        pytest.skip("C oracle not built")
    g = load(golden_dir, "msmv_small.npz")
    feats = [t(g[f"c2345_feat{i}"]).double() for i in range(4)]
    out = R.msmv_gather(feats, t(g["c2345_loc"]).double(), t(g["c2345_w"]).double())
    assert out.dtype == torch.float64
    assert_close(out, g["c2345_out"], 2e-5, 0, "msmv f64 instance")
    ref64 = R.msmv_gather_torch(feats, t(g["c2345_loc"]).double(), t(g["c2345_w"]).double()).double()
    assert float((out - ref64).abs().max()) < 1e-12
    m = load(golden_dir, "msda_small.npz")
    o = R.msda(t(m["value"]).double(), m["shapes"].tolist(), [0], t(m["loc"]).double(), t(m["attn"]).double())
    assert o.dtype == torch.float64
    assert_close(o, m["out"], 1e-5, 0, "msda f64 instance")


def test_sampling_4d_slot_quirk(golden_dir):
    g = load(golden_dir, "sampling4d_small.npz")
    feats = [t(g[f"feat{i}"]) for i in range(4)]
    H, W = [int(v) for v in g["image_hw"]]
    out = R.sampling_4d(t(g["pts"]), feats, t(g["scale_weights"]), t(g["lidar2img"]), H, W)
    assert_close(out, g["out"], 3e-5, 0, "sampling_4d")


@pytest.mark.parametrize("force_torch", [False, True])
def test_msda(golden_dir, force_torch):
    g = load(golden_dir, "msda_small.npz")
    out = R.msda(t(g["value"]), g["shapes"].tolist(), [0], t(g["loc"]), t(g["attn"]),
                 force_torch=force_torch)
    assert_close(out, g["out"], 1e-5, 0, "msda L=1")
    hw2 = g["shapes2"].tolist()
    out2 = R.msda(t(g["value2"]), hw2, [0, hw2[0][0] * hw2[0][1]], t(g["loc2"]), t(g["attn2"]),
                  force_torch=force_torch)
    assert_close(out2, g["out2"], 1e-5, 0, "msda L=2")


def _run_decoder(cfg, g, stages=None):
    """-> cls, box of the oracle with the camera choices of the fixture (its own, unless some differ: tests/parity.py)."""
    seed = int(g["seed"])
    sd = load_rig_state_dict(cfg, g, GOLDEN)
    qb, qf = syn.make_queries(cfg, seed)

    def run(force):
        if stages is not None:
            del stages[:]
        return oracle_decoder(R, sd, qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1),
                              syn.make_img_metas(cfg), cfg, stages, force)

    (cls, box, _), _ = run_with_reference_views(run, g["views"], "oracle")
    return cls, box


@pytest.mark.parametrize("name,cfg", [("decoder_small.npz", syn.SMALL), ("decoder_small6.npz", syn.SMALL6)])
def test_decoder_small(golden_dir, name, cfg):
    g = load(golden_dir, name)
    stages = []
    cls, box = _run_decoder(cfg, g, stages)
    for li, tol in ((0, 1e-4), (cfg.num_layers - 1, 1e-3)):
        for s in ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling",
                  "mixing", "ffn"):
            assert_close(stages[li][s], g[f"{s}_L{li}"], tol, tol, f"{s} L{li}")
    decoder_parity(cls, box, g["cls"], g["box"], what=name, tail_budget=None)


@pytest.mark.parametrize("name,cfg", [("decoder_f8.npz", syn.F8), ("decoder_f8_3cam.npz", syn.F8_3CAM)])
def test_decoder_f8(golden_dir, name, cfg):
    """Full f8 shapes (about 15 s of CPU each): north_star tolerance -- box regressions within
    1e-3, class argmax bit-exact -- for the oracle against the reference CPU forward."""
    g = load(golden_dir, name)
    torch.set_num_threads(min(16, os.cpu_count()))
    cls, box = _run_decoder(cfg, g)
    decoder_parity(cls, box, g["cls"], g["box"], what=name)


@pytest.mark.parametrize("name,cfg", [("decoder_f8_init.npz", syn.F8), ("decoder_f8_3cam_init.npz", syn.F8_3CAM)])
def test_decoder_f8_init_weights_rig_literal(golden_dir, name, cfg):
    """The oracle on SURVEY 8d's second rig (torch-constructor weights + the reference's own init_weights()): all six
    free-running layers literal -- every query within 1e-3, argmax identical -- against the reference's CPU forward."""
    g = load(golden_dir, name)
    torch.set_num_threads(min(16, os.cpu_count()))
    cls, box = _run_decoder(cfg, g)
    rows = decoder_parity(cls, box, g["cls"], g["box"], what=name, tail_budget=None, argmax_margin=ARGMAX_MARGIN_INIT_RIG)
    assert max(float(r["eb"].max()) for r in rows) < 2e-4        # measured 2.7e-5 / 4.2e-5: the rig does not amplify


def test_product_init_weights_reproduces_the_references(golden_dir):
    """RaCFormerTransformer.init_weights() of the product (the drop-in's counterpart of racformer_transformer.py:218-228,
    292-294, 355-358, 470-476, 577-578 and bev_self_attention.py:104-112) draws from torch's generator in the reference's
    order: under the seed the golden generator used it writes the very parameters the reference's init_weights() wrote,
    bit for bit, and touches no other."""
    from racformer_amd.transformer import RaCFormerTransformer
    cfg = syn.F8
    g = load(golden_dir, "decoder_f8_init.npz")
    want = init_rig_params(g, golden_dir)
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    syn.fill_params(tr, int(g["weight_seed"]), scheme="torch_default")
    before = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    torch.manual_seed(int(g["init_seed"]))
    tr.init_weights()
    after = tr.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    assert changed == sorted(want), (set(changed) ^ set(want))
    for k in changed:
        assert torch.equal(after[k], want[k]), k
