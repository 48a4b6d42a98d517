"""Host-side logic of the product modules (keypoint generation, projection / view selection, slot
regrouping, hoisting, refine) checked on CPU against the reference goldens.  The three HIP entry
points are replaced HERE, in the test, by the oracle's gathers -- the product has no such path and
raises on CPU tensors (see test_no_cpu_fallback)."""
import os

import numpy as np
import pytest
import torch

import plans  # noqa: F401  (registers the reference's op decomposition: decoder_layer.fused = False)
from oracle import restate as R
from parity import decoder_parity
from racformer_amd import synthetic as syn
from racformer_amd import transformer as T


def _oracle_msmv(feats, loc, w, out_layout=0, num_frames=1, num_groups=1, out=None):
    o = R.msmv_gather(list(feats), loc, w)                     # [S,Q,C,P]
    if out_layout == 0:
        return o
    S, Q, C, P = o.shape
    B = S // (num_frames * num_groups)
    return o.reshape(B, num_frames, num_groups, Q, C, P).permute(0, 3, 2, 1, 5, 4).flatten(3, 4).contiguous()


def _oracle_msda(value, shapes, starts, loc, attn, out=None):
    return R.msda(value, shapes, starts, loc, attn)


def _oracle_regroup(feats, num_cams, groups=4, out_dtype=torch.float32):
    return R.regroup_pyramid(feats, num_cams, groups)


@pytest.fixture
def host_ops(monkeypatch):
    monkeypatch.setattr(T, "msmv_forward", _oracle_msmv)
    monkeypatch.setattr(T, "msda_forward", _oracle_msda)
    monkeypatch.setattr(T, "regroup_pyramid", _oracle_regroup)


@pytest.mark.parametrize("name,cfg", [("decoder_small.npz", syn.SMALL), ("decoder_small6.npz", syn.SMALL6)])
def test_decoder_host_logic(golden_dir, host_ops, name, cfg):
    g = np.load(os.path.join(golden_dir, name))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    tr = T.RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    tr.decoder.decoder_layer.fused = False   # host logic of the op-decomposed plan
    syn.fill_params(tr, wseed)
    qb, qf = syn.make_queries(cfg, seed)
    stages = []
    with torch.no_grad():
        cls, box = tr(qb, qf, syn.make_pyramid(cfg, seed), syn.make_bev(cfg, seed, 0), syn.make_bev(cfg, seed, 1),
                      None, syn.make_img_metas(cfg), stages_per_layer=stages)
    for s in ("position_encoder", "self_attn", "sampling_radar_bev", "sampling_lss_bev", "sampling", "mixing", "ffn"):
        err = (stages[0][s] - torch.from_numpy(g[f"{s}_L0"])).abs().max().item()
        assert err < 1e-4, (s, err)
    decoder_parity(cls, box, g["cls"], g["box"], what=name)


def test_state_dict_contract():
    for cfg in (syn.SMALL, syn.F8, syn.F8_3CAM):
        tr = T.RaCFormerTransformer(**cfg.transformer_kwargs())
        mine = {k: tuple(v.shape) for k, v in tr.state_dict().items()}
        assert mine == syn.transformer_param_shapes(cfg)


def test_no_cpu_fallback():
    cfg = syn.SMALL
    tr = T.RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    qb, qf = syn.make_queries(cfg, 1)
    with torch.no_grad(), pytest.raises(RuntimeError, match="CUDA tensor|not built"):
        tr(qb, qf, syn.make_pyramid(cfg, 1), syn.make_bev(cfg, 1, 0), syn.make_bev(cfg, 1, 1), None,
           syn.make_img_metas(cfg))


def test_mixing_split_k_out_proj_is_equivalent():
    """The split-K (batched) form of AdaptiveMixing.out_proj is the same linear map."""
    torch.manual_seed(0)
    mix = T.AdaptiveMixing(in_dim=256, in_points=12, n_groups=4, out_points=128).eval()
    x, q = torch.randn(1, 5, 4, 12, 64), torch.randn(1, 5, 256)
    with torch.no_grad():
        a = mix(x, q)
        b = mix(x, q, mix.split_out_proj())
    assert (a - b).abs().max().item() < 1e-4


def test_rig_coverage_selects_the_sampling_kernel_variant():
    """The decoder measures, on the staged sample's own projection matrices, which share of a ring of probe points some camera
    sees and picks the sampling kernel's variant from that -- not from the number of cameras."""
    import numpy as np
    from racformer_amd.transformer import compact_variant, rig_coverage
    cov6 = rig_coverage(np.asarray(syn.make_img_metas(syn.F8)[0]["lidar2img"]), 6, syn.F8.image_hw, syn.F8.pc_range)
    cov3 = rig_coverage(np.asarray(syn.make_img_metas(syn.F8_3CAM)[0]["lidar2img"]), 3, syn.F8_3CAM.image_hw, syn.F8_3CAM.pc_range)
    assert cov6 > 0.9 and 0.3 < cov3 < 0.6
    assert compact_variant(cov6) is False and compact_variant(cov3) is True and compact_variant(None) is None
    failed = np.asarray(syn.make_img_metas(syn.F8)[0]["lidar2img"]).copy()
    failed[[0, 2, 4]] = 0.0                                    # three of the six cameras deliver nothing
    assert compact_variant(rig_coverage(failed, 6, syn.F8.image_hw, syn.F8.pc_range)) is True


def test_scratch_namespaces_and_fpn_writer_contract():
    """Host logic of round 3's additions: scratch namespaces nest and restore (plans in flight own their buffers), the FPN
    output writer carries mmdet's FPN.fpn_convs state-dict keys, and both refuse host tensors (no CPU fallback)."""
    from racformer_amd import fused
    from racformer_amd.fpn_writer import FPNOutputWriter
    assert fused._scratch_ns[0] is None
    with fused.scratch_namespace("a"):
        assert fused._scratch_ns[0] == "a"
        with fused.scratch_namespace(("b", 1)):
            assert fused._scratch_ns[0] == ("b", 1)
        assert fused._scratch_ns[0] == "a"
    assert fused._scratch_ns[0] is None
    fused._conv_images[(1, 2, 3, 32, "cpu", "gone")] = object()
    fused.release_scratch("gone")
    assert not any(k[-1] == "gone" for k in fused._conv_images)
    wr = FPNOutputWriter(num_levels=4, num_cams=6)
    keys = set(wr.state_dict())
    assert keys == {f"fpn_convs.{i}.conv.{n}" for i in range(4) for n in ("weight", "bias")}
    assert tuple(wr.fpn_convs[0].conv.weight.shape) == (256, 256, 3, 3) and wr.fpn_convs[0].conv.padding == (1, 1)
    with pytest.raises(RuntimeError, match="CUDA tensor|no CPU fallback|HIP library"):
        wr([torch.zeros(6, 256, 4, 8)])
    with pytest.raises(ValueError):
        FPNOutputWriter(channels=128)
