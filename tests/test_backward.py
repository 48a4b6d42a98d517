"""Backward of the two gather operators: the oracle's autograd against the
reference CPU path's gradients (CPU), and rac_msmv_bwd / rac_msda_bwd against both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R


def t(a, grad=False):
    x = torch.from_numpy(np.asarray(a)).clone()
    return x.requires_grad_() if grad else x


def _oracle_msmv_grads(feats, loc, w, gout):
    feats = [f.clone().requires_grad_() for f in feats]
    loc, w = loc.clone().requires_grad_(), w.clone().requires_grad_()
    out = R.msmv_gather_torch(feats, loc, w)
    (out * gout).sum().backward()
    return out.detach(), [f.grad for f in feats], loc.grad, w.grad


def _oracle_msda_grads(value, shapes, starts, loc, attn, gout):
    value, loc, attn = value.clone().requires_grad_(), loc.clone().requires_grad_(), attn.clone().requires_grad_()
    out = R.msda_torch(value, shapes, starts, loc, attn)
    (out * gout).sum().backward()
    return out.detach(), value.grad, loc.grad, attn.grad


def test_oracle_autograd_vs_reference_gradients(golden_dir):
    g = np.load(os.path.join(golden_dir, "backward_small.npz"))
    feats = [t(g[f"feat{i}"]) for i in range(4)]
    out, gf, gl, gw = _oracle_msmv_grads(feats, t(g["loc"]), t(g["w"]), t(g["gout"]))
    assert (out - t(g["out"])).abs().max().item() < 2e-5
    for i in range(4):
        assert (gf[i] - t(g[f"gfeat{i}"])).abs().max().item() < 2e-5
    # the fallback's 5-D grid_sample also differentiates along the view axis; the operator does not
    assert (gl[..., :2] - t(g["gloc"])[..., :2]).abs().max().item() < 2e-4
    assert gl[..., 2].abs().max().item() == 0.0
    assert (gw - t(g["gw"])).abs().max().item() < 2e-5
    _, gv, gml, ga = _oracle_msda_grads(t(g["value"]), g["mshape"].tolist(), [0], t(g["mloc"]), t(g["attn"]), t(g["mgout"]))
    assert (gv - t(g["gvalue"])).abs().max().item() < 2e-5
    assert (gml - t(g["gmloc"])).abs().max().item() < 2e-4
    assert (ga - t(g["gattn"])).abs().max().item() < 2e-5


@pytest.mark.gpu
def test_msmv_backward_golden_and_fast_path(golden_dir):
    from racformer_amd.msmv import msmv_sampling
    dev = "cuda:0"
    g = np.load(os.path.join(golden_dir, "backward_small.npz"))
    feats = [t(g[f"feat{i}"]).to(dev).requires_grad_() for i in range(4)]
    loc, w = t(g["loc"]).to(dev).requires_grad_(), t(g["w"]).to(dev).requires_grad_()
    out = msmv_sampling(feats, loc, w)                                  # C=8 -> generic kernels
    (out * t(g["gout"]).to(dev)).sum().backward()
    for i in range(4):
        assert (feats[i].grad.cpu() - t(g[f"gfeat{i}"])).abs().max().item() < 2e-5
    assert (loc.grad.cpu()[..., :2] - t(g["gloc"])[..., :2]).abs().max().item() < 2e-4
    assert (w.grad.cpu() - t(g["gw"])).abs().max().item() < 2e-5
    # C=64 fast path against the oracle's autograd (ragged sizes, stress locations)
    rng = np.random.default_rng(5)
    S, N, Q, P = 3, 2, 5, 7
    hws = [(12, 20), (6, 10), (3, 5), (2, 3)]
    cf = [t(rng.standard_normal((S, N, h, ww, 64), dtype=np.float32)) for h, ww in hws]
    cl = rng.random((S, Q, P, 3), dtype=np.float32) * 1.1 - 0.05
    cl[..., 2] = rng.integers(0, N, size=(S, Q, P)).astype(np.float32) / np.float32(N - 1)
    cl, cw = t(cl), t(rng.random((S, Q, P, 4), dtype=np.float32))
    go = t(rng.standard_normal((S, Q, 64, P), dtype=np.float32))
    _, rgf, rgl, rgw = _oracle_msmv_grads(cf, cl, cw, go)
    gfeats = [f.to(dev).requires_grad_() for f in cf]
    gl_, gw_ = cl.to(dev).requires_grad_(), cw.to(dev).requires_grad_()
    (msmv_sampling(gfeats, gl_, gw_) * go.to(dev)).sum().backward()
    for i in range(4):
        assert (gfeats[i].grad.cpu() - rgf[i]).abs().max().item() < 5e-5
    assert (gl_.grad.cpu() - rgl).abs().max().item() < 5e-4
    assert (gw_.grad.cpu() - rgw).abs().max().item() < 5e-5


@pytest.mark.gpu
def test_msda_backward_golden_and_fast_path(golden_dir):
    from racformer_amd.msda import MultiScaleDeformableAttnFunction_fp32 as F32
    dev = "cuda:0"
    g = np.load(os.path.join(golden_dir, "backward_small.npz"))
    v, l, a = (t(g[k]).to(dev).requires_grad_() for k in ("value", "mloc", "attn"))
    out = F32.apply(v, t(g["mshape"]).to(dev), torch.tensor([0], device=dev), l, a, 64)   # dim=8 -> generic
    (out * t(g["mgout"]).to(dev)).sum().backward()
    assert (v.grad.cpu() - t(g["gvalue"])).abs().max().item() < 2e-5
    assert (l.grad.cpu() - t(g["gmloc"])).abs().max().item() < 2e-4
    assert (a.grad.cpu() - t(g["gattn"])).abs().max().item() < 2e-5
    rng = np.random.default_rng(9)
    bs, Q, heads, P = 3, 6, 4, 5
    hws = [[8, 6], [4, 3]]
    starts = [0, 48]
    value = t(rng.standard_normal((bs, 60, heads, 64), dtype=np.float32))
    loc = t(rng.random((bs, Q, heads, 2, P, 2), dtype=np.float32) * 1.2 - 0.1)
    attn = t(rng.random((bs, Q, heads, 2, P), dtype=np.float32))
    go = t(rng.standard_normal((bs, Q, heads * 64), dtype=np.float32))
    _, rgv, rgl, rga = _oracle_msda_grads(value, hws, starts, loc, attn, go)
    v, l, a = value.to(dev).requires_grad_(), loc.to(dev).requires_grad_(), attn.to(dev).requires_grad_()
    out = F32.apply(v, torch.tensor(hws, device=dev), torch.tensor(starts, device=dev), l, a, 64)
    (out * go.to(dev)).sum().backward()
    assert (v.grad.cpu() - rgv).abs().max().item() < 5e-5
    assert (l.grad.cpu() - rgl).abs().max().item() < 5e-4
    assert (a.grad.cpu() - rga).abs().max().item() < 5e-5
