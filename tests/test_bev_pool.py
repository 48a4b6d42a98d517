"""bev_pool_v2: oracle and HIP kernels against the reference's own inline
known-answer vectors (models/csrc/bev_pool_v2/bev_pool.py:147-178) and against each other."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import restate as R


def _known(golden_dir, device="cpu"):
    k = json.load(open(os.path.join(golden_dir, "bev_pool_known_answer.json")))
    depth = torch.tensor(k["depth"], dtype=torch.float32, device=device).view(*k["depth_shape"]).requires_grad_()
    feat = torch.full(k["feat_shape"], k["feat_fill"], dtype=torch.float32, device=device).requires_grad_()
    rd, rf, rb = (torch.tensor(k[n], dtype=torch.int32, device=device) for n in ("ranks_depth", "ranks_feat", "ranks_bev"))
    return k, depth, feat, rd, rf, rb


def _intervals(rb):
    _, counts = torch.unique_consecutive(rb, return_counts=True)
    return (torch.cumsum(counts, 0) - counts).int(), counts.int()


def test_oracle_known_answer(golden_dir):
    k, depth, feat, rd, rf, rb = _known(golden_dir)
    starts, lengths = _intervals(rb)
    out = R.bev_pool_v2(depth, feat, rd, rf, rb, tuple(k["bev_feat_shape"]), starts, lengths)
    loss = out.sum()
    loss.backward()
    assert abs(loss.item() - k["loss"]) < 1e-6
    assert torch.allclose(depth.grad.reshape(-1), torch.tensor(k["grad_depth"]))
    assert torch.allclose(feat.grad.reshape(-1), torch.tensor(k["grad_feat"]))


@pytest.mark.gpu
def test_hip_known_answer(golden_dir):
    from racformer_amd.bev_pool import bev_pool_v2, intervals_from_ranks
    k, depth, feat, rd, rf, rb = _known(golden_dir, "cuda:0")
    starts, lengths = intervals_from_ranks(rb)
    out = bev_pool_v2(depth, feat, rd, rf, rb, tuple(k["bev_feat_shape"]), starts, lengths)
    assert out.shape == (1, 2, 1, 2, 2)
    loss = out.sum()
    loss.backward()
    assert abs(loss.item() - k["loss"]) < 1e-6                       # the reference asserts loss == 4.4
    assert torch.allclose(depth.grad.reshape(-1).cpu(), torch.tensor(k["grad_depth"]))
    assert torch.allclose(feat.grad.reshape(-1).cpu(), torch.tensor(k["grad_feat"]))


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [12, 3])
@pytest.mark.parametrize("c", [256, 128, 64, 6, 32, 80, 192, 320, 4])
def test_hip_vs_oracle_random(c, grid):
    """LSS-shaped random case: B=1, N=2 cams, D=8 depth bins, 6x10 feature map -> 12x12 BEV grid; ragged
    intervals including empty BEV cells; c=6 exercises the scalar-channel fallback.  c = 32 / 80 / 192 / 320 / 4 are not
    multiples of 4 x (lanes of a group): the last channel pass leaves part of the group without a channel quad, and with the
    3x3 grid the intervals (mean 64 points) are longer than the lanes that still have one (BEVDet's numC_Trans = 80 is such
    a width; round 4's kernel dropped those points)."""
    from racformer_amd.bev_pool import bev_pool_v2, intervals_from_ranks
    rng = np.random.default_rng(c)
    B, N, D, H, W, Z, Y, X = 1, 2, 8, 6, 10, 1, grid, grid
    depth = torch.from_numpy(rng.random((B, N, D, H, W), dtype=np.float32))
    feat = torch.from_numpy(rng.standard_normal((B, N, H, W, c), dtype=np.float32))
    n_pts = B * N * D * H * W
    keep = rng.random(n_pts) < 0.6                                     # points that fall inside the BEV grid
    rd = np.nonzero(keep)[0].astype(np.int32)
    rf = (rd // (D * H * W)) * (H * W) + rd % (H * W)                  # same (cam,h,w) feature for every depth bin
    rb = rng.integers(0, B * Z * Y * X, size=rd.shape[0]).astype(np.int32)
    order = np.argsort(rb, kind="stable")
    rd, rf, rb = (torch.from_numpy(a[order].astype(np.int32)) for a in (rd, rf, rb))
    starts, lengths = _intervals(rb)
    shape = (B, Z, Y, X, c)
    d0, f0 = depth.clone().requires_grad_(), feat.clone().requires_grad_()
    ref = R.bev_pool_v2(d0, f0, rd, rf, rb, shape, starts, lengths)
    gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    ref.backward(gout)
    dev = "cuda:0"
    d1, f1 = depth.to(dev).requires_grad_(), feat.to(dev).requires_grad_()
    gs, gl = intervals_from_ranks(rb.to(dev))
    out = bev_pool_v2(d1, f1, rd.to(dev), rf.to(dev), rb.to(dev), shape, gs, gl)
    out.backward(gout.to(dev))
    assert (out.cpu() - ref).abs().max().item() < 1e-5
    assert (d1.grad.cpu() - d0.grad).abs().max().item() < 1e-4
    assert (f1.grad.cpu() - f0.grad).abs().max().item() < 1e-4
