"""Parity of the HIP operators (through the C-ABI) against the CPU oracle and the reference
goldens.  Runs on the GPU box only (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from racformer_amd import _lib, synthetic as syn
from racformer_amd.msda import MultiScaleDeformableAttnFunction_fp32, msda_forward
from racformer_amd.msmv import MSMVSamplingC2345, msmv_forward, msmv_sampling

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


# ----------------------------------------------------------------------------------- msmv
@pytest.mark.parametrize("tag,L", [("c2345", 4), ("c45", 2), ("c23456", 5)])
def test_msmv_golden_generic_path(golden_dir, tag, L):
    """C=8 fixtures from the reference's own CPU path -> generic kernel."""
    g = np.load(os.path.join(golden_dir, "msmv_small.npz"))
    feats = [t(g[f"{tag}_feat{i}"]).to(DEV) for i in range(L)]
    out = msmv_sampling(feats, t(g[f"{tag}_loc"]).to(DEV), t(g[f"{tag}_w"]).to(DEV))
    assert out.shape == g[f"{tag}_out"].shape
    assert maxerr(out, g[f"{tag}_out"]) < 2e-5   # reference fallback is trilinear in the view axis


def _rand_case(seed, S, N, Q, P, C, hws, dtype=torch.float32):
    rng = np.random.default_rng(seed)
    feats = [t(rng.standard_normal((S, N, h, w, C), dtype=np.float32)) for h, w in hws]
    loc = rng.random((S, Q, P, 3), dtype=np.float32) * 1.1 - 0.05
    loc[..., 2] = rng.integers(0, N, size=(S, Q, P)).astype(np.float32) / np.float32(max(N - 1, 1))
    if Q and P:
        loc[0, 0, 0, :2] = (0.0, 0.0)
        loc[0, 0, P - 1, :2] = (1.0, 1.0)
        loc[0, Q - 1, 0, :2] = (-1e5, 0.5)
        loc[S - 1, Q - 1, P - 1, :2] = (1.0 + 1e-3, -1e-3)
    w = rng.random((S, Q, P, len(hws)), dtype=np.float32)
    if dtype == torch.bfloat16:
        feats = [f.to(torch.bfloat16).float() for f in feats]   # oracle sees the rounded values
    return feats, t(loc), t(w)


@pytest.mark.parametrize("S,N,Q,P,L", [(3, 2, 5, 12, 4), (9, 6, 7, 3, 4), (2, 3, 4, 1, 2), (8, 1, 9, 13, 5),
                                       (1, 6, 1, 128, 4)])
@pytest.mark.parametrize("layout", [0, 1])
def test_msmv_c64_fast_path(S, N, Q, P, L, layout):
    """Ragged Q (not a multiple of 4), P not a multiple of 4, P at the 128 limit, 1 view."""
    hws = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)][:L]
    feats, loc, w = _rand_case(S * 100 + P, S, N, Q, P, 64, hws)
    ref = R.msmv_gather(feats, loc, w)                                   # [S,Q,C,P]
    T_, G_ = (1, S) if layout else (1, 1)
    out = msmv_forward([f.to(DEV) for f in feats], loc.to(DEV), w.to(DEV), out_layout=layout,
                       num_frames=T_, num_groups=G_)
    if layout:   # [B=1,Q,G=S,T*P,C] -> [S,Q,C,P]
        out = out.reshape(1, Q, S, 1, P, 64).permute(0, 3, 2, 1, 5, 4).reshape(S, Q, 64, P)
    assert maxerr(out, ref) < 1e-5


def test_msmv_bf16_features():
    hws = [(12, 20), (6, 10), (3, 5), (2, 3)]
    feats, loc, w = _rand_case(5, 4, 3, 6, 12, 64, hws, dtype=torch.bfloat16)
    ref = R.msmv_gather(feats, loc, w)
    out = msmv_forward([f.to(DEV).to(torch.bfloat16) for f in feats], loc.to(DEV), w.to(DEV))
    assert maxerr(out, ref) < 1e-5     # same bf16-rounded inputs, fp32 arithmetic


def test_msmv_empty_and_errors():
    hws = [(4, 6), (2, 3)]
    feats, loc, w = _rand_case(1, 2, 2, 0, 5, 64, hws)
    out = msmv_forward([f.to(DEV) for f in feats], loc.to(DEV), w.to(DEV))
    assert out.shape == (2, 0, 64, 5)
    feats, loc, w = _rand_case(2, 2, 2, 3, 4, 64, hws)
    gf = [f.to(DEV) for f in feats]
    with pytest.raises(RuntimeError, match="contiguous"):
        msmv_sampling(gf, loc.to(DEV)[:, :, ::2], w.to(DEV)[:, :, ::2])
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        msmv_sampling(gf, loc, w.to(DEV))
    big = torch.zeros(2, 3, 129, 3, device=DEV)
    with pytest.raises(RuntimeError, match="num_point exceed limits"):
        msmv_sampling(gf, big, torch.zeros(2, 3, 129, 2, device=DEV))
    out = MSMVSamplingC2345.apply(*(gf + gf), loc.to(DEV), torch.cat([w, w], -1).to(DEV).contiguous())
    assert out.shape == (2, 3, 64, 4)


def test_msmv_f8_full_size_vs_oracle():
    """BASELINE config 2: 900 queries x 6 cams x 4 levels, S=32 slots, fp32, uniform stress set."""
    cfg = syn.F8
    S, N, Q, P, C = 32, 6, 900, 12, 64
    rng = np.random.default_rng(0)
    feats = [t(syn.smooth_noise(70 + i, (S, N), h, w * C).reshape(S, N, h, w, C)) for i, (h, w) in enumerate(cfg.fpn_hw)]
    loc = rng.random((S, Q, P, 3), dtype=np.float32) * 1.1 - 0.05
    loc[..., 2] = rng.integers(0, N, size=(S, Q, P)).astype(np.float32) / np.float32(N - 1)
    w = rng.standard_normal((S, Q, P, 4), dtype=np.float32)
    w = np.exp(w) / np.exp(w).sum(-1, keepdims=True)
    loc, w = t(loc), t(w.astype(np.float32))
    ref = R.msmv_gather(feats, loc, w)
    gf = [f.to(DEV) for f in feats]
    out0 = msmv_forward(gf, loc.to(DEV), w.to(DEV))
    assert maxerr(out0, ref) < 2e-5
    out1 = msmv_forward(gf, loc.to(DEV), w.to(DEV), out_layout=1, num_frames=8, num_groups=4)
    ref1 = ref.reshape(1, 8, 4, Q, C, P).permute(0, 3, 2, 1, 5, 4).flatten(3, 4)
    assert maxerr(out1, ref1) < 2e-5
    # size-independent properties: linearity in the weights, zero weights -> zeros
    out2 = msmv_forward(gf, loc.to(DEV), (2 * w).to(DEV))
    assert torch.allclose(out2, 2 * out0, rtol=1e-6, atol=1e-6)
    assert msmv_forward(gf, loc.to(DEV), torch.zeros_like(w).to(DEV)).abs().max().item() == 0.0
    far = loc.clone()
    far[..., 0] = 5.0    # every point outside every map -> exact zeros (zero padding)
    assert msmv_forward(gf, far.to(DEV), w.to(DEV)).abs().max().item() == 0.0


# ----------------------------------------------------------------------------------- msda
def test_msda_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "msda_small.npz"))
    out = MultiScaleDeformableAttnFunction_fp32.apply(
        t(g["value"]).to(DEV), t(g["shapes"]).to(DEV), torch.tensor([0], device=DEV), t(g["loc"]).to(DEV),
        t(g["attn"]).to(DEV), 64)
    assert maxerr(out, g["out"]) < 1e-5
    hw2 = g["shapes2"].tolist()
    out2 = msda_forward(t(g["value2"]).to(DEV), hw2, [0, hw2[0][0] * hw2[0][1]], t(g["loc2"]).to(DEV),
                        t(g["attn2"]).to(DEV))
    assert maxerr(out2, g["out2"]) < 1e-5


@pytest.mark.parametrize("bs,Q,heads,L,P", [(8, 37, 4, 1, 20), (3, 5, 3, 2, 7), (9, 1, 1, 1, 1), (2, 900, 4, 1, 20)])
def test_msda_d64_fast_path(bs, Q, heads, L, P):
    rng = np.random.default_rng(bs * 10 + P)
    hws = [(16, 12), (5, 7)][:L]
    keys = sum(h * w for h, w in hws)
    starts = [0, hws[0][0] * hws[0][1]][:L]
    value = t(rng.standard_normal((bs, keys, heads, 64), dtype=np.float32))
    loc = t(rng.random((bs, Q, heads, L, P, 2), dtype=np.float32) * 1.2 - 0.1)
    attn = t(rng.random((bs, Q, heads, L, P), dtype=np.float32))
    ref = R.msda(value, hws, starts, loc, attn)
    out = msda_forward(value.to(DEV), hws, starts, loc.to(DEV), attn.to(DEV))
    assert maxerr(out, ref) < 1e-5
    outb = msda_forward(value.to(DEV).to(torch.bfloat16), hws, starts, loc.to(DEV), attn.to(DEV))
    refb = R.msda(value.to(torch.bfloat16).float(), hws, starts, loc, attn)
    assert maxerr(outb, refb) < 1e-5


def test_msda_f8_full_size_vs_oracle():
    bs, Q, heads, P, H, W = 8, 900, 4, 20, 128, 128
    rng = np.random.default_rng(3)
    value = t(syn.smooth_noise(5, (bs,), H, W * 256).reshape(bs, H * W, heads, 64))
    loc = t(rng.random((bs, Q, heads, 1, P, 2), dtype=np.float32) * 1.1 - 0.05)
    attn = rng.random((bs, Q, heads, 1, P), dtype=np.float32)
    attn = t(attn / attn.sum(-1, keepdims=True))
    ref = R.msda(value, [[H, W]], [0], loc, attn)
    out = msda_forward(value.to(DEV), [[H, W]], [0], loc.to(DEV), attn.to(DEV))
    assert maxerr(out, ref) < 1e-5
    with pytest.raises(RuntimeError, match="exceeds keys"):
        msda_forward(value.to(DEV), [[H, W + 1]], [0], loc.to(DEV), attn.to(DEV))


# ----------------------------------------------------------------------------------- regroup
@pytest.mark.parametrize("cfg", [syn.SMALL, syn.SMALL6])
def test_regroup(cfg):
    from racformer_amd.transformer import regroup_pyramid
    feats = syn.make_pyramid(cfg, 3)
    ref = R.regroup_pyramid(feats, cfg.num_cams)
    out = regroup_pyramid([f.to(DEV) for f in feats], cfg.num_cams)
    for a, b in zip(out, ref):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)      # pure data movement: bit-exact
    outb = regroup_pyramid([f.to(DEV) for f in feats], cfg.num_cams, out_dtype=torch.bfloat16)
    for a, b in zip(outb, ref):
        assert torch.equal(a.cpu(), b.to(torch.bfloat16))
