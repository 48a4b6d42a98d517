"""Drop-in for the reference's ``models/csrc/wrapper.py`` (msmv sampling operator surface):
``msmv_sampling``, ``MSMVSamplingC2345/C45/C23456`` and the ``MSMV_CUDA`` flag, on top of
``rac_msmv_fwd`` (hand-written HIP, racformer_amd/csrc/msmv_fwd.hip).

Same names, argument meaning and error behaviour as the reference (wrapper.py:78-153,
msmv_sampling.cpp:132-184): features channel-last ``[B', N, H, W, C]``, contiguous device tensors,
``RuntimeError`` on non-contiguous / non-device inputs and on ``P > 128``; the result is a new
``[B', Q, C, P]`` float32 tensor.  ``backward`` runs rac_msmv_bwd (fp32 features).
"""
import ctypes

import torch

from . import _lib

MSMV_CUDA = True  # the HIP operator is the only path; there is no torch fallback in this package


def msmv_forward(mlvl_feats, sampling_locations, scale_weights, out_layout=_lib.OUT_SQCP,
                 num_frames=1, num_groups=1, out=None):
    """Launches rac_msmv_fwd on the current stream.  ``out_layout=OUT_BQGTPC`` writes
    ``[B, Q, G, T*P, C]`` directly (what sampling_4d returns, sparsebev_sampling.py:128-131)."""
    feats = list(mlvl_feats)
    L = len(feats)
    _lib.require_gpu(*feats, sampling_locations, scale_weights, what="msmv_sampling")
    S, N, H0, W0, C = feats[0].shape
    _, Q, P, three = sampling_locations.shape
    if three != 3 or sampling_locations.shape[0] != S:
        raise RuntimeError("msmv_sampling: sampling_locations must be [B', Q, P, 3]")
    if tuple(scale_weights.shape) != (S, Q, P, L):
        raise RuntimeError(f"msmv_sampling: scale_weights must be [B', Q, P, {L}], got {tuple(scale_weights.shape)}")
    if P > 128:
        raise RuntimeError("num_point exceed limits")
    code = _lib.dtype_code(feats[0])
    for f in feats:
        if f.dtype != feats[0].dtype or f.shape[0] != S or f.shape[1] != N or f.shape[4] != C:
            raise RuntimeError("msmv_sampling: all levels must share dtype and [B', N, ., ., C]")
    if sampling_locations.dtype != torch.float32 or scale_weights.dtype != torch.float32:
        raise RuntimeError("msmv_sampling: locations / weights must be float32")
    if out_layout == _lib.OUT_SQCP:
        shape = (S, Q, C, P)
    else:
        B = S // (num_frames * num_groups)
        shape = (B, Q, num_groups, num_frames * P, C)
    if out is None:
        out = torch.empty(shape, device=feats[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hw = (ctypes.c_int32 * (2 * L))(*[int(x) for f in feats for x in f.shape[2:4]])
    ev = _lib.timer.record("msmv_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_msmv_fwd(ptrs, hw, L, _lib.ptr(sampling_locations), _lib.ptr(scale_weights),
                                 _lib.ptr(out), S, N, Q, P, C, code, out_layout, num_frames, num_groups,
                                 _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_msmv_fwd")
    if _lib.timer is not None and getattr(_lib.timer, "capture_inputs", False):
        _lib.timer.captured.append((sampling_locations.detach(), [tuple(f.shape) for f in feats]))
    return out


def msmv_backward(grad_output, mlvl_feats, sampling_locations, scale_weights):
    """rac_msmv_bwd: -> (grad_feats (list), grad_sampling_locations, grad_scale_weights), the tuple the
    reference's ``_ms_deform_attn_cuda_*_backward`` returns (msmv_sampling.cpp:302-497).  fp32 only."""
    feats = list(mlvl_feats)
    L = len(feats)
    grad_output = grad_output.contiguous()
    _lib.require_gpu(grad_output, *feats, sampling_locations, scale_weights, what="msmv_sampling backward")
    if any(f.dtype != torch.float32 for f in feats):
        raise RuntimeError("msmv_sampling backward: float32 features only")
    S, N, _, _, C = feats[0].shape
    _, Q, P, _ = sampling_locations.shape
    grad_feats = [torch.zeros_like(f) for f in feats]
    grad_loc = torch.empty_like(sampling_locations)
    grad_w = torch.empty_like(scale_weights)
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    gptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in grad_feats])
    hw = (ctypes.c_int32 * (2 * L))(*[int(x) for f in feats for x in f.shape[2:4]])
    rc = _lib.lib().rac_msmv_bwd(_lib.ptr(grad_output), ptrs, hw, L, _lib.ptr(sampling_locations),
                                 _lib.ptr(scale_weights), gptrs, _lib.ptr(grad_loc), _lib.ptr(grad_w), S, N, Q, P, C,
                                 _lib.stream_ptr())
    _lib.check(rc, "rac_msmv_bwd")
    return grad_feats, grad_loc, grad_w


class _MSMVBase(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *args):
        *feats, sampling_locations, scale_weights = args
        ctx.save_for_backward(*feats, sampling_locations, scale_weights)
        return msmv_forward(feats, sampling_locations, scale_weights)

    @staticmethod
    def backward(ctx, grad_output):
        *feats, sampling_locations, scale_weights = ctx.saved_tensors
        grad_feats, grad_loc, grad_w = msmv_backward(grad_output, feats, sampling_locations, scale_weights)
        return (*grad_feats, grad_loc, grad_w)


class MSMVSamplingC2345(_MSMVBase):
    """wrapper.py:78-97 -- apply(feat_c2, feat_c3, feat_c4, feat_c5, sampling_locations, scale_weights)"""


class MSMVSamplingC45(_MSMVBase):
    """wrapper.py:99-118 -- apply(feat_c4, feat_c5, sampling_locations, scale_weights)"""


class MSMVSamplingC23456(_MSMVBase):
    """wrapper.py:120-142 -- apply(feat_c2, ..., feat_c6, sampling_locations, scale_weights)"""


def msmv_sampling(mlvl_feats, sampling_locations, scale_weights):
    """wrapper.py:145-153.  Any level count 1..8 is served by the HIP operator."""
    if len(mlvl_feats) == 2:
        return MSMVSamplingC45.apply(*mlvl_feats, sampling_locations, scale_weights)
    if len(mlvl_feats) == 4:
        return MSMVSamplingC2345.apply(*mlvl_feats, sampling_locations, scale_weights)
    if len(mlvl_feats) == 5:
        return MSMVSamplingC23456.apply(*mlvl_feats, sampling_locations, scale_weights)
    return msmv_forward(mlvl_feats, sampling_locations, scale_weights)
