"""Drop-in for ``models/multi_scale_deformable_attn_function.py`` of the reference:
``MultiScaleDeformableAttnFunction_fp32`` / ``_fp16`` with the same ``apply`` signature
(:93-128), on top of ``rac_msda_fwd`` (hand-written HIP, racformer_amd/csrc/msda_fwd.hip).
Forward and backward (fp32)."""
import ctypes

import torch

from . import _lib


_host_cache = {}   # (data_ptr, version, numel, device) of a shape tensor -> (tensor kept alive, host copy)


def _host_i64(x):
    """spatial_shapes / level_start_index as a host int64 array.  The reference passes device tensors
    (bev_self_attention.py:189-201); reading one back synchronises the host with the stream, so the copy is made once per
    tensor (identity + in-place version; the entry holds the tensor, so its address cannot be reused) and every later call
    with that tensor is sync-free."""
    if isinstance(x, torch.Tensor):
        key = (x.data_ptr(), x._version, x.numel(), str(x.device))
        hit = _host_cache.get(key)
        if hit is None:
            if len(_host_cache) >= 64:
                _host_cache.clear()
            hit = _host_cache[key] = (x, x.detach().cpu().reshape(-1).tolist())
        x = hit[1]
    else:
        x = [int(v) for row in x for v in (row if isinstance(row, (list, tuple)) else [row])]
    return (ctypes.c_int64 * len(x))(*[int(v) for v in x]), len(x)


def msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                 out=None):
    """value [bs,keys,heads,dim] (f32/bf16), sampling_locations [bs,Q,heads,L,P,2],
    attention_weights [bs,Q,heads,L,P] -> [bs,Q,heads*dim] f32."""
    _lib.require_gpu(value, sampling_locations, attention_weights, what="ms_deform_attn_forward")
    bs, keys, heads, dim = value.shape
    _, Q, h2, L, P, two = sampling_locations.shape
    if h2 != heads or two != 2 or tuple(attention_weights.shape) != (bs, Q, heads, L, P):
        raise RuntimeError("ms_deform_attn_forward: inconsistent shapes")
    shapes, n = _host_i64(spatial_shapes)
    starts, m = _host_i64(level_start_index)
    if n != 2 * L or m != L:
        raise RuntimeError("ms_deform_attn_forward: spatial_shapes must be [L,2], level_start_index [L]")
    if sampling_locations.dtype != torch.float32 or attention_weights.dtype != torch.float32:
        raise RuntimeError("ms_deform_attn_forward: locations / weights must be float32")
    if out is None:
        out = torch.empty(bs, Q, heads * dim, device=value.device, dtype=torch.float32)
    ev = _lib.timer.record("msda_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_msda_fwd(_lib.ptr(value), shapes, starts, _lib.ptr(sampling_locations),
                                 _lib.ptr(attention_weights), _lib.ptr(out), bs, keys, heads, dim, Q, L, P,
                                 _lib.dtype_code(value), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_msda_fwd")
    return out


class MultiScaleDeformableAttnFunction_fp32(torch.autograd.Function):
    """apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
    attention_weights, im2col_step) -> [bs, num_queries, embed_dims]; inputs are cast to float32
    as the reference's ``custom_fwd(cast_inputs=torch.float32)`` does (:93)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        bs = value.shape[0]
        step = min(bs, int(im2col_step))
        if step > 0 and bs % step != 0:
            raise RuntimeError(f"batch({bs}) must divide im2col_step({step})")  # mmcv's check
        value, loc, attn = (value.float().contiguous(), sampling_locations.float().contiguous(),
                            attention_weights.float().contiguous())
        ctx.levels = (_host_i64(value_spatial_shapes), _host_i64(value_level_start_index))
        ctx.save_for_backward(value, loc, attn)
        return msda_forward(value, value_spatial_shapes, value_level_start_index, loc, attn)

    @staticmethod
    def backward(ctx, grad_output):
        """-> (grad_value, None, None, grad_sampling_loc, grad_attn_weight, None), as
        multi_scale_deformable_attn_function.py:130-162."""
        value, loc, attn = ctx.saved_tensors
        (shapes, _), (starts, _) = ctx.levels
        bs, keys, heads, dim = value.shape
        _, Q, _, L, P, _ = loc.shape
        grad_output = grad_output.contiguous().float()
        grad_value = torch.zeros_like(value)
        grad_loc, grad_attn = torch.empty_like(loc), torch.empty_like(attn)
        rc = _lib.lib().rac_msda_bwd(_lib.ptr(grad_output), _lib.ptr(value), shapes, starts, _lib.ptr(loc), _lib.ptr(attn),
                                     _lib.ptr(grad_value), _lib.ptr(grad_loc), _lib.ptr(grad_attn), bs, keys, heads, dim,
                                     Q, L, P, _lib.stream_ptr())
        _lib.check(rc, "rac_msda_bwd")
        return grad_value, None, None, grad_loc, grad_attn, None


class MultiScaleDeformableAttnFunction_fp16(MultiScaleDeformableAttnFunction_fp32):
    """The reference routes fp16 values to the fp32 function as well
    (models/bev_self_attention.py:195-198); kept for name compatibility."""
