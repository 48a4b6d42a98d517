"""Python launchers of the two fully fused per-layer kernels (C-ABI: rac_sampling4d_fwd,
rac_bev_sampling_fwd).  The Linear outputs they consume may be column slices of one wide GEMM
output: only the last dimension has to be contiguous, the row stride is passed through."""
import ctypes
import functools

import torch

from . import _lib


@functools.lru_cache(maxsize=64)
def _depth_base(d_region, depth_num):
    """torch.linspace(-d_region, d_region, depth_num) as the reference evaluates it
    (racformer_transformer.py:395,515), computed once per (layer, depth_num) on the host."""
    vals = torch.linspace(-d_region, d_region, depth_num).tolist()
    return (ctypes.c_float * depth_num)(*vals)


def _rows(t, width, what):
    """[B,Q,width] view with unit last stride -> (pointer, row stride in floats)."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"racformer_amd.{what}: expected a float32 CUDA tensor")
    if t.shape[-1] != width or t.stride(-1) != 1 or (t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
        raise RuntimeError(f"racformer_amd.{what}: expected [B,Q,{width}] rows with unit inner stride")
    return _lib.ptr(t), int(t.stride(-2))


def box_prep(query_bbox, pc_range):
    """[B,Q,10] polar boxes -> [B,Q,8] (cx,cy,cz,w,l,h,cos yaw,sin yaw), once per decoder layer."""
    _lib.require_gpu(query_bbox, what="box_prep")
    table = torch.empty(query_bbox.shape[:-1] + (8,), device=query_bbox.device, dtype=torch.float32)
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    rc = _lib.lib().rac_box_prep_fwd(_lib.ptr(query_bbox), _lib.ptr(table), query_bbox.numel() // 10, pc,
                                     _lib.stream_ptr())
    _lib.check(rc, "rac_box_prep_fwd")
    return table


_checked_views = set()


def sampling4d_fused(mlvl_feats, query_bbox, offsets, ray_logits, scale_logits, time_diff, lidar2img,
                     num_frames, num_groups, num_points, depth_num, pc_range, d_region, image_h, image_w,
                     eps=1e-5, debug=False, box_table=None, view_in=None, compact=None):
    """-> [B,Q,G,T*P,C] (and, with debug=True, the kernel's own locations [S,Q,P,3] and softmaxed scale
    weights [S,Q,P,L] for parity checks).  ``view_in`` (u8 [S,Q,P], parity tests only): camera index per point that
    replaces the kernel's own first-valid-view selection.  ``compact``: True / False selects the kernel variant that sets points
    without any tap aside (rigs that do not cover the full circle); None: by the number of cameras."""
    feats = list(mlvl_feats)
    L = len(feats)
    _lib.require_gpu(*feats, query_bbox, time_diff, lidar2img, what="sampling4d_fused")
    B, Q, _ = query_bbox.shape
    T, G, NP, D = num_frames, num_groups, num_points, depth_num
    P = NP * D
    S, N, _, _, C = feats[0].shape
    if S != B * T * G or lidar2img.shape[1] != T * N:
        raise RuntimeError("sampling4d_fused: feature slots / lidar2img do not match B*T*G / T*N")
    p_off, ld_off = _rows(offsets, G * P * 3, "sampling4d_fused(offsets)")
    p_ray, ld_ray = _rows(ray_logits, D, "sampling4d_fused(ray_logits)")
    p_sc, ld_sc = _rows(scale_logits, G * T * P * L, "sampling4d_fused(scale_logits)")
    if box_table is None:
        box_table = box_prep(query_bbox, pc_range)
    out = torch.empty(B, Q, G, T * P, C, device=query_bbox.device, dtype=torch.float32)
    loc_out = w_out = None
    capture = _lib.timer is not None and getattr(_lib.timer, "capture_inputs", False)
    want_debug, debug = debug, debug or capture
    if debug:
        loc_out = torch.empty(S, Q, P, 3, device=out.device, dtype=torch.float32)
        w_out = torch.empty(S, Q, P, L, device=out.device, dtype=torch.float32)
    if view_in is not None:
        if view_in.dtype != torch.uint8 or tuple(view_in.shape) != (S, Q, P) or not view_in.is_cuda or not view_in.is_contiguous():
            raise RuntimeError(f"sampling4d_fused: view_in must be a contiguous CUDA uint8 [{S},{Q},{P}] tensor")
        key = (view_in.data_ptr(), view_in._version, N)
        if key not in _checked_views:        # (a host read: once per tensor, so that a captured plan's forwards issue none)
            if int(view_in.max()) >= N:
                raise RuntimeError("sampling4d_fused: view_in holds a camera index >= N")
            if len(_checked_views) > 256:
                _checked_views.clear()
            _checked_views.add(key)
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hw = (ctypes.c_int32 * (2 * L))(*[int(x) for f in feats for x in f.shape[2:4]])
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    ev = _lib.timer.record("sampling4d_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_sampling4d_fwd(
        ptrs, hw, L, _lib.ptr(query_bbox), _lib.ptr(box_table), p_off, p_ray, p_sc, _lib.ptr(time_diff), _lib.ptr(lidar2img),
        _lib.ptr(out), _lib.ptr(loc_out) if debug else None, _lib.ptr(w_out) if debug else None,
        _lib.ptr(view_in) if view_in is not None else None, ld_off, ld_ray, ld_sc, B, T, N, G, Q, NP, D, C, pc, _depth_base(float(d_region), D), float(d_region),
        float(image_h), float(image_w), float(eps), _lib.dtype_code(feats[0]), -1 if compact is None else int(bool(compact)), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_sampling4d_fwd")
    if capture:  # bench.py: the locations this launch sampled at, for the algorithmic-byte count
        _lib.timer.captured.append((loc_out, [tuple(f.shape) for f in feats]))
    return (out, loc_out, w_out) if want_debug else out


def bev_sampling_fused(value, hw, query_bbox, offsets, ray_logits, scale_logits, queue_logits, time_diff,
                       num_frames, num_heads, num_points, depth_num, pc_range, d_region, debug=False, box_table=None,
                       out=None):
    """value [B*T, H*W, heads, 64] -> [B,Q,heads*64] (frame-fused, before output_proj)."""
    _lib.require_gpu(value, query_bbox, time_diff, what="bev_sampling_fused")
    B, Q, _ = query_bbox.shape
    T, Hn, NP, D = num_frames, num_heads, num_points, depth_num
    P = NP * D
    H, W = hw
    if tuple(value.shape) != (B * T, H * W, Hn, 64):
        raise RuntimeError(f"bev_sampling_fused: value must be [{B * T},{H * W},{Hn},64], got {tuple(value.shape)}")
    p_off, ld_off = _rows(offsets, Hn * P * 2, "bev_sampling_fused(offsets)")
    p_ray, ld_ray = _rows(ray_logits, D, "bev_sampling_fused(ray_logits)")
    p_sc, ld_sc = _rows(scale_logits, Hn * P, "bev_sampling_fused(scale_logits)")
    p_qu, ld_qu = _rows(queue_logits, T, "bev_sampling_fused(queue_logits)")
    if box_table is None:
        box_table = box_prep(query_bbox, pc_range)
    if out is None:
        out = torch.empty(B, Q, Hn * 64, device=query_bbox.device, dtype=torch.float32)
    elif not out.is_contiguous() or tuple(out.shape) != (B, Q, Hn * 64):
        raise RuntimeError("bev_sampling_fused: out must be a contiguous [B,Q,heads*64] tensor")
    loc_out = torch.empty(B, Q, Hn, T, P, 2, device=out.device, dtype=torch.float32) if debug else None
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    ev = _lib.timer.record("bev_sampling_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_bev_sampling_fwd(
        _lib.ptr(value), _lib.ptr(query_bbox), _lib.ptr(box_table), p_off, p_ray, p_sc, p_qu, _lib.ptr(time_diff), _lib.ptr(out),
        _lib.ptr(loc_out) if debug else None, ld_off, ld_ray, ld_sc, ld_qu, B, T, Q, Hn, NP, D, H, W, 64, pc,
        _depth_base(float(d_region), D), float(d_region), _lib.dtype_code(value), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_bev_sampling_fwd")
    return (out, loc_out) if debug else out


def quantize_values_i16(value):
    """A hoisted value stream [B*T, H*W, heads, 64] f32 -> (int16 mantissas of the same shape, scales [B*T, H*W, heads] f32):
    one power-of-two scale per (pixel, head) block of 64 channels, value = q * scale (rac_quant_i16_fwd; opt-in storage of
    RaCFormerTransformerDecoderLayer.value_storage = "i16")."""
    _lib.require_gpu(value, what="quantize_values_i16")
    if value.dtype != torch.float32 or value.shape[-1] != 64:
        raise RuntimeError("quantize_values_i16: float32 [..., 64] value stream expected")
    q = torch.empty(value.shape, device=value.device, dtype=torch.int16)
    scale = torch.empty(value.shape[:-1], device=value.device, dtype=torch.float32)
    ev = _lib.timer.record("quant_i16_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_quant_i16_fwd(_lib.ptr(value), _lib.ptr(q), _lib.ptr(scale), scale.numel(), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_quant_i16_fwd")
    return q, scale


def bev_sampling_multi_fused(streams, hw, query_bbox, time_diff, num_frames, num_heads, num_points, depth_num, pc_range,
                             d_region, box_table, out, value_scales=None):
    """The BEV streams of one decoder layer in one launch (rac_bev_sampling_multi_fwd).  ``streams``: list of
    (value [B*T,H*W,heads,64], offsets, ray_logits, scale_logits, queue_logits) with equal row strides; ``out`` [n,B,Q,heads*64].
    ``value_scales``: per stream the [B*T,H*W,heads] scale table of an int16 block-stored value stream (quantize_values_i16;
    rac_bev_sampling_multi_q16_fwd)."""
    B, Q, _ = query_bbox.shape
    T, Hn, NP, D = num_frames, num_heads, num_points, depth_num
    P = NP * D
    H, W = hw
    n = len(streams)
    if tuple(out.shape) != (n, B, Q, Hn * 64) or not out.is_contiguous():
        raise RuntimeError("bev_sampling_multi_fused: out must be a contiguous [streams,B,Q,heads*64] tensor")
    lds = None
    cols = [[], [], [], [], []]
    for value, off, ray, sc, qu in streams:
        _lib.require_gpu(value, what="bev_sampling_multi_fused")
        if tuple(value.shape) != (B * T, H * W, Hn, 64):
            raise RuntimeError(f"bev_sampling_multi_fused: value must be [{B * T},{H * W},{Hn},64], got {tuple(value.shape)}")
        p_off, ld_off = _rows(off, Hn * P * 2, "bev_sampling_multi_fused(offsets)")
        p_ray, ld_ray = _rows(ray, D, "bev_sampling_multi_fused(ray_logits)")
        p_sc, ld_sc = _rows(sc, Hn * P, "bev_sampling_multi_fused(scale_logits)")
        p_qu, ld_qu = _rows(qu, T, "bev_sampling_multi_fused(queue_logits)")
        if lds is None:
            lds = (ld_off, ld_ray, ld_sc, ld_qu)
        elif lds != (ld_off, ld_ray, ld_sc, ld_qu) or value.dtype != streams[0][0].dtype:
            raise RuntimeError("bev_sampling_multi_fused: the streams must share row strides and dtype")
        for c, v in zip(cols, (value.data_ptr(), p_off.value, p_ray.value, p_sc.value, p_qu.value)):
            c.append(v)
    arr = [(ctypes.c_void_p * n)(*c) for c in cols]
    outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    ev = _lib.timer.record(f"bev_sampling_x{n}_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    if value_scales is not None:
        if len(value_scales) != n or any(v[0].dtype != torch.int16 for v in streams):
            raise RuntimeError("bev_sampling_multi_fused: int16 value streams and one scale table per stream expected")
        for sc_ in value_scales:
            _lib.require_gpu(sc_, what="bev_sampling_multi_fused(value_scales)")
            if tuple(sc_.shape) != (B * T, H * W, Hn) or sc_.dtype != torch.float32:
                raise RuntimeError(f"bev_sampling_multi_fused: scale table must be f32 [{B * T},{H * W},{Hn}]")
        vsc = (ctypes.c_void_p * n)(*[sc_.data_ptr() for sc_ in value_scales])
        rc = _lib.lib().rac_bev_sampling_multi_q16_fwd(
            n, arr[0], vsc, arr[1], arr[2], arr[3], arr[4], outs, _lib.ptr(query_bbox), _lib.ptr(box_table), _lib.ptr(time_diff),
            lds[0], lds[1], lds[2], lds[3], B, T, Q, Hn, NP, D, H, W, 64, pc, _depth_base(float(d_region), D), float(d_region),
            _lib.stream_ptr())
    else:
        rc = _lib.lib().rac_bev_sampling_multi_fwd(
            n, arr[0], arr[1], arr[2], arr[3], arr[4], outs, _lib.ptr(query_bbox), _lib.ptr(box_table), _lib.ptr(time_diff),
            lds[0], lds[1], lds[2], lds[3], B, T, Q, Hn, NP, D, H, W, 64, pc, _depth_base(float(d_region), D), float(d_region),
            _lib.dtype_code(streams[0][0]), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_bev_sampling_multi_fwd")
    return out


def sasa_fused(qkv, tau, query_bbox, num_heads, pc_range, box_table=None):
    """qkv [B,Q,3*E] (q|k|v, each [heads, E/heads]; may be a column slice), tau [B,Q,heads] ->
    attention output [B,Q,E] before out_proj."""
    _lib.require_gpu(query_bbox, what="sasa_fused")
    B, Q, _ = query_bbox.shape
    E = qkv.shape[-1] // 3
    p_qkv, ld_qkv = _rows(qkv, 3 * E, "sasa_fused(qkv)")
    p_tau, ld_tau = _rows(tau, num_heads, "sasa_fused(tau)")
    out = torch.empty(B, Q, E, device=qkv.device, dtype=torch.float32)
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    ev = _lib.timer.record("sasa_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_sasa_fwd(p_qkv, p_tau, _lib.ptr(query_bbox), _lib.ptr(box_table) if box_table is not None else None,
                                 _lib.ptr(out), ld_qkv, ld_tau, B, Q, num_heads,
                                 E // num_heads, pc, _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_sasa_fwd")
    return out


def mixing_fused(x, params, in_points, n_groups, out_points=128, eps=1e-5, split=False, param_scale=1.0, f16x3=False, out=None):
    """x [B,Q,G,P,64] (contiguous), params [B,Q,G*(64*64+128*P)] (unit inner stride) ->
    relu(LN(S @ relu(LN(x @ M)))) as [B,Q,G*128*64], ready for out_proj.
    ``split=True``: instead returns the f16 line image [B*Q, G*256, hi 32 | lo 32] of the same values * SPLIT_ACT_SCALE
    (A operand of rac_outproj_fwd: every value stored once as hi + lo).
    ``param_scale``: factor applied to every parameter on load (the power-of-two alpha of a split generator GEMM).
    ``f16x3``: run the two products as 3-product split-precision f16 MFMAs (RAC_MIX_F16X3) instead of f32-input MFMAs.
    ``out``: write into this tensor (a row range of a larger image) instead of allocating."""
    _lib.require_gpu(x, what="mixing_fused")
    B, Q, G, P, C = x.shape
    if G != n_groups or P != in_points or x.dtype != torch.float32:
        raise RuntimeError("mixing_fused: x must be float32 [B,Q,G,P,64]")
    width = G * (C * C + out_points * P)
    p_par, ld_par = _rows(params, width, "mixing_fused(params)")
    if out is not None:
        want = ((B * Q, G * out_points * C // 32, 64), torch.float16) if split else ((B, Q, G * out_points * C), torch.float32)
        if tuple(out.shape) != want[0] or out.dtype != want[1] or not out.is_contiguous() or not out.is_cuda:
            raise RuntimeError(f"mixing_fused: out must be a contiguous CUDA {want[1]} tensor of shape {want[0]}")
    elif split:
        out = torch.empty(B * Q, G * out_points * C // 32, 64, device=x.device, dtype=torch.float16)
    else:
        out = torch.empty(B, Q, G * out_points * C, device=x.device, dtype=torch.float32)
    ev = _lib.timer.record("mixing_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_mixing_fwd(_lib.ptr(x), p_par, float(param_scale), None if split else _lib.ptr(out),
                                   _lib.ptr(out) if split else None,
                                   SPLIT_ACT_SCALE, ld_par, B * Q, G, P, C, out_points, float(eps),
                                   _lib.MIX_F16X3 if f16x3 else _lib.MIX_F32, _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_mixing_fwd")
    return out


def mixing_sampled_fused(mlvl_feats, query_bbox, offsets, ray_logits, scale_logits, time_diff, lidar2img, num_frames, num_groups,
                         num_points, depth_num, pc_range, d_region, image_h, image_w, params, out_points=128, eps_proj=1e-5, eps=1e-5,
                         debug=False, box_table=None, view_in=None, param_scale=1.0):
    """sampling4d_fused + mixing_fused(split=True, f16x3=True) as ONE launch (rac_mixing_sampled_fwd): the mixing workgroup of an
    item gathers its own sampled features (bit for bit what sampling4d_fused returns) while its parameters stream in; the
    [B,Q,G,T*P,C] tensor never exists.  -> the f16 line image [B*Q, G*256, 64] for outproj_fused (with debug=True also the kernel's
    locations [S,Q,P,3] and level weights [S,Q,P,L]).  Same hooks as sampling4d_fused (view_in, the timer's capture_inputs)."""
    feats = list(mlvl_feats)
    L = len(feats)
    _lib.require_gpu(*feats, query_bbox, time_diff, lidar2img, params, what="mixing_sampled_fused")
    B, Q, _ = query_bbox.shape
    T, G, NP, D = num_frames, num_groups, num_points, depth_num
    P = NP * D
    S, N, _, _, C = feats[0].shape
    if S != B * T * G or lidar2img.shape[1] != T * N:
        raise RuntimeError("mixing_sampled_fused: feature slots / lidar2img do not match B*T*G / T*N")
    if not mixing_sampled_supported(feats, T, P):
        raise RuntimeError("mixing_sampled_fused: built for 4 contiguous fp32 levels of 64 channels and T*P <= 96 points per item")
    p_off, ld_off = _rows(offsets, G * P * 3, "mixing_sampled_fused(offsets)")
    p_ray, ld_ray = _rows(ray_logits, D, "mixing_sampled_fused(ray_logits)")
    p_sc, ld_sc = _rows(scale_logits, G * T * P * L, "mixing_sampled_fused(scale_logits)")
    p_par, ld_par = _rows(params, G * (C * C + out_points * T * P), "mixing_sampled_fused(params)")
    if box_table is None:
        box_table = box_prep(query_bbox, pc_range)
    out = torch.empty(B * Q, G * out_points * C // 32, 64, device=query_bbox.device, dtype=torch.float16)
    loc_out = w_out = None
    capture = _lib.timer is not None and getattr(_lib.timer, "capture_inputs", False)
    want_debug, debug = debug, debug or capture
    if debug:
        loc_out = torch.empty(S, Q, P, 3, device=out.device, dtype=torch.float32)
        w_out = torch.empty(S, Q, P, L, device=out.device, dtype=torch.float32)
    if view_in is not None:
        if view_in.dtype != torch.uint8 or tuple(view_in.shape) != (S, Q, P) or not view_in.is_cuda or not view_in.is_contiguous():
            raise RuntimeError(f"mixing_sampled_fused: view_in must be a contiguous CUDA uint8 [{S},{Q},{P}] tensor")
        key = (view_in.data_ptr(), view_in._version, N)
        if key not in _checked_views:        # (a host read: once per tensor, so that a captured plan's forwards issue none)
            if int(view_in.max()) >= N:
                raise RuntimeError("mixing_sampled_fused: view_in holds a camera index >= N")
            if len(_checked_views) > 256:
                _checked_views.clear()
            _checked_views.add(key)
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hw = (ctypes.c_int32 * (2 * L))(*[int(x) for f in feats for x in f.shape[2:4]])
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    ev = _lib.timer.record("mixing_sampled_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_mixing_sampled_fwd(
        ptrs, hw, L, _lib.ptr(query_bbox), _lib.ptr(box_table), p_off, p_ray, p_sc, _lib.ptr(time_diff), _lib.ptr(lidar2img),
        _lib.ptr(loc_out) if debug else None, _lib.ptr(w_out) if debug else None, _lib.ptr(view_in) if view_in is not None else None,
        ld_off, ld_ray, ld_sc, B, T, N, G, Q, NP, D, C, pc, _depth_base(float(d_region), D), float(d_region), float(image_h),
        float(image_w), float(eps_proj), _lib.dtype_code(feats[0]), p_par, float(param_scale), None, _lib.ptr(out), SPLIT_ACT_SCALE,
        ld_par, out_points, float(eps), _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_mixing_sampled_fwd")
    if capture:  # bench.py: the locations this launch sampled at, for the algorithmic-byte count
        _lib.timer.captured.append((loc_out, [tuple(f.shape) for f in feats]))
    return (out, loc_out, w_out) if want_debug else out


def mixing_sampled_supported(feats, num_frames, points_per_frame):
    """Shapes rac_mixing_sampled_fwd is built for: 4 contiguous fp32 CUDA levels [S,N,H,W,64], at most 96 points per item, and every
    level's slots of one sample below 2 GiB."""
    feats = list(feats)
    if len(feats) != 4 or num_frames * points_per_frame > 96:
        return False
    for f in feats:
        if not (f.is_cuda and f.dtype == torch.float32 and f.dim() == 5 and f.shape[-1] == 64 and f.is_contiguous()):
            return False
    return True


def refine_fused(proposal, delta, time_diff_safe, num_ray):
    """refine_bbox + velocity / time_diff + theta_d2xy in one launch.
    -> (bbox_pred [B,Q,10] polar, bbox_xy [B,Q,10] normalised xy)."""
    proposal, delta = proposal.contiguous(), delta.contiguous()
    _lib.require_gpu(proposal, delta, time_diff_safe, what="refine_fused")
    B, Q, _ = proposal.shape
    pred, xy = torch.empty_like(proposal), torch.empty_like(proposal)
    rc = _lib.lib().rac_refine_fwd(_lib.ptr(proposal), _lib.ptr(delta), _lib.ptr(time_diff_safe), _lib.ptr(pred),
                                   _lib.ptr(xy), B, Q, time_diff_safe.shape[1], float(num_ray), _lib.stream_ptr())
    _lib.check(rc, "rac_refine_fwd")
    return pred, xy


SPLIT_ACT_SCALE = 16.0   # power of two applied to activations before the f16 hi/lo split (keeps lo out of f16 subnormals)
SPLIT_BIAS_PAD = 64      # extra K columns of a split image that carry the bias ([1, 1, 0...] against [b_hi, b_lo, 0...]);
                         # 3*256 + 64 = 832 = 13 x 64 keeps hipBLASLt on its fast kernels (776: 187 us, 832: 134 us)
SPLIT_SLICE = 2048       # K slice of out_proj's split-K (16 slices of the 32768-long reduction)


def head_finish_fused(cls_scores, bbox_xy, pc_range):
    """(nan_to_num(cls_scores) in place, boxes [..,10] = denormalised + reordered nan_to_num(bbox_xy)): the element-wise tail of
    RaCFormerTransformer.forward / RaCFormer_head.forward in one launch (rac_head_finish_fwd)."""
    _lib.require_gpu(cls_scores, bbox_xy, what="head_finish_fused")
    if cls_scores.dtype != torch.float32 or bbox_xy.dtype != torch.float32 or bbox_xy.shape[-1] != 10:
        raise RuntimeError("head_finish_fused: float32 tensors, boxes with 10 columns")
    box = torch.empty_like(bbox_xy)
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    rc = _lib.lib().rac_head_finish_fwd(_lib.ptr(cls_scores), cls_scores.numel(), _lib.ptr(bbox_xy), _lib.ptr(box), bbox_xy.numel() // 10, 10,
                                        pc, _lib.stream_ptr())
    _lib.check(rc, "rac_head_finish_fwd")
    return cls_scores, box


def layer_boundary_fused(proposal, delta, time_diff_safe, num_ray, pc_range, pe_linear, pe_norm, xy_out=None):
    """refine_fused + box_prep + pe_head for the refined boxes in one launch (rac_layer_boundary_fwd).
    -> (bbox_pred [B,Q,10], bbox_xy [B,Q,10], box_table [B,Q,8], pe_head output [B,Q,256])."""
    proposal, delta = proposal.contiguous(), delta.contiguous()
    _lib.require_gpu(proposal, delta, time_diff_safe, what="layer_boundary_fused")
    B, Q, _ = proposal.shape
    pred = torch.empty_like(proposal)
    if xy_out is None:
        xy = torch.empty_like(proposal)
    else:
        if tuple(xy_out.shape) != tuple(proposal.shape) or not xy_out.is_contiguous() or xy_out.dtype != torch.float32:
            raise RuntimeError("layer_boundary_fused: xy_out must be a contiguous float32 tensor shaped like the boxes")
        xy = xy_out
    table = torch.empty(B, Q, 8, device=proposal.device, dtype=torch.float32)
    h = torch.empty(B, Q, pe_linear.weight.shape[0], device=proposal.device, dtype=torch.float32)
    pc = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    rc = _lib.lib().rac_layer_boundary_fwd(_lib.ptr(proposal), _lib.ptr(delta), _lib.ptr(time_diff_safe), _lib.ptr(pred),
                                           _lib.ptr(xy), _lib.ptr(table), pc, _lib.ptr(pe_linear.weight), _lib.ptr(pe_linear.bias),
                                           _lib.ptr(pe_norm.weight), _lib.ptr(pe_norm.bias), _lib.ptr(h), B, Q,
                                           time_diff_safe.shape[1], pe_linear.weight.shape[0], float(num_ray), float(pe_norm.eps),
                                           _lib.stream_ptr())
    _lib.check(rc, "rac_layer_boundary_fwd")
    return pred, xy, table, h


def add_ln(a, norm, residual=None, bias=None, relu=False, num_partials=1, post=None, out=None, split=False, a_scale=1.0,
           split_lines=False, split_out=None):
    """[relu](LayerNorm(a_scale * sum_s a[s] + residual + bias)) [+ post] with ``norm`` an nn.LayerNorm; a is [..., dim]
    (unit inner stride; rows may be a column slice of a wider tensor) or [S, ..., dim] with num_partials=S.
    ``out``: optional destination (may itself be a column slice).  One launch.
    ``split=True``: also returns the f16 [rows, 3*dim + SPLIT_BIAS_PAD] = [hi | hi | lo | 1 1 0..] image of
    ``out * SPLIT_ACT_SCALE`` (A operand of a split GEMM, see ``split_weight_f16``); with ``split_lines`` the image is the
    line image [rows, dim/32 * 64] = [hi 32 | lo 32] per 32 columns that ``generator_fused`` reads."""
    _lib.require_gpu(norm.weight, what="add_ln")
    dim = a.shape[-1]
    if num_partials > 1:
        a = a.contiguous()
        rows, ld_a, shape = a.numel() // dim // num_partials, dim, a.shape[1:]
    else:
        if a.stride(-1) != 1 or not a.is_cuda:
            raise RuntimeError("add_ln: input must be a CUDA tensor with unit inner stride")
        lead = a.shape[:-1]
        rows, shape = int(torch.Size(lead).numel()), a.shape
        ld_a = a.stride(-2) if a.dim() > 1 else dim
        if a.dim() > 2 and a.stride(0) != a.shape[1] * a.stride(1):
            raise RuntimeError("add_ln: rows must be equally strided")
    if residual is not None:
        residual = residual.contiguous()
        if residual.numel() == rows * dim:
            shape = residual.shape
    if post is not None:
        post = post.contiguous()
    if out is None:
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
        ld_out = dim
    else:
        if out.stride(-1) != 1 or out.shape[-1] != dim:
            raise RuntimeError("add_ln: out must have unit inner stride and the normalised width")
        ld_out = out.stride(-2)
    if split:
        width = 2 * dim if split_lines else 3 * dim + SPLIT_BIAS_PAD
        if split_out is None:
            split_out = torch.empty(rows, width, device=a.device, dtype=torch.float16)
        elif split_out.dtype != torch.float16 or not split_out.is_contiguous() or tuple(split_out.shape) != (rows, width):
            raise RuntimeError(f"add_ln: split_out must be a contiguous f16 [{rows}, {width}] tensor")
    rc = _lib.lib().rac_add_ln_fwd(_lib.ptr(a), num_partials, rows * dim, ld_a, float(a_scale),
                                   _lib.ptr(residual) if residual is not None else None,
                                   _lib.ptr(bias) if bias is not None else None, _lib.ptr(norm.weight), _lib.ptr(norm.bias),
                                   _lib.ptr(post) if post is not None else None, _lib.ptr(out), ld_out, rows, dim,
                                   float(norm.eps), int(relu), _lib.ptr(split_out) if split else None,
                                   SPLIT_ACT_SCALE, 0 if split_lines else SPLIT_BIAS_PAD, 1 if split_lines else 0, _lib.stream_ptr())
    _lib.check(rc, "rac_add_ln_fwd")
    return (out, split_out) if split else out


def split_weight_f16(weight, bias=None):
    """nn.Linear weight [N,K] fp32 -> (f16 [N,3K] = [hi | lo | hi] of weight * 2^s, alpha) such that
        x @ weight.T  ==  alpha * ([x_hi | x_hi | x_lo] @ [w_hi | w_lo | w_hi].T)        (x * SPLIT_ACT_SCALE = x_hi + x_lo)
    up to the dropped lo*lo term (2^-22 relative): fp32-GEMM accuracy from three f16 MFMA products accumulated in
    fp32.  2^s brings max|w| to [2^13, 2^14) so that the lo parts stay clear of f16 subnormals; alpha undoes both
    power-of-two scalings exactly.  With ``bias`` the image is [N, 3K + SPLIT_BIAS_PAD] = [.. | b_hi | b_lo | 0..]
    (bias * 2^s): against the [.. | 1 | 1 | 0..] columns of add_ln's activation image the GEMM adds the bias
    itself.  Done once per set of weights (the caller caches it).  (None, None) if f16 cannot hold the operands."""
    import math
    w = weight.detach().float()
    amax = float(w.abs().max())
    if not (amax > 0.0) or amax != amax or amax == float("inf"):
        return None, None
    s = 13 - math.frexp(amax)[1] + 1          # amax * 2^s in [2^13, 2^14)
    ws = w * (2.0 ** s)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    parts = [hi, lo, hi]
    if bias is not None:
        bs = bias.detach().float() * (2.0 ** s)
        if not float(bs.abs().max()) < 6.0e4:
            return None, None
        bh = bs.to(torch.float16)
        bl = (bs - bh.float()).to(torch.float16)
        parts += [bh[:, None], bl[:, None], hi.new_zeros(hi.shape[0], SPLIT_BIAS_PAD - 2)]
    return torch.cat(parts, dim=1).contiguous(), 2.0 ** (-s) / SPLIT_ACT_SCALE


def pe_head(x3, linear, norm):
    """relu(LayerNorm(linear(x3))) for the 3-wide position-encoder input, one launch.  x3: [..., 3] view."""
    if x3.stride(-1) != 1 or not x3.is_cuda:
        raise RuntimeError("pe_head: input must be a CUDA tensor with unit inner stride")
    rows = int(torch.Size(x3.shape[:-1]).numel())
    out = torch.empty(x3.shape[:-1] + (linear.weight.shape[0],), device=x3.device, dtype=torch.float32)
    rc = _lib.lib().rac_pe_head_fwd(_lib.ptr(x3), x3.stride(-2), _lib.ptr(linear.weight), _lib.ptr(linear.bias),
                                    _lib.ptr(norm.weight), _lib.ptr(norm.bias), _lib.ptr(out), rows, linear.weight.shape[0],
                                    float(norm.eps), _lib.stream_ptr())
    _lib.check(rc, "rac_pe_head_fwd")
    return out


# ------------------------------------------------------------------------------------------- 3x3 convolution
def pack_conv3x3_weight(weight, cout=256):
    """nn.Conv2d weight [cout, Cin, 3, 3] fp32 -> (ws f16 [9, Cin/32, cout, 2, 32], w_alpha) for rac_conv3x3_fwd (cout 256) /
    rac_conv3x3s2_fwd (cout 64):
    hi / lo of weight * 2^s per (tap, 32-channel chunk, output channel), w_alpha = 2^-s.  (None, None) if the
    weights cannot be held."""
    import math
    w = weight.detach().float()
    co, ci, kh, kw = w.shape
    amax = float(w.abs().max())
    if (kh, kw) != (3, 3) or co != cout or ci % 32 != 0 or not (amax > 0.0) or amax != amax or amax == float("inf"):
        return None, None
    s = 13 - math.frexp(amax)[1] + 1
    ws = (w * (2.0 ** s)).permute(2, 3, 1, 0).reshape(9, ci // 32, 32, co).permute(0, 1, 3, 2)      # [tap, chunk, co, 32]
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    return torch.stack([hi, lo], dim=3).contiguous(), 2.0 ** (-s)


_conv_images = {}
_scratch_ns = [None]


class scratch_namespace:
    """Reusable device scratch (the convolution kernels' activation images) is shared by every forward of a process -- fine while
    forwards follow each other on one stream.  A captured plan that is replayed BESIDE another one (racformer_amd/graph.py, several
    samples in flight on streams of their own) has to own its scratch: forwards run inside ``with scratch_namespace(key)`` get
    buffers of that namespace."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        self.prev, _scratch_ns[0] = _scratch_ns[0], self.key
        return self

    def __exit__(self, *exc):
        _scratch_ns[0] = self.prev


def release_scratch(key):
    """Drops the scratch buffers of a namespace (a captured plan that owned them is gone)."""
    for k in [k for k in _conv_images if k[-1] == key]:
        del _conv_images[k]


class ConvImage:
    """The padded channel-last f16 hi/lo activation image of the convolution kernels, filled in stages:
    ``begin`` (absmax -> the activations' power-of-two scale), ``pack`` (NCHW fp32 source -> a channel range), then
    ``conv`` (3x3 stride 1 -> [N,H,W,256] channel-last) and / or ``conv_s2`` (3x3 stride 2 over the first channels ->
    [N,64,H/2,W/2]).  The buffer (zero border = the convolutions' padding) is allocated once per shape and reused."""

    def __init__(self, N, H, W, cin, device):
        self.N, self.H, self.W, self.cin, self.dev = N, H, W, cin, device
        key = (N, H, W, cin, str(device), _scratch_ns[0])
        xs = _conv_images.get(key)
        if xs is None:
            xs = _conv_images[key] = torch.zeros(N, H + 2, W + 2, cin // 32, 2, 32, device=device, dtype=torch.float16)
        self.xs = xs
        self.amax = torch.empty(1, device=device, dtype=torch.float32)

    def begin(self, scan, floor=0.0):
        """amax = max(floor, max |v| over the tensors in ``scan``); every source packed later must be covered by it."""
        n = len(scan)
        _lib.require_gpu(*scan, what="ConvImage.begin")
        ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in scan])
        counts = (ctypes.c_int64 * max(n, 1))(*[t.numel() for t in scan])
        _lib.check(_lib.lib().rac_absmax_fwd(ptrs, counts, n, float(floor), _lib.ptr(self.amax), _lib.stream_ptr()),
                   "rac_absmax_fwd")
        return self

    def pack(self, src, c_offset):
        _lib.require_gpu(src, what="ConvImage.pack")
        if tuple(src.shape[0:1] + src.shape[2:]) != (self.N, self.H, self.W) or src.dtype != torch.float32:
            raise RuntimeError("ConvImage.pack: sources must be float32 [N,C,H,W] matching the image")
        _lib.check(_lib.lib().rac_conv_pack_fwd(_lib.ptr(src), _lib.ptr(self.amax), _lib.ptr(self.xs), self.N, int(src.shape[1]),
                                                self.H, self.W, self.cin, int(c_offset), _lib.stream_ptr()), "rac_conv_pack_fwd")
        return self

    def pack_live(self, src, bias, c_offset, frames_per_group):
        """Channel range [c_offset, c_offset + C) from ``src`` [G * live, C, H, W] (+ per-channel ``bias``) where the image's N frames
        come in G groups of ``frames_per_group`` and only the first ``live`` of a group exist in src; the others are the bias alone
        (rac_conv_pack_bias_fwd)."""
        _lib.require_gpu(src, what="ConvImage.pack_live")
        G = self.N // frames_per_group
        if self.N % frames_per_group != 0 or src.shape[0] % G != 0 or tuple(src.shape[2:]) != (self.H, self.W) or src.dtype != torch.float32:
            raise RuntimeError("ConvImage.pack_live: src must be float32 [groups * live, C, H, W] matching the image")
        _lib.check(_lib.lib().rac_conv_pack_bias_fwd(_lib.ptr(src), _lib.ptr(bias) if bias is not None else None, _lib.ptr(self.amax),
                                                     _lib.ptr(self.xs), self.N, int(src.shape[1]), self.H, self.W, self.cin, int(c_offset),
                                                     int(frames_per_group), int(src.shape[0] // G), _lib.stream_ptr()),
                   "rac_conv_pack_bias_fwd")
        return self

    def conv(self, ws, w_alpha, bias=None, pixel_bias=None, q16=False):
        """-> [N,H,W,256] fp32 channel-last, or with ``q16`` the same result in the int16 block storage of quantize_values_i16
        (q int16 [N, H*W, 4, 64], scale f32 [N, H*W, 4]), quantised in the kernel's epilogue (rac_conv3x3_q16_fwd)."""
        N, H, W = self.N, self.H, self.W
        if pixel_bias is not None and (tuple(pixel_bias.shape) != (H * W, 256) or not pixel_bias.is_contiguous()
                                       or pixel_bias.dtype != torch.float32 or not pixel_bias.is_cuda):
            raise RuntimeError("ConvImage.conv: pixel_bias must be a contiguous float32 CUDA [H*W, 256] tensor")
        if q16:
            out = torch.empty(N, H * W, 4, 64, device=self.dev, dtype=torch.int16)
            scale = torch.empty(N, H * W, 4, device=self.dev, dtype=torch.float32)
        else:
            out = torch.empty(N, H, W, 256, device=self.dev, dtype=torch.float32)
        ev = _lib.timer.record("temporal_fusion_conv") if _lib.timer is not None else None
        if ev:
            ev[0].record()
        head = (_lib.ptr(self.xs), _lib.ptr(ws), _lib.ptr(bias) if bias is not None else None,
                _lib.ptr(pixel_bias) if pixel_bias is not None else None, _lib.ptr(self.amax), float(w_alpha))
        tail = (N, H, W, self.cin, 256, _lib.stream_ptr())
        if q16:
            rc, what = _lib.lib().rac_conv3x3_q16_fwd(*head, _lib.ptr(out), _lib.ptr(scale), *tail), "rac_conv3x3_q16_fwd"
        else:
            rc, what = _lib.lib().rac_conv3x3_fwd(*head, _lib.ptr(out), *tail), "rac_conv3x3_fwd"
        if ev:
            ev[1].record()
        _lib.check(rc, what)
        return (out, scale) if q16 else out

    def conv_temporal(self, ws, w_alpha, pixel_bias_live, pixel_bias_dead, cin_dead, frames_per_group, live_per_group, q16=False):
        """``conv`` for a stack of [groups, frames_per_group] images whose frames past the first ``live_per_group`` of a group carry a
        per-channel CONSTANT in the channels from ``cin_dead`` on: those images multiply only their first ``cin_dead`` channels and add
        ``pixel_bias_dead`` (the constant's contribution through the zero padding, folded in by the caller) instead of
        ``pixel_bias_live`` (rac_conv3x3_temporal_fwd).  Same outputs as ``conv``."""
        N, H, W = self.N, self.H, self.W
        for pb in (pixel_bias_live, pixel_bias_dead):
            if tuple(pb.shape) != (H * W, 256) or not pb.is_contiguous() or pb.dtype != torch.float32 or not pb.is_cuda:
                raise RuntimeError("ConvImage.conv_temporal: the per-pixel maps must be contiguous float32 CUDA [H*W, 256] tensors")
        if q16:
            out = torch.empty(N, H * W, 4, 64, device=self.dev, dtype=torch.int16)
            scale = torch.empty(N, H * W, 4, device=self.dev, dtype=torch.float32)
        else:
            out, scale = torch.empty(N, H, W, 256, device=self.dev, dtype=torch.float32), None
        ev = _lib.timer.record("temporal_fusion_conv") if _lib.timer is not None else None
        if ev:
            ev[0].record()
        rc = _lib.lib().rac_conv3x3_temporal_fwd(_lib.ptr(self.xs), _lib.ptr(ws), _lib.ptr(pixel_bias_live), _lib.ptr(pixel_bias_dead),
                                                 _lib.ptr(self.amax), float(w_alpha), None if q16 else _lib.ptr(out),
                                                 _lib.ptr(out) if q16 else None, _lib.ptr(scale) if q16 else None, N, H, W, self.cin,
                                                 int(cin_dead), int(frames_per_group), int(live_per_group), _lib.stream_ptr())
        if ev:
            ev[1].record()
        _lib.check(rc, "rac_conv3x3_temporal_fwd")
        return (out, scale) if q16 else out

    def conv_s2(self, ws, w_alpha, bias, cin, out=None):
        """3x3 / stride 2 / pad 1 convolution of the image's first ``cin`` channels -> [N, 64, H/2, W/2] fp32 (NCHW), or
        into channels 0..63 of a given contiguous ``out`` [N, Ctot, H/2, W/2]."""
        N, H, W = self.N, self.H, self.W
        if out is None:
            out = torch.empty(N, 64, H // 2, W // 2, device=self.dev, dtype=torch.float32)
        elif not out.is_contiguous() or tuple(out.shape[0:1] + out.shape[2:]) != (N, H // 2, W // 2) or out.shape[1] < 64:
            raise RuntimeError("ConvImage.conv_s2: out must be a contiguous [N, >=64, H/2, W/2] tensor")
        _lib.check(_lib.lib().rac_conv3x3s2_fwd(_lib.ptr(self.xs), _lib.ptr(ws), _lib.ptr(bias) if bias is not None else None,
                                                _lib.ptr(self.amax), float(w_alpha), _lib.ptr(out), int(out.shape[1]), N, H, W,
                                                int(cin), self.cin, 64, _lib.stream_ptr()), "rac_conv3x3s2_fwd")
        return out


def conv3x3_fused(sources, ws, w_alpha, bias, bounds=None, pixel_bias=None):
    """3x3 / stride 1 / pad 1 convolution of the channel concatenation of ``sources`` (NCHW fp32 tensors with
    equal N, H, W) -> [N, H, W, 256] fp32 channel-last: ConvImage.begin -> pack ... -> conv in one call.
    ``bounds[i]``: a known upper bound of |sources[i]| (the source is then not scanned by the absmax pass).
    ``pixel_bias``: [H*W, 256] additive map (per pixel and output channel, shared by the N images) instead of ``bias``."""
    _lib.require_gpu(*sources, ws, what="conv3x3_fused")
    N, _, H, W = sources[0].shape
    cin = sum(int(t.shape[1]) for t in sources)
    bounds = list(bounds) if bounds is not None else [None] * len(sources)
    img = ConvImage(N, H, W, cin, sources[0].device)
    img.begin([t for t, bnd in zip(sources, bounds) if bnd is None], max([0.0] + [float(b) for b in bounds if b is not None]))
    off = 0
    for t in sources:
        img.pack(t, off)
        off += int(t.shape[1])
    return img.conv(ws, w_alpha, bias, pixel_bias)


# ------------------------------------------------------------------------------------------- ConvGRU branch, own kernels (round 5)
def act_image(tag, frames, H, W, channels, device):
    """A zero-bordered activation image f16 [frames, H+2, W+2, channels/32, 2, 32] of the convolution kernels, from the reusable
    scratch of the current namespace (scratch_namespace): allocated zeroed once per (tag, shape); its producers write interior
    pixels only, so the border stays the convolutions' zero padding."""
    key = ("act", tag, frames, H, W, channels, str(device), _scratch_ns[0])
    img = _conv_images.get(key)
    if img is None:
        img = _conv_images[key] = torch.zeros(frames, H + 2, W + 2, channels // 32, 2, 32, device=device, dtype=torch.float16)
    return img


def _cd_scale(amax=None, mul=0.0, add=0.0):
    return _lib.CdScale(_lib.ptr(amax) if amax is not None else None, float(mul), float(add))


def _cd_frames(live=1, stride=1, first=0):
    return _lib.CdFrames(int(live), int(stride), int(first))


def conv_direct(mode, N, H, W, in_img, in_chunks_total, chunks, ws, w_alpha, cout, in_scale, conv_stride=1, in_chunk0=0,
                in_frames=None, bias=None, out_img=None, out_chunks_total=0, out_chunk0=0, out_frames=None, out_scale=None,
                out_f32=None, pixel_map=None, xpart=None, xpart_frames=None, h_prev=None, h_prev_frames=None, h_out=None,
                h_out_frames=None):
    """rac_conv_direct_fwd (include/racformer_hip.h): 3x3 convolution of a small activation image without LDS staging, epilogue
    ``mode`` = _lib.CD_IMAGE (another activation image) / CD_F32 (channel-last fp32 + per-pixel map) / CD_GRU (the ConvGRU update).
    Scales are (amax tensor | None, mul, add) triples, frame maps (live, stride, first) triples."""
    opt = lambda t: _lib.ptr(t) if t is not None else None      # noqa: E731
    fr = lambda f: _cd_frames(*(f or (1, 1, 0)))                # noqa: E731
    d = _lib.ConvDirect()
    d.mode, d.conv_stride, d.N, d.H, d.W = int(mode), int(conv_stride), int(N), int(H), int(W)
    d.in_img, d.in_chunks_total, d.in_chunk0, d.chunks = opt(in_img), int(in_chunks_total), int(in_chunk0), int(chunks)
    d.in_frames, d.in_scale = fr(in_frames), _cd_scale(*in_scale)
    d.ws, d.w_alpha, d.Cout, d.bias = opt(ws), float(w_alpha), int(cout), opt(bias)
    d.out_img, d.out_chunks_total, d.out_chunk0 = opt(out_img), int(out_chunks_total), int(out_chunk0)
    d.out_frames, d.out_scale = fr(out_frames), _cd_scale(*(out_scale or (None, 0.0, 0.0)))
    d.out_f32, d.pixel_map = opt(out_f32), opt(pixel_map)
    d.xpart, d.xpart_frames = opt(xpart), fr(xpart_frames)
    d.h_prev, d.h_prev_frames = opt(h_prev), fr(h_prev_frames)
    d.h_out, d.h_out_frames = opt(h_out), fr(h_out_frames)
    for t in (in_img, ws, bias, out_img, out_f32, pixel_map, xpart, h_prev, h_out):
        if t is not None:
            _lib.require_gpu(t, what="conv_direct")
    _lib.check(_lib.lib().rac_conv_direct_fwd(ctypes.byref(d), _lib.stream_ptr()), "rac_conv_direct_fwd")


def upsample2x_image(src, img, bound):
    """nn.Upsample(x2, bilinear, align_corners=True) of channel-last maps ``src`` f32 [frames, h*w, C] (given as [frames, h, w, C])
    into the activation image ``img`` [frames, 2h+2, 2w+2, C/32, 2, 32] with the scale of ``bound`` (rac_upsample2x_image_fwd)."""
    _lib.require_gpu(src, img, what="upsample2x_image")
    frames, h, w, C = src.shape
    if tuple(img.shape) != (frames, 2 * h + 2, 2 * w + 2, C // 32, 2, 32) or img.dtype != torch.float16 or src.dtype != torch.float32:
        raise RuntimeError("upsample2x_image: src f32 [frames,h,w,C], img f16 [frames,2h+2,2w+2,C/32,2,32] expected")
    _lib.check(_lib.lib().rac_upsample2x_image_fwd(_lib.ptr(src), _lib.ptr(img), frames, h, w, C, float(bound), _lib.stream_ptr()),
               "rac_upsample2x_image_fwd")
    return img


# ------------------------------------------------------------------------------------------- temporal encoder pieces
def gru_gate_fused(gates, h_prev, h_out, bias_map=None, h_out2=None):
    """ConvGRUCell's element-wise update, one launch: gates [B,3C,H,W] contiguous; h_prev / h_out (/ h_out2) [B,C,H,W]
    views whose per-batch blocks are contiguous (e.g. the [:, t] slot of a [B,T,C,H,W] tensor).  ``bias_map`` [3C,H,W]
    is added to the gates first.  Writes h_out (and h_out2)."""
    B, C3, H, W = gates.shape
    C = C3 // 3
    if bias_map is not None and (tuple(bias_map.shape) != (C3, H, W) or not bias_map.is_contiguous() or not bias_map.is_cuda):
        raise RuntimeError("gru_gate_fused: bias_map must be a contiguous CUDA [3C,H,W] tensor")
    for t in (h_prev, h_out) + ((h_out2,) if h_out2 is not None else ()):
        if not t.is_cuda or tuple(t.shape) != (B, C, H, W) or t[0].is_contiguous() is False or t.dtype != torch.float32:
            raise RuntimeError("gru_gate_fused: h_prev / h_out must be float32 CUDA [B,C,H,W] views with contiguous batches")
    _lib.require_gpu(gates, what="gru_gate_fused")
    bs = lambda t: t.stride(0) if B > 1 else C * H * W   # noqa: E731
    rc = _lib.lib().rac_gru_gate_fwd(_lib.ptr(gates), _lib.ptr(h_prev), bs(h_prev), _lib.ptr(h_out), bs(h_out),
                                     _lib.ptr(bias_map) if bias_map is not None else None,
                                     _lib.ptr(h_out2) if h_out2 is not None else None, bs(h_out2) if h_out2 is not None else 0,
                                     B, C, H * W, _lib.stream_ptr())
    _lib.check(rc, "rac_gru_gate_fwd")
    return h_out


def upsample2x_fused(x):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) on a contiguous [N,C,h,w] tensor, one launch."""
    _lib.require_gpu(x, what="upsample2x_fused")
    N, C, h, w = x.shape
    out = torch.empty(N, C, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    rc = _lib.lib().rac_upsample2x_fwd(_lib.ptr(x), _lib.ptr(out), N * C, h, w, _lib.stream_ptr())
    _lib.check(rc, "rac_upsample2x_fwd")
    return out


# ------------------------------------------------------------------------------------------- row GEMMs
def _rows2d(t, what):
    """-> (tensor kept alive, pointer, row stride) of a float32 CUDA [rows, W] / [B, Q, W] view with unit inner stride."""
    if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1:
        raise RuntimeError(f"racformer_amd.rowgemm({what}): expected a float32 CUDA tensor with unit inner stride")
    if t.dim() == 3:
        if t.stride(0) != t.shape[1] * t.stride(1):
            raise RuntimeError(f"racformer_amd.rowgemm({what}): rows must be equally strided")
        ld = t.stride(1)
    elif t.dim() == 2:
        ld = t.stride(0)
    else:
        raise RuntimeError(f"racformer_amd.rowgemm({what}): expected [rows, W] or [B, Q, W]")
    return t, ctypes.c_void_p(t.data_ptr()), int(ld)


def row_seg(a, num_partials=1, a_scale=1.0, bias0=None, residual=None, norm=None, relu=False, post=None, x_out=None,
            split_out=None, split_lines=False):
    """One 256-wide segment of a rowgemm's A operand (see rac_rowgemm_fwd):
    [relu](LN_norm(a_scale * sum_p a[p] + bias0 + residual)) [+ post]; ``a`` is [rows,256] (any row stride) or, with
    num_partials = S > 1, a contiguous [S, rows, 256].  ``x_out`` / ``split_out``: destinations for the finished rows
    (fp32 [rows,256] / f16 [rows, 768 + SPLIT_BIAS_PAD] for a K-concatenated library GEMM, or -- ``split_lines`` -- the f16
    line image [rows, 8, hi 32 | lo 32] that ``generator_fused`` reads)."""
    g = _lib.RowSeg()
    keep = []
    if num_partials > 1:
        if not a.is_contiguous() or a.shape[0] != num_partials or a.shape[-1] != 256:
            raise RuntimeError("row_seg: partials must be a contiguous [S, rows, 256] tensor")
        g.a, g.ld_a, g.partial_stride = ctypes.c_void_p(a.data_ptr()), 256, a.numel() // num_partials
        keep.append(a)
    else:
        if a.shape[-1] != 256:
            raise RuntimeError("row_seg: segments are 256 wide")
        t, g.a, g.ld_a = _rows2d(a, "a")
        g.partial_stride = 0
        keep.append(t)
    g.num_partials, g.a_scale, g.relu = num_partials, float(a_scale), int(relu)
    if bias0 is not None:
        g.bias0 = ctypes.c_void_p(bias0.data_ptr())
        keep.append(bias0)
    if residual is not None:
        t, g.residual, g.ld_res = _rows2d(residual, "residual")
        keep.append(t)
    if norm is not None:
        g.gamma, g.beta, g.eps = ctypes.c_void_p(norm.weight.data_ptr()), ctypes.c_void_p(norm.bias.data_ptr()), float(norm.eps)
    if post is not None:
        t, g.post, g.ld_post = _rows2d(post, "post")
        keep.append(t)
    if x_out is not None:
        t, g.x_out, g.ld_xout = _rows2d(x_out, "x_out")
        keep.append(t)
    if split_out is not None:
        width = 512 if split_lines else 768 + SPLIT_BIAS_PAD
        if split_out.dtype != torch.float16 or not split_out.is_contiguous() or split_out.shape[-1] != width:
            raise RuntimeError(f"row_seg: split_out must be a contiguous f16 [rows, {width}] tensor")
        g.split_out, g.split_scale = ctypes.c_void_p(split_out.data_ptr()), SPLIT_ACT_SCALE
        g.split_pad, g.split_layout = (0, 1) if split_lines else (SPLIT_BIAS_PAD, 0)
        keep.append(split_out)
    g._keep = keep
    return g


def row_gemm(segs, weight, bias, out, relu_from=None):
    """out = [relu on columns >= relu_from](cat(segs) @ weight.T + bias); weight [N, 256*len(segs)] contiguous."""
    d = _lib.RowGemm()
    N, K = weight.shape
    if K != 256 * len(segs) or not weight.is_contiguous() or not weight.is_cuda or weight.dtype != torch.float32:
        raise RuntimeError("row_gemm: weight must be a contiguous float32 CUDA [N, 256 * segments] tensor")
    for i, g in enumerate(segs):
        d.seg[i] = g
    d.num_seg, d.N = len(segs), N
    d.w = ctypes.c_void_p(weight.data_ptr())
    d.b = ctypes.c_void_p(bias.data_ptr()) if bias is not None else None
    t, d.out, d.ld_out = _rows2d(out, "out")
    if out.shape[-1] != N:
        raise RuntimeError("row_gemm: out must be [rows, N]")
    d.relu_from = N if relu_from is None else int(relu_from)
    d._keep = [segs, weight, bias, t]
    return d


def rowgemm_launch(descs, rows):
    """One launch for up to 3 independent row GEMMs over the same rows."""
    arr = (_lib.RowGemm * len(descs))(*descs)
    rc = _lib.lib().rac_rowgemm_fwd(arr, len(descs), int(rows), _lib.stream_ptr())
    _lib.check(rc, "rac_rowgemm_fwd")


# ------------------------------------------------------------------------------------------- split-precision GEMM
def pack_gemm_split_weight(weight):
    """nn.Linear weight [N, K] fp32 (device) -> (f16 line image [N, K/32, 64] = [hi 32 | lo 32] of weight * 2^s, alpha) for
    rac_outproj_fwd: alpha = 2^-s / SPLIT_ACT_SCALE undoes the weight's and the activation image's power-of-two scalings.
    (None, None) if f16 cannot hold the weights or K is not a multiple of 32."""
    import math
    w = weight.detach().float().contiguous()
    N, K = w.shape
    amax = float(w.abs().max())
    if K % 32 != 0 or not w.is_cuda or not (amax > 0.0) or amax != amax or amax == float("inf"):
        return None, None
    s = 13 - math.frexp(amax)[1] + 1          # amax * 2^s in [2^13, 2^14)
    img = torch.empty(N, K // 32, 64, device=w.device, dtype=torch.float16)
    _lib.check(_lib.lib().rac_gemm_split_pack_fwd(_lib.ptr(w), _lib.ptr(img), N, K, float(2.0 ** s), _lib.stream_ptr()),
               "rac_gemm_split_pack_fwd")
    return img, 2.0 ** (-s) / SPLIT_ACT_SCALE


def value_proj_fused(maps, w_image, w_alpha, add=None, bias=None, q16=False):
    """maps f32 [F, 256, H, W] (channel-first BEV maps), w_image f16 [256, 8, 64] (pack_gemm_split_weight), w_alpha = 2^-s of the
    image, add f32 [H*W, 256] (frame-independent term) or bias [256] -> f32 [F, H*W, 256] = maps^T @ W^T + add (rac_value_proj_fwd);
    with ``q16`` the same result in the int16 block storage of quantize_values_i16: (q int16 [F, H*W, 4, 64], scale f32 [F, H*W, 4]),
    quantised in the kernel's epilogue (rac_value_proj_q16_fwd)."""
    _lib.require_gpu(maps, w_image, what="value_proj_fused")
    F_, C, H, W = maps.shape
    if maps.dtype != torch.float32 or not maps.is_contiguous() or w_image.dtype != torch.float16 or tuple(w_image.shape) != (256, 8, 64):
        raise RuntimeError("value_proj_fused: maps must be contiguous float32 [F,256,H,W], w_image f16 [256,8,64]")
    if add is not None and (add.dtype != torch.float32 or not add.is_contiguous() or tuple(add.shape) != (H * W, 256)):
        raise RuntimeError("value_proj_fused: add must be a contiguous float32 [H*W, 256] tensor")
    if q16:
        out = torch.empty(F_, H * W, 4, 64, device=maps.device, dtype=torch.int16)
        scale = torch.empty(F_, H * W, 4, device=maps.device, dtype=torch.float32)
    else:
        out = torch.empty(F_, H * W, 256, device=maps.device, dtype=torch.float32)
    ev = _lib.timer.record("value_proj_fwd") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    head = (_lib.ptr(maps), _lib.ptr(w_image), float(w_alpha), _lib.ptr(add) if add is not None else None,
            _lib.ptr(bias) if bias is not None else None)
    tail = (F_, C, H * W, 256, _lib.stream_ptr())
    if q16:
        rc, what = _lib.lib().rac_value_proj_q16_fwd(*head, _lib.ptr(out), _lib.ptr(scale), *tail), "rac_value_proj_q16_fwd"
    else:
        rc, what = _lib.lib().rac_value_proj_fwd(*head, _lib.ptr(out), *tail), "rac_value_proj_fwd"
    if ev:
        ev[1].record()
    _lib.check(rc, what)
    return (out, scale) if q16 else out


def generator_fused(x_image, w_image, bias, alpha, timer_name="mixing_generator_gemm", ld_out=None):
    """x_image f16 [M, K/32 * 64] (row_seg(split_lines=True) / add_ln(split_lines=True)), w_image f16 [N, K/32, 64] -> fp32 [M, N] =
    alpha * X @ W^T + bias (rac_generator_fwd), one launch."""
    _lib.require_gpu(x_image, w_image, what="generator_fused")
    M = x_image.shape[0]
    N, lines, _ = w_image.shape
    if x_image.dtype != torch.float16 or w_image.dtype != torch.float16 or x_image.numel() != M * lines * 64 or w_image.shape[2] != 64:
        raise RuntimeError("generator_fused: operand images do not match")
    ld_out = N if ld_out is None else int(ld_out)     # (row stride: a multiple of 4, >= N; the result has ld_out columns)
    out = torch.empty(M, ld_out, device=x_image.device, dtype=torch.float32)
    ev = _lib.timer.record(timer_name) if _lib.timer is not None and timer_name else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_generator_fwd(_lib.ptr(x_image), _lib.ptr(w_image), _lib.ptr(bias) if bias is not None else None, float(alpha),
                                      _lib.ptr(out), ld_out, M, N, lines * 32, _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_generator_fwd")
    return out


def outproj_fused(z_image, w_image, slices):
    """z_image f16 [M, K/32, 64], w_image f16 [N, K/32, 64] -> fp32 partials [slices, M, N] (unscaled), one launch."""
    _lib.require_gpu(z_image, w_image, what="outproj_fused")
    M, lines, _ = z_image.shape
    N = w_image.shape[0]
    if z_image.dtype != torch.float16 or w_image.dtype != torch.float16 or tuple(w_image.shape[1:]) != (lines, 64) \
            or z_image.shape[2] != 64 or lines % slices != 0:
        raise RuntimeError("outproj_fused: operand images do not match")
    out = torch.empty(slices, M, N, device=z_image.device, dtype=torch.float32)
    ev = _lib.timer.record("mixing_out_proj_gemm") if _lib.timer is not None else None
    if ev:
        ev[0].record()
    rc = _lib.lib().rac_outproj_fwd(_lib.ptr(z_image), _lib.ptr(w_image), _lib.ptr(out), M, N, lines * 32, slices, _lib.stream_ptr())
    if ev:
        ev[1].record()
    _lib.check(rc, "rac_outproj_fwd")
    return out


# ------------------------------------------------------------------------------------------- decode
def decode_fused(cls_scores, bbox_preds, max_num, post_center_range, score_threshold=None, out=None):
    """NMSFreeCoder.decode_single + get_bboxes' reshuffle for one sample in one launch (rac_decode_fwd):
    cls_scores [Q,C] logits, bbox_preds [Q,10] -> [max_num, 11] = (x, y, z_bottom, w, l, h, yaw, vx, vy, score, label),
    score = -1 on rows that fail the centre-range / score masks."""
    cls_scores, bbox_preds = cls_scores.contiguous(), bbox_preds.contiguous()
    _lib.require_gpu(cls_scores, bbox_preds, what="decode_fused")
    Q, C = cls_scores.shape
    if out is None:
        out = torch.empty(max_num, 11, device=cls_scores.device, dtype=torch.float32)
    rng = (ctypes.c_float * 6)(*[float(v) for v in post_center_range])
    rc = _lib.lib().rac_decode_fwd(_lib.ptr(cls_scores), _lib.ptr(bbox_preds), _lib.ptr(out), Q, C, int(max_num), rng,
                                   float(score_threshold or 0.0), int(bool(score_threshold)), _lib.stream_ptr())
    _lib.check(rc, "rac_decode_fwd")
    return out

