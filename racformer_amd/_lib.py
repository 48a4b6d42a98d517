"""ctypes binding of libracformer_hip.so (C-ABI in include/racformer_hip.h).

No fallback: if the library is missing, or a call fails, this raises.  Tensors are plumbing
(device memory + the current HIP stream); the kernels themselves are hand-written HIP.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RACFORMER_HIP_LIB: another build of the same library, for A/B experiments of tools/exp_*.py; never set in tests or bench)
LIB_PATH = os.environ.get("RACFORMER_HIP_LIB") or os.path.join(_HERE, "csrc", "libracformer_hip.so")
RAC_F32, RAC_BF16, RAC_I16 = 0, 1, 2
OUT_SQCP, OUT_BQGTPC = 0, 1
MIX_F32, MIX_F16X3 = 0, 1
_lib = None

# name -> (restype, argtypes); must list every symbol include/racformer_hip.h declares
_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
SIGNATURES = {
    "rac_abi_version": (_i, []),
    "rac_last_error": (ctypes.c_char_p, []),
    "rac_msmv_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rac_msda_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rac_regroup_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rac_regroup_multi_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rac_box_prep_fwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "rac_sampling4d_fwd": (_i, [_vp, _vp, _i] + [_vp] * 11 + [_i] * 3 + [_i] * 8 + [_vp, _vp] + [_f] * 4 + [_i, _i, _vp]),
    "rac_msmv_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp] + [_i] * 5 + [_vp]),
    "rac_msda_bwd": (_i, [_vp] * 9 + [_i] * 7 + [_vp]),
    "rac_bev_pool_v2_fwd": (_i, [_vp] * 8 + [_i, _i, _vp]),
    "rac_bev_pool_v2_bwd": (_i, [_vp] * 10 + [_i, _i, _vp]),
    "rac_add_ln_fwd": (_i, [_vp, _i, ctypes.c_int64, _i, _f] + [_vp] * 6 + [_i, _i, _i, _f, _i, _vp, _f, _i, _i, _vp]),
    "rac_pe_head_fwd": (_i, [_vp, _i] + [_vp] * 5 + [_i, _i, _f, _vp]),
    "rac_layer_boundary_fwd": (_i, [_vp] * 12 + [_i] * 4 + [_f, _f, _vp]),
    "rac_refine_fwd": (_i, [_vp] * 5 + [_i] * 3 + [_f, _vp]),
    "rac_head_finish_fwd": (_i, [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _i, _vp, _vp]),
    "rac_mixing_fwd": (_i, [_vp, _vp, _f, _vp, _vp, _f] + [_i] * 6 + [_f, _i, _vp]),
    "rac_mixing_sampled_fwd": (_i, [_vp, _vp, _i] + [_vp] * 10 + [_i] * 3 + [_i] * 8 + [_vp, _vp] + [_f] * 4 + [_i]
                               + [_vp, _f, _vp, _vp, _f, _i, _i, _f, _vp]),
    "rac_sasa_fwd": (_i, [_vp] * 5 + [_i] * 6 + [_vp, _vp]),
    "rac_decode_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _f, _i, _vp]),
    "rac_rowgemm_fwd": (_i, [_vp, _i, _i, _vp]),
    "rac_gemm_split_pack_fwd": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "rac_value_proj_fwd": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rac_value_proj_q16_fwd": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rac_outproj_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rac_generator_fwd": (_i, [_vp, _vp, _vp, _f, _vp, ctypes.c_int64, _i, _i, _i, _vp]),
    "rac_gru_gate_fwd": (_i, [_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp]),
    "rac_upsample2x_fwd": (_i, [_vp, _vp, ctypes.c_int64, _i, _i, _vp]),
    "rac_absmax_fwd": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "rac_conv_pack_fwd": (_i, [_vp, _vp, _vp] + [_i] * 6 + [_vp]),
    "rac_conv_pack_bias_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "rac_conv3x3_fwd": (_i, [_vp] * 5 + [_f, _vp] + [_i] * 5 + [_vp]),
    "rac_conv3x3_q16_fwd": (_i, [_vp] * 5 + [_f, _vp, _vp] + [_i] * 5 + [_vp]),
    "rac_fpn_conv_fwd": (_i, [_vp] * 4 + [_f, _vp] + [_i] * 5 + [_vp]),
    "rac_conv3x3s2_fwd": (_i, [_vp] * 4 + [_f, _vp] + [_i] * 7 + [_vp]),
    "rac_bev_sampling_fwd": (_i, [_vp] * 10 + [_i] * 4 + [_i] * 9 + [_vp, _vp, _f, _i, _vp]),
    "rac_bev_sampling_multi_fwd": (_i, [_i] + [_vp] * 9 + [_i] * 4 + [_i] * 9 + [_vp, _vp, _f, _i, _vp]),
    "rac_bev_sampling_multi_q16_fwd": (_i, [_i] + [_vp] * 10 + [_i] * 4 + [_i] * 9 + [_vp, _vp, _f, _vp]),
    "rac_quant_i16_fwd": (_i, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    "rac_conv_direct_fwd": (_i, [_vp, _vp]),
    "rac_upsample2x_image_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "rac_conv3x3_temporal_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp] + [_i] * 7 + [_vp]),
}


class RowSeg(ctypes.Structure):
    """rac_rowseg (include/racformer_hip.h)"""
    _fields_ = [("a", _vp), ("partial_stride", ctypes.c_int64), ("bias0", _vp), ("residual", _vp), ("gamma", _vp),
                ("beta", _vp), ("post", _vp), ("x_out", _vp), ("split_out", _vp), ("ld_a", _i), ("num_partials", _i),
                ("ld_res", _i), ("ld_post", _i), ("ld_xout", _i), ("relu", _i), ("split_pad", _i), ("split_layout", _i), ("a_scale", _f),
                ("eps", _f), ("split_scale", _f)]


class RowGemm(ctypes.Structure):
    """rac_rowgemm (include/racformer_hip.h)"""
    _fields_ = [("seg", RowSeg * 3), ("w", _vp), ("b", _vp), ("out", _vp), ("num_seg", _i), ("N", _i), ("ld_out", _i),
                ("relu_from", _i)]


class CdScale(ctypes.Structure):
    """rac_cd_scale (include/racformer_hip.h)"""
    _fields_ = [("amax", _vp), ("mul", _f), ("add", _f)]


class CdFrames(ctypes.Structure):
    """rac_cd_frames (include/racformer_hip.h)"""
    _fields_ = [("live", _i), ("stride", _i), ("first", _i)]


class ConvDirect(ctypes.Structure):
    """rac_conv_direct (include/racformer_hip.h)"""
    _fields_ = [("mode", _i), ("conv_stride", _i), ("N", _i), ("H", _i), ("W", _i), ("in_img", _vp), ("in_chunks_total", _i),
                ("in_chunk0", _i), ("chunks", _i), ("in_frames", CdFrames), ("in_scale", CdScale), ("ws", _vp), ("w_alpha", _f),
                ("Cout", _i), ("bias", _vp), ("out_img", _vp), ("out_chunks_total", _i), ("out_chunk0", _i),
                ("out_frames", CdFrames), ("out_scale", CdScale), ("out_f32", _vp), ("pixel_map", _vp), ("xpart", _vp),
                ("xpart_frames", CdFrames), ("h_prev", _vp), ("h_prev_frames", CdFrames), ("h_out", _vp),
                ("h_out_frames", CdFrames)]


CD_IMAGE, CD_F32, CD_GRU = 0, 1, 2


def lib():
    """The loaded C-ABI library.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"racformer_amd: HIP library not built ({LIB_PATH}); run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C racformer_amd/csrc`. "
                "There is no CPU fallback for the hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().rac_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def stream_ptr():
    """torch's current stream on the current device, as the raw hipStream_t the C-ABI takes.  Through the C bindings
    directly: torch.cuda.current_stream() builds a Stream object (9 us a call -- a fifth of the host time of a decode step,
    which issues ~140 launches)."""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:      # (private bindings: fall back to the public API if a torch release renames them)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return RAC_F32
    if t.dtype == torch.bfloat16:
        return RAC_BF16
    raise TypeError(f"racformer_amd: unsupported feature dtype {t.dtype} (float32 or bfloat16)")


def require_gpu(*tensors, what="op"):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"racformer_amd.{what}: tensor must be a CUDA tensor "
                               "(HIP device); the hot path has no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError(f"racformer_amd.{what}: tensor has to be contiguous")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


# ---- optional per-kernel timing hook (bench.py): events recorded on the launch stream ---------
class KernelTimer:
    """Brackets selected launches with HIP events on torch's current stream (the stream the
    C-ABI launches on) and reports the mean elapsed time per launch after a synchronise.
    ``only``: names to time (None = every instrumented launch).  An event pair costs a 5-10 us bubble in the queue (measured
    with rocprofv3: the marker packets drain the pipeline), so bench.py times only the dominant kernel inside its timed
    region and the other instrumented launches in extra, untimed passes."""

    def __init__(self, only=None):
        self.events = {}
        self.only = set(only) if only is not None else None

    def record(self, name):
        if self.only is not None and name not in self.only:
            return None
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((start, end))
        return start, end

    def mean_ms(self, name):
        ev = self.events.get(name, [])
        if not ev:
            return None
        return sum(s.elapsed_time(e) for s, e in ev) / len(ev)

    def reset(self):
        self.events = {}


timer = None  # set to a KernelTimer by bench.py for the timed region
