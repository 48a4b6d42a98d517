"""Drop-in for ``models/csrc/bev_pool_v2/bev_pool.py`` of the reference ("next" row f2):
``bev_pool_v2`` and the ``QuickCumsumCuda`` autograd Function (forward AND backward), on the HIP kernels
``rac_bev_pool_v2_fwd`` / ``rac_bev_pool_v2_bwd``.  Same argument order and dtypes handling as
bev_pool.py:16-92; the ONNX/TensorRT symbolic (:95-144) is out of scope."""
import torch

from . import _lib


class QuickCumsumCuda(torch.autograd.Function):
    """BEVPoolv2 for the Lift-Splat-Shoot view transformation (bev_pool.py:11-84)."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
        ranks_bev = ranks_bev.int()
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        ranks_depth = ranks_depth.contiguous().int()
        ranks_feat = ranks_feat.contiguous().int()
        interval_lengths = interval_lengths.contiguous().int()
        interval_starts = interval_starts.contiguous().int()
        _lib.require_gpu(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts,
                         what="bev_pool_v2")
        out = feat.new_zeros(bev_feat_shape)
        c = feat.shape[-1]
        rc = _lib.lib().rac_bev_pool_v2_fwd(_lib.ptr(depth), _lib.ptr(feat), _lib.ptr(out), _lib.ptr(ranks_depth),
                                            _lib.ptr(ranks_feat), _lib.ptr(ranks_bev), _lib.ptr(interval_lengths),
                                            _lib.ptr(interval_starts), c, interval_lengths.shape[0], _lib.stream_ptr())
        _lib.check(rc, "rac_bev_pool_v2_fwd")
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        # regroup the points by feature index (bev_pool.py:50-63)
        order = ranks_feat.argsort()
        ranks_feat, ranks_depth, ranks_bev = ranks_feat[order], ranks_depth[order], ranks_bev[order]
        kept = torch.ones(ranks_bev.shape[0], device=ranks_bev.device, dtype=torch.bool)
        kept[1:] = ranks_feat[1:] != ranks_feat[:-1]
        interval_starts_bp = torch.where(kept)[0].int()
        interval_lengths_bp = torch.zeros_like(interval_starts_bp)
        interval_lengths_bp[:-1] = interval_starts_bp[1:] - interval_starts_bp[:-1]
        interval_lengths_bp[-1] = ranks_bev.shape[0] - interval_starts_bp[-1]
        ranks_depth, ranks_feat, ranks_bev = ranks_depth.contiguous(), ranks_feat.contiguous(), ranks_bev.contiguous()
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        out_grad = out_grad.contiguous()
        rc = _lib.lib().rac_bev_pool_v2_bwd(_lib.ptr(out_grad), _lib.ptr(depth_grad), _lib.ptr(feat_grad), _lib.ptr(depth),
                                            _lib.ptr(feat), _lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev),
                                            _lib.ptr(interval_lengths_bp), _lib.ptr(interval_starts_bp), feat.shape[-1],
                                            interval_lengths_bp.shape[0], _lib.stream_ptr())
        _lib.check(rc, "rac_bev_pool_v2_bwd")
        return depth_grad, feat_grad, None, None, None, None, None, None


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts, interval_lengths):
    """bev_pool.py:87-92: -> [B, C, Z, Y, X]"""
    x = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                              interval_lengths)
    return x.permute(0, 4, 1, 2, 3).contiguous()


def intervals_from_ranks(ranks_bev):
    """interval_starts / interval_lengths of consecutive equal ranks_bev (as the reference's own test builds
    them, bev_pool.py:158-166)."""
    kept = torch.ones(ranks_bev.shape[0], device=ranks_bev.device, dtype=torch.bool)
    kept[1:] = ranks_bev[1:] != ranks_bev[:-1]
    starts = torch.where(kept)[0].int()
    lengths = torch.zeros_like(starts)
    lengths[:-1] = starts[1:] - starts[:-1]
    lengths[-1] = ranks_bev.shape[0] - starts[-1]
    return starts, lengths
