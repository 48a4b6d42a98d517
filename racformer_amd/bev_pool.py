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
        # The backward kernel wants the points grouped by the feature cell they read (what bev_pool.py:50-63
        # prepares): a stable sort by ranks_feat, then run lengths of equal keys.
        order = torch.argsort(ranks_feat, stable=True)
        ranks_feat, ranks_depth, ranks_bev = ranks_feat[order], ranks_depth[order], ranks_bev[order]
        interval_starts_bp, interval_lengths_bp = intervals_from_ranks(ranks_feat)
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


def intervals_from_ranks(ranks):
    """(starts, lengths), int32, of the runs of equal consecutive values in ``ranks`` -- the interval tables the
    pooling kernels consume (the reference's own test builds them the same way, bev_pool.py:158-166)."""
    if ranks.numel() == 0:
        empty = torch.zeros(0, dtype=torch.int32, device=ranks.device)
        return empty, empty.clone()
    _, counts = torch.unique_consecutive(ranks, return_counts=True)
    starts = torch.cumsum(counts, dim=0) - counts
    return starts.int(), counts.int()
