"""One sample's hot path as ONE submission: the ~140 kernel launches of regroup + hoisted prologue + six decoder layers +
NMS-free decode (racformer_head.py:82-134 -> racformer_transformer.py:95-142 -> nms_free_coder.py:37-88) captured once
into a HIP graph and replayed per sample.

Why it can be captured as it stands: every entry point of the C-ABI takes a stream, allocates nothing and never
synchronises (include/racformer_hip.h), descriptors travel as kernel arguments, and all weight-derived operands are
cached on the parameters' versions at the first (warm-up) forward.  What is NOT in the graph, by design:

* the per-sample host arithmetic of racformer_transformer.py:99-109 (float64 timestamps -> float32 time_diff, the
  lidar2img stack): ``replay(img_metas)`` stages it into the SAME device block the captured kernels read (one pinned
  asynchronous copy in front of the graph launch);
* the RCCL all-gather of the detections (racformer_amd/dp.py): issued by the caller on the replay's output;
* HIP-event brackets of bench.py (the dominant kernel is timed in separate, uncaptured steps).

The captured kernels read the producer's tensors in place: the feature pyramid and the two BEV maps a step was captured
with are the buffers the producer has to write the next sample into (``inputs``); a caller that hands over other
tensors pays a device copy into them (``replay(mlvl_feats=...)``)."""
import numpy as np
import torch

from . import _lib


class CapturedForward:
    """``fn()`` (kernel launches on torch's current stream over tensors that stay where they are; no host synchronisation)
    warmed up, captured once and replayable: ``replay()`` -> what ``fn`` returned at capture (the same tensors, rewritten)."""

    def __init__(self, fn, device, warmup=2):
        _lib.lib()
        if _lib.timer is not None:
            raise RuntimeError("racformer_amd.graph: event brackets (racformer_amd._lib.timer) cannot be captured; clear the timer")
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):       # packs weights, sets kernel attributes, lets the library convolutions pick their plans
                fn()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs


class CapturedStep:
    """head.forward + head.get_detections_fixed of one sample per GPU as a replayable HIP graph."""

    _serial = [0]

    def __init__(self, head, mlvl_feats, lss_bev_feats, radar_bev_feats, img_metas, warmup=2, decode=True, own_scratch=False):
        """``own_scratch``: the plan gets reusable scratch buffers of its own (racformer_amd.fused.scratch_namespace) -- required
        for a plan that is replayed on another stream beside other plans (several samples in flight); plans that follow each
        other on one stream share them."""
        if not lss_bev_feats.is_cuda:
            raise RuntimeError("racformer_amd.graph: capture needs device tensors (the hot path has no CPU fallback)")
        self.head, self.decode = head, decode
        self.inputs = (list(mlvl_feats), lss_bev_feats, radar_bev_feats)
        dec = head.transformer.decoder
        self.B, self.device = lss_bev_feats.shape[0], lss_bev_feats.device
        # metas staged once, outside the graph; the captured forward finds them staged and copies nothing
        self.metas = [dict(m) for m in img_metas]
        dec.stage_metas(self.metas, self.B, self.device)
        m0 = self.metas[0]
        self._meta_dev = (m0["time_diff"], m0["time_diff_safe"], m0["lidar2img"])
        self._meta_block = m0.get("_rac_meta_block")     # the three of them as one contiguous block (None: lidar2img came pre-staged)

        def run():
            preds = head(list(self.inputs[0]), self.inputs[1], self.inputs[2], self.metas)
            return preds, (head.get_detections_fixed(preds) if decode else None)

        from .fused import scratch_namespace
        CapturedStep._serial[0] += 1
        self._scratch_key = ("captured_step", CapturedStep._serial[0]) if own_scratch else None
        with scratch_namespace(self._scratch_key):
            self._cap = CapturedForward(run, self.device, warmup)
        self.graph = self._cap.graph
        self.preds, self.det = self._cap.outputs

    def close(self):
        """Releases the graph and, if the plan owned scratch buffers, those (they are otherwise kept for reuse by later forwards)."""
        from .fused import release_scratch
        self._cap = self.graph = self.preds = self.det = None
        if self._scratch_key is not None:
            release_scratch(self._scratch_key)
            self._scratch_key = None

    def _stage(self, img_metas):
        """time_diff / time_diff_safe / lidar2img of a new sample into the device block the captured kernels read."""
        dec = self.head.transformer.decoder
        ts = np.array([m["img_timestamp"] for m in img_metas], dtype=np.float64).reshape(self.B, -1, dec.num_cams)
        td = np.mean(ts[:, :1, :] - ts, axis=-1).astype(np.float32)
        td_safe = td.copy()
        td_safe[td_safe < 1e-5] = 1.0
        l2i = np.asarray([m["lidar2img"].cpu().numpy() if isinstance(m["lidar2img"], torch.Tensor) else m["lidar2img"]
                          for m in img_metas]).astype(np.float32)
        for dst, src in zip(self._meta_dev, (td, td_safe, l2i)):
            if tuple(dst.shape) != src.shape:
                raise RuntimeError(f"racformer_amd.graph: metas of another shape than the captured ones {tuple(dst.shape)} vs {src.shape}")
        if self._meta_block is not None:                 # one pinned block, one asynchronous copy (as the eager plan stages it)
            flat = torch.from_numpy(np.concatenate([td.ravel(), td_safe.ravel(), l2i.ravel()]))
            self._meta_block.copy_(flat.pin_memory(), non_blocking=True)
        else:
            for dst, src in zip(self._meta_dev, (td, td_safe, l2i)):
                dst.copy_(torch.from_numpy(src).pin_memory(), non_blocking=True)

    def replay(self, img_metas=None, mlvl_feats=None, lss_bev_feats=None, radar_bev_feats=None):
        """-> (preds, det): the captured step on what the input buffers hold now.  ``img_metas``: a new sample's timestamps
        and projection matrices (staged in front of the launch).  Feature tensors other than the captured buffers are
        copied into them (a producer should write into ``self.inputs`` instead)."""
        if img_metas is not None:
            self._stage(img_metas)
        if mlvl_feats is not None:
            for dst, src in zip(self.inputs[0], mlvl_feats):
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
        for dst, src in ((self.inputs[1], lss_bev_feats), (self.inputs[2], radar_bev_feats)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.preds, self.det
