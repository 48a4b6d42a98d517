"""Data-parallel inference plumbing: whole samples are sharded round-robin over the ranks of one
node and the fixed-shape detections are all-gathered (RCCL over xGMI on the GPU box, gloo in the
CPU tests).  Mirrors what ``val.py:106-135`` does through mmdet's ``DistributedSampler(shuffle=
False)`` + ``multi_gpu_test`` (round-robin indices padded to a multiple of the world size, parts
re-interleaved rank by rank, truncated to the dataset length), with one collective per step
instead of pickles on a shared file system."""
import torch
import torch.distributed as dist


def shard_indices(num_samples, rank, world_size):
    """Indices of this rank: ``rank::world`` over the dataset padded (by wrapping around) to a
    multiple of the world size -- mmdet DistributedSampler semantics (loaders/builder.py:18-28)."""
    per_rank = -(-num_samples // world_size)
    total = per_rank * world_size
    idx = list(range(num_samples))
    idx += idx[: total - num_samples] if num_samples else []
    return idx[rank:total:world_size]


def all_gather_detections(det, group=None, force_collective=False):
    """det [S_local, K, 11] on every rank (same shape) -> [world, S_local, K, 11].  A world of one rank returns the local
    block without a collective unless ``force_collective`` (the single-GPU rehearsal of the RCCL path:
    tests/test_rccl_gpu.py, ``bench.py --force-collective``)."""
    if not (dist.is_available() and dist.is_initialized()):
        if force_collective:
            raise RuntimeError("all_gather_detections: force_collective needs an initialised process group")
        return det[None]
    if dist.get_world_size(group) == 1 and not force_collective:
        return det[None]
    out = torch.empty((dist.get_world_size(group),) + tuple(det.shape), device=det.device, dtype=det.dtype)
    # per-rank views of one contiguous buffer: portable across RCCL and gloo, one collective
    dist.all_gather(list(out.unbind(0)), det.contiguous(), group=group)
    return out


def merge_interleaved(gathered, num_samples):
    """[world, S_local, ...] -> [num_samples, ...] in dataset order (sample i lives at
    [i % world, i // world]); drops the wrap-around padding (mmdet collect_results ordering)."""
    world, s_local = gathered.shape[:2]
    merged = gathered.transpose(0, 1).reshape((world * s_local,) + tuple(gathered.shape[2:]))
    return merged[:num_samples]


def detections_to_results(det):
    """[K,11] fixed rows -> dict of the kept boxes (score >= 0), as get_bboxes would return."""
    keep = det[:, 9] >= 0
    return {"boxes_3d": det[keep, :9], "scores_3d": det[keep, 9], "labels_3d": det[keep, 10].long()}
