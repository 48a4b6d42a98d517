"""N>1 path on CPU: round-robin sharding + all-gather + rank-interleaved merge with the gloo
backend, world_size 2 (the GPU box runs the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from racformer_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_samples, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = dp.shard_indices(num_samples, rank, world)
    # a fake fixed-shape detection block per local sample: row 0 encodes the dataset index
    det = torch.zeros(len(mine), 4, 11)
    for j, idx in enumerate(mine):
        det[j, :, 0] = float(idx)
        det[j, :, 9] = 0.5
    gathered = dp.all_gather_detections(det)
    merged = dp.merge_interleaved(gathered, num_samples)
    if rank == 0:
        q.put((mine, merged[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_semantics():
    assert dp.shard_indices(5, 0, 2) == [0, 2, 4]
    assert dp.shard_indices(5, 1, 2) == [1, 3, 0]          # padded by wrap-around
    assert dp.shard_indices(8, 3, 4) == [3, 7]
    assert dp.shard_indices(0, 0, 2) == []
    for n, w in ((7, 3), (16, 8), (1, 8)):
        got = sorted(i for r in range(w) for i in dp.shard_indices(n, r, w))
        assert set(got) == set(range(n))


def test_single_process_gather_is_identity():
    det = torch.rand(2, 3, 11)
    assert torch.equal(dp.all_gather_detections(det), det[None])
    res = dp.detections_to_results(torch.tensor([[0.] * 9 + [0.7, 3.0], [0.] * 9 + [-1.0, 1.0]]))
    assert res["scores_3d"].tolist() == [0.699999988079071] and res["labels_3d"].tolist() == [3]


def test_gloo_world2_gather_and_merge():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    num_samples = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_samples, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine, merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert mine == [0, 2, 4]
    assert merged == [0.0, 1.0, 2.0, 3.0, 4.0]            # dataset order restored, padding dropped
