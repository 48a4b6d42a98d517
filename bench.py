#!/usr/bin/env python3
"""bench.py -- throughput of RaCFormer's query-decoder hot path on MI355X.

One "step" = one pass of the hot path over one synthetic nuScenes-shaped sample per GPU
(racformer_r50_nuimg_704x256_f8: 6 cams x 8 frames, 704x256, 900 queries): pyramid regroup ->
hoisted BEV value streams -> 6 decoder layers (scale-adaptive self-attention, radar/LSS BEV
deformable attention, adaptive 4D sampling, adaptive mixing, FFN, heads) -> NMS-free top-300
decode -> (N>1) RCCL all-gather of the fixed-shape detections.  Inputs (FPN pyramid, BEV maps,
weights) are resident in HBM before the timed region.  Data-parallel: every rank processes its
own sample, so scaling is weak and `value` = samples of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract: see the task statement); `roofline` is for the
dominant hand-written kernel (msmv sampling), timed live with HIP events on the launch stream;
`cpu_baseline` is the CPU oracle (port of the reference's CPU forward) on this box's host cores.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its own N ranks (one fresh process per GPU
through torch.distributed.run, before this process has touched the GPU) and exits with their status; started by a launcher
(RANK / LOCAL_RANK / WORLD_SIZE set, as the driver does) it is one of the ranks.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from racformer_amd import _lib, dp, synthetic as syn  # noqa: E402
from racformer_amd.head import RaCFormer_head  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate


def build_head(cfg, device, feature_dtype=torch.float32):
    head = RaCFormer_head(
        num_classes=cfg.num_classes, in_channels=cfg.embed_dims, num_query=cfg.num_query,
        num_clusters=cfg.num_clusters, code_size=cfg.code_size,
        transformer=dict(type="RaCFormerTransformer", **cfg.transformer_kwargs()),
        bbox_coder=dict(type="NMSFreeCoder", post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        pc_range=list(cfg.pc_range), max_num=300, score_threshold=0.05, num_classes=cfg.num_classes))
    syn.fill_params(head.transformer, 0)
    with torch.no_grad():
        head.label_enc.weight.copy_(torch.from_numpy(syn.rng_normal(77, tuple(head.label_enc.weight.shape), 0.1)))
        # a trained head's query embedding is a tame polar grid; nn.Embedding's N(0,1) default would put
        # e^N(0,1)-sized boxes and random yaw in every query (see racformer_amd/synthetic.py)
        head.init_query_bbox.weight.copy_(syn.make_queries(cfg, 0)[0][0])
    head.transformer.decoder.feature_dtype = feature_dtype
    return head.eval().to(device)


def host_threads():
    """Threads the CPU baseline may use: the cgroup CPU quota of this box (the GPU pool gives a
    1-GPU job about 16 cores of a 256-core host), never the raw core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def source_sha(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(kernel_substr, source_rel, config):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC summary of this same command (separate
    FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads;
    tools/pmc_traffic.py).  The summary records the sha256 of the kernel's source file at capture time: a summary taken from
    another version of the kernel is NOT reported (None) -- the PMC passes cannot run inside this process (rocprofv3 wraps
    the command), so staleness is the one thing that has to be excluded here."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_{config}.json")))
    if not files:
        return None, f"no profiles/*_pmc_traffic_{config}.json"
    data = json.load(open(files[-1]))
    name = os.path.basename(files[-1])
    want = source_sha(source_rel)
    if data.get("_source_sha", {}).get(source_rel) != want:
        return None, f"{name} was captured from another version of {source_rel} (stale): not reported"
    for k, v in data.items():
        if kernel_substr in k and isinstance(v, dict) and v.get("read_bytes_corrected") is not None and v.get("write_bytes") is not None:
            return v["read_bytes_corrected"] + v["write_bytes"], \
                f"{name}: 2*FETCH_SIZE + WRITE_SIZE, avg per launch (rocprofv3 --pmc, separate passes; {source_rel} sha {want})"
    return None, "kernel not in " + name


def msmv_algorithmic_bytes(loc, feat_shapes, elt_bytes, out_elems):
    """sum_l min(in-range points x 4 taps x C x s, level bytes) + loc + weights +
    output; `loc` is the [S,Q,P,3] tensor the kernel was launched with."""
    S, Q, P, _ = loc.shape
    n_pts = S * Q * P
    L = len(feat_shapes)
    total, fracs = 0.0, []
    u, v = loc[..., 0], loc[..., 1]
    for (s_, n_, h, w, c) in feat_shapes:
        h_im, w_im = v * (h - 1), u * (w - 1)
        n_in = int(((h_im > -1) & (w_im > -1) & (h_im < h) & (w_im < w)).sum())
        fracs.append(n_in / max(n_pts, 1))
        total += min(n_in * 4 * c * elt_bytes, s_ * n_ * h * w * c * elt_bytes)
    total += n_pts * 3 * 4 + n_pts * L * 4 + out_elems * 4
    return total, fracs


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def launch_ranks(args):
    """Parent of a multi-GPU run: N fresh child processes (one rank per GPU, RCCL) through torch.distributed.run, started
    BEFORE this process makes any HIP call; the parent only waits and passes the children's exit status on (the reference's
    dist_test.sh:1 / val.py:93-135 launch the same way).  Never an exec of a process that has initialised the GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def stress_block(cfg, device, reps=20):
    """BASELINE config 2 / SURVEY 8(d) uniform-stress set on the two operator-boundary kernels (rac_msmv_fwd, rac_msda_fwd)
    and a scattered-query set on the fused kernel: N(0,1) feature maps (no spatial smoothness), u,v ~ U(-0.05,1.05),
    view ~ randint(N), weights = softmax(N(0,1)).  HIP-event timing on the launch stream, algorithmic bytes of 8(d)."""
    from racformer_amd.msda import msda_forward
    from racformer_amd.msmv import msmv_forward
    g = torch.Generator(device="cpu").manual_seed(1234)
    S, N, Q, C = cfg.batch * cfg.num_frames * cfg.num_groups, cfg.num_cams, cfg.num_query, cfg.channels
    P, L = cfg.num_points * cfg.img_depth_num, cfg.num_levels
    feats = [torch.randn(S, N, h, w, C, generator=g).to(device) for (h, w) in cfg.fpn_hw]
    loc = torch.rand(S, Q, P, 3, generator=g) * 1.1 - 0.05
    loc[..., 2] = torch.randint(0, N, (S, Q, P), generator=g).float() / (N - 1)
    w = torch.softmax(torch.randn(S, Q, P, L, generator=g), dim=-1)
    loc, w = loc.to(device), w.to(device)
    out = {}

    def timed(fn):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return statistics.median(a.elapsed_time(b) for a, b in ev)

    ms = timed(lambda: msmv_forward(feats, loc, w, out_layout=_lib.OUT_BQGTPC, num_frames=cfg.num_frames, num_groups=cfg.num_groups))
    b_alg, fr = msmv_algorithmic_bytes(loc.cpu(), [tuple(f.shape) for f in feats], 4, S * Q * C * P)
    out["rac_msmv_fwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "in_range_fraction": float(np.mean(fr)),
                           "achieved": b_alg / (ms * 1e-3) / 1e9, "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del feats
    # MSDA: [B*T, HW, heads, 64] value stream, 20 points per head, uniform locations
    bs, heads, Pm = cfg.batch * cfg.num_frames, 4, cfg.num_points_bev * cfg.bev_depth_num
    H, W = cfg.bev_hw
    value = torch.randn(bs, H * W, heads, 64, generator=g).to(device)
    mloc = (torch.rand(bs, Q, heads, 1, Pm, 2, generator=g) * 1.1 - 0.05).to(device)
    attn = torch.softmax(torch.randn(bs, Q, heads, 1, Pm, generator=g), dim=-1).to(device)
    ms = timed(lambda: msda_forward(value, [[H, W]], [0], mloc, attn))
    taps = bs * Q * heads * Pm * 4 * 64 * 4
    b_alg = min(taps, value.numel() * 4) + mloc.numel() * 4 + attn.numel() * 4 + bs * Q * heads * 64 * 4
    out["rac_msda_fwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "achieved": b_alg / (ms * 1e-3) / 1e9,
                           "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # Rows f2 / f4 at the reference's sizes (round-3 verdict, item 7): the backward operators at the launch shapes above, and
    # bev_pool_v2 on the f8 Lift-Splat shape.  Timed is the operator as the autograd Function runs it (gradient buffers zeroed /
    # allocated inside).  The feature / value gradients are scattered with memory-side float atomics, whose chip-wide rate is
    # ~1.3 TB/s of added bytes (MI355X_MICROARCH.md): `atomic_frac` prices the scattered bytes against THAT, `frac` all bytes against HBM.
    from racformer_amd.msda import MultiScaleDeformableAttnFunction_fp32 as _F32
    from racformer_amd.msmv import msmv_backward
    ATOMIC_PEAK_GBS = 1300.0
    g_out = torch.randn(bs, Q, heads * 64, generator=g).to(device)
    v_, l_, a_ = value.requires_grad_(), mloc.requires_grad_(), attn.requires_grad_()
    o_ = _F32.apply(v_, torch.tensor([[H, W]], device=device), torch.tensor([0], device=device), l_, a_, 64)
    ms = timed(lambda: torch.autograd.grad(o_, (v_, l_, a_), g_out, retain_graph=True))
    scat = taps                                                     # every tap adds one 256-byte row of the value gradient
    b_alg = scat + min(taps, value.numel() * 4) + g_out.numel() * 4 + 2 * (mloc.numel() + attn.numel()) * 4 + value.numel() * 4
    out["rac_msda_bwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "scattered_atomic_bytes": scat, "achieved": b_alg / (ms * 1e-3) / 1e9,
                           "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "atomic_frac": scat / (ms * 1e-3) / 1e9 / ATOMIC_PEAK_GBS,
                           "note": "incl. zeroing the 134 MB value gradient; grad_loc / grad_attn one writer per element"}
    del value, v_, o_
    feats = [torch.randn(S, N, h, w, C, generator=g).to(device) for (h, w) in cfg.fpn_hw]
    g_out = torch.randn(S, Q, C, P, generator=g).to(device)
    ms = timed(lambda: msmv_backward(g_out, feats, loc, w))
    fb, fr = msmv_algorithmic_bytes(loc.cpu(), [tuple(f.shape) for f in feats], 4, S * Q * C * P)
    scat = int(sum(fr) * S * Q * P) * 4 * C * 4                      # in-range (point, level) pairs x 4 taps x 256 B
    b_alg = fb + scat + sum(f.numel() for f in feats) * 4 + (loc.numel() + w.numel()) * 4
    out["rac_msmv_bwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "scattered_atomic_bytes": scat, "achieved": b_alg / (ms * 1e-3) / 1e9,
                           "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "atomic_frac": scat / (ms * 1e-3) / 1e9 / ATOMIC_PEAK_GBS,
                           "note": "incl. zeroing the 735 MB feature gradient; grad_loc / grad_weight one writer per element"}
    del feats, g_out
    from racformer_amd.bev_pool import QuickCumsumCuda, intervals_from_ranks
    Nc, Dd, Hh, Ww, Cc, Gg = 6, 96, 16, 44, 256, 128                 # configs/racformer_r50_nuimg_704x256_f8.py:55-63,100-104
    rd, rf, rb = (t_.to(device) for t_ in syn.make_lss_ranks(Nc, Dd, Hh, Ww, Gg))
    depth = torch.softmax(torch.randn(1, Nc, Dd, Hh, Ww, generator=g), 2).to(device).requires_grad_()
    feat = torch.randn(1, Nc, Hh, Ww, Cc, generator=g).to(device).requires_grad_()
    gs, gl = intervals_from_ranks(rb)
    shape = (1, 1, Gg, Gg, Cc)
    with torch.no_grad():
        ms = timed(lambda: QuickCumsumCuda.apply(depth, feat, rd, rf, rb, shape, gs, gl))
    n_pts = rd.numel()
    b_alg = n_pts * (Cc * 4 + 4 + 12) + gs.numel() * 8 + 2 * Gg * Gg * Cc * 4            # feature row + depth + 3 ranks per point; cells zeroed + written
    out["rac_bev_pool_v2_fwd"] = {"avg_launch_ms": ms, "points": n_pts, "cells": int(gs.numel()), "algorithmic_bytes": b_alg,
                                  "achieved": b_alg / (ms * 1e-3) / 1e9, "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "f8 Lift-Splat shape: 6 cams x 96 depth bins x 16x44 -> 128x128 cells x 256 channels; feature rows are "
                                          "re-read out of L2 (1.3 MB feature map)"}
    o_ = QuickCumsumCuda.apply(depth, feat, rd, rf, rb, shape, gs, gl)
    g_o = torch.randn(tuple(o_.shape), generator=g).to(device)
    ms = timed(lambda: torch.autograd.grad(o_, (depth, feat), g_o, retain_graph=True))
    b_alg = n_pts * (2 * Cc * 4 + 4 + 12 + 4) + feat.numel() * 4 + depth.numel() * 4
    out["rac_bev_pool_v2_bwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "achieved": b_alg / (ms * 1e-3) / 1e9, "unit": "GB/s",
                                  "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "incl. the stable sort by feature cell and the interval tables (torch ops, as bev_pool.py:50-63 prepares "
                                          "them); no atomics: one writer per gradient element"}
    del depth, feat, o_, g_o
    # The FUSED kernel (keypoints + projection + view selection + gather) on a scattered-query set: boxes anywhere in the
    # range (theta, d uniform; z, sizes, yaw, velocity random), the three sampling Linears' outputs ~ N(0,1) (offsets, ray
    # jitter, level logits -- what a random-feature query would produce), N(0,1) maps, the rig's own cameras and timestamps.
    # Neighbouring queries then have nothing in common: no L2 locality between the rows of a workgroup.
    from racformer_amd.fused import sampling4d_fused
    T, G, NP, D = cfg.num_frames, cfg.num_groups, cfg.num_points, cfg.img_depth_num
    feats = [torch.randn(S, N, h, w, C, generator=g).to(device) for (h, w) in cfg.fpn_hw]
    qb = torch.zeros(cfg.batch, Q, 10)
    qb[..., 0:2] = torch.rand(cfg.batch, Q, 2, generator=g)
    qb[..., 2] = 0.3 + 0.4 * torch.rand(cfg.batch, Q, generator=g)
    qb[..., 3:6] = 0.5 + 0.4 * torch.randn(cfg.batch, Q, 3, generator=g)
    yaw = 2 * np.pi * torch.rand(cfg.batch, Q, generator=g)
    qb[..., 6], qb[..., 7] = torch.sin(yaw), torch.cos(yaw)
    qb[..., 8:10] = 2.0 * torch.randn(cfg.batch, Q, 2, generator=g)
    offs = torch.randn(cfg.batch, Q, G * P * 3, generator=g)
    rays = torch.randn(cfg.batch, Q, D, generator=g)
    scl = torch.randn(cfg.batch, Q, G * T * P * L, generator=g)
    metas = syn.make_img_metas(cfg)
    ts = np.array([m["img_timestamp"] for m in metas], dtype=np.float64).reshape(cfg.batch, -1, N)
    td = torch.from_numpy(np.mean(ts[:, :1, :] - ts, axis=-1).astype(np.float32)).to(device)
    l2i = torch.from_numpy(np.asarray([m["lidar2img"] for m in metas]).astype(np.float32)).to(device)
    qb, offs, rays, scl = qb.to(device), offs.to(device), rays.to(device), scl.to(device)
    H_img, W_img = cfg.image_hw

    def fused(debug=False):
        return sampling4d_fused(feats, qb, offs, rays, scl, td, l2i, T, G, NP, D, list(cfg.pc_range), cfg.d_region_list[0], H_img, W_img,
                                debug=debug)
    ms = timed(fused)
    _, loc_f, _ = fused(debug=True)
    torch.cuda.synchronize()
    b_alg, fr = msmv_algorithmic_bytes(loc_f.cpu(), [tuple(f.shape) for f in feats], 4, S * Q * C * P)
    out["rac_sampling4d_fwd"] = {"avg_launch_ms": ms, "algorithmic_bytes": b_alg, "in_range_fraction": float(np.mean(fr)),
                                 "achieved": b_alg / (ms * 1e-3) / 1e9, "unit": "GB/s", "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "set": "scattered queries: boxes uniform over the range, random yaw / size / velocity, N(0,1) "
                                        "offsets and level logits, N(0,1) maps"}
    return out


def plumbing_only(args, rank, world):
    """CPU rehearsal of the multi-rank plumbing (tests/test_bench_launcher.py): same launcher, rendezvous, barrier / timing
    protocol, all-gather and rank-interleaved merge as the real run, over gloo, with a stand-in detection block instead of
    the HIP hot path.  Its JSON line says so and carries no throughput."""
    dist.init_process_group("gloo", init_method="env://")
    if os.environ.get("RAC_BENCH_TEST_FAIL") and rank == world - 1:
        raise SystemExit("plumbing-only: simulated rank failure (tests/test_bench_launcher.py)")
    samples = 5
    mine = dp.shard_indices(samples, rank, world)

    def step(i):
        det = torch.zeros(1, 300, 11)
        det[0, :, 0] = float(mine[i % len(mine)])      # the dataset index rides in the first column
        det[0, :, 9] = 0.5
        return dp.all_gather_detections(det)

    dist.barrier()
    t0 = time.perf_counter()
    outs = [step(i) for i in range(len(mine))]
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, elapsed)
    merged = dp.merge_interleaved(torch.cat(outs, dim=1), samples)          # [world, S_local, ...] -> dataset order
    if rank == 0:
        print(json.dumps({"metric": "plumbing-only rehearsal (no throughput)", "value": None, "unit": "samples/s",
                          "n_gpus": world, "data": "plumbing-only", "backend": "gloo",
                          "merged_sample_order": merged[:, 0, 0].tolist(), "gathered_shape": list(outs[0].shape),
                          "per_rank_ms": [1e3 * float(t) for t in per_rank]}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 80 timed steps = 20 per lane at four samples in flight (0.3 s).  With 20 (five per lane) the drain at the end of the
    # timed region -- the lanes do not finish together, the last replays run alone -- cost 2 % of the measured rate (275.6 against
    # 280.2 samples/s at 80 and 281.1 at 200 steps on one box); warm-up: two replays per lane (the plans were warmed up when captured)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="f8", choices=["f8", "f8_3cam"])
    ap.add_argument("--feature-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the uniform-stress timing of the operator kernels")
    ap.add_argument("--stress-only", action="store_true", help="print the uniform-stress block alone (kernel A/B runs) and exit")
    ap.add_argument("--pregrouped", action="store_true",
                    help="feed the pyramid already in the sampling layout (producer-side layout, row f2): no regroup")
    ap.add_argument("--no-graph", action="store_true",
                    help="time the eager plan (one Python-issued launch per kernel) instead of the captured HIP graph")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="samples in flight per GPU: that many captured plans on streams of their own, replayed round-robin "
                         "(the latency-bound launches of one sample run beside the bandwidth-bound ones of the other)")
    ap.add_argument("--same-inputs-per-lane", action="store_true",
                    help="A/B switch: every lane replays lane 0's input buffers (round 3's arrangement) instead of a sample of its own")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the RCCL process group and issue the all-gather even in a world of one rank "
                         "(single-GPU rehearsal of the N > 1 path)")
    ap.add_argument("--value-storage", default="f32", choices=["f32", "i16"],
                    help="storage of the two hoisted BEV value streams (internal tensors of the path): the reference's fp32 maps (the "
                         "product default and the headline) or the opt-in int16 block storage written by their producers' epilogues")
    ap.add_argument("--blas", default="", help="torch.backends.cuda.preferred_blas_library override (experiment)")
    ap.add_argument("--plumbing-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))                      # parent: no GPU call has been made in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_pg = world > 1 or args.force_collective
    if use_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", init_method="env://", device_id=device)  # RCCL on ROCm
    _lib.lib()  # fail loudly if the HIP library is missing
    if args.blas:
        torch.backends.cuda.preferred_blas_library(args.blas)

    cfg = syn.F8 if args.config == "f8" else syn.F8_3CAM
    fdt = torch.float32 if args.feature_dtype == "f32" else torch.bfloat16
    if args.stress_only:
        print(json.dumps({"roofline_stress": stress_block(cfg, device)}))
        return
    head = build_head(cfg, device, fdt)
    head.transformer.decoder.decoder_layer.value_storage = args.value_storage
    n_lanes = max(1, args.in_flight) if not args.no_graph else 1

    def sample_inputs(seed):
        """One synthetic nuScenes-shaped sample, resident in HBM: pyramid, the two BEV stacks and metas of its own."""
        return ([f.to(device) for f in syn.make_pyramid(cfg, seed)], syn.make_bev(cfg, seed, 0).to(device),
                syn.make_bev(cfg, seed, 1).to(device), syn.make_img_metas(cfg, sample=seed))

    # every rank decodes samples of its own, and every lane of a rank (samples in flight) a DIFFERENT one: sample index
    # rank * lanes + lane, as a sampler that deals consecutive samples to the lanes would (val.py:105-114,133-136 evaluates distinct
    # samples one after another); lane 0 of rank 0 is sample 0
    seed = rank * n_lanes
    pyramid, lss, radar, metas = sample_inputs(seed)
    if args.pregrouped:
        from racformer_amd.transformer import regroup_pyramid
        pyramid = regroup_pyramid(pyramid, cfg.num_cams, 4, fdt)
        head.transformer.decoder.pregrouped = True

    def eager_step():
        with torch.no_grad():
            fresh = [dict(m) for m in metas]                # new sample -> metas are staged (H2D) again
            preds = head(list(pyramid), lss, radar, fresh)
            det = head.get_detections_fixed(preds)          # [1,300,11]
            return dp.all_gather_detections(det, force_collective=args.force_collective)   # [world,1,300,11]

    captured, capture_error = None, None
    if not args.no_graph:
        # the whole step (regroup + prologue + 6 layers + decode) as ONE HIP-graph submission; per step the host stages the
        # sample's metas (timestamps -> time_diff, lidar2img: the float64 host arithmetic of racformer_transformer.py:99-109)
        # in front of the replay and issues the all-gather behind it
        from racformer_amd.graph import CapturedStep
        try:
            captured = CapturedStep(head, pyramid, lss, radar, metas)
        except Exception as e:       # (a box whose runtime refuses the capture still gets a measured, eager number -- and says so)
            capture_error = f"{type(e).__name__}: {e}"[:300]
            torch.cuda.synchronize()
    lanes, turn = [], [0]
    lane_inputs = [(pyramid, lss, radar, metas)]
    if captured is not None and args.in_flight > 1:
        lanes = [(captured, torch.cuda.Stream(device=device), metas)]
        for i in range(1, args.in_flight):
            if args.same_inputs_per_lane:        # (round 3's arrangement, kept as an A/B switch: every lane reads lane 0's buffers)
                lane_inputs.append(lane_inputs[0])
            else:
                lane_inputs.append(sample_inputs(seed + i))
            p_i, l_i, r_i, m_i = lane_inputs[-1]
            lanes.append((CapturedStep(head, p_i, l_i, r_i, m_i, own_scratch=True), torch.cuda.Stream(device=device), m_i))

    def step():
        if captured is None:
            return eager_step()
        if lanes:
            # the lane's stream carries the metas staging, the replay AND the all-gather of its step (the process group orders the
            # collectives among themselves; a lane never waits for another lane's replay)
            cap, st, m_i = lanes[turn[0] % len(lanes)]
            turn[0] += 1
            with torch.cuda.stream(st):
                _, det = cap.replay(img_metas=m_i)
                return dp.all_gather_detections(det, force_collective=args.force_collective)
        _, det = captured.replay(img_metas=metas)
        return dp.all_gather_detections(det, force_collective=args.force_collective)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Eager plan: inside the timed region only the dominant kernel is bracketed with HIP events (each pair costs a 5-10 us
    # bubble); the other instrumented launches are timed in three extra, untimed steps below.  Captured plan: a graph holds
    # no event brackets, so the timed region is K replays and the dominant kernel is bracketed in K eager steps of the same
    # work right behind it (same kernels, same inputs, same stream).
    if captured is None:
        _lib.timer = _lib.KernelTimer(only=("sampling4d_fwd", "mixing_sampled_fwd", "msmv_fwd"))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    host_elapsed = time.perf_counter() - t0     # the host has enqueued every step (it runs ahead of the GPU unless it is the limiter)
    fence()
    elapsed = time.perf_counter() - t0
    single = lanes_match = None
    if captured is not None:
        if not lanes:
            # --in-flight 1: the timed region itself is the one-plan figure (per rank)
            single = {"value": args.steps / elapsed, "unit": "samples/s", "ms_per_step": 1e3 * elapsed / args.steps,
                      "note": "the timed region itself (one captured plan in flight)"}
        if lanes:
            # every lane ran a sample of its own beside the others: what its last replay left must be, bit for bit, what ONE plan
            # alone produces on that lane's sample -- the eager step (one launch per kernel, nothing else on the GPU), computed here,
            # outside the timed region (a cross-plan race or a lane reading another lane's buffers would show here) ...
            lanes_match = []
            for (cap, _, m_i), (p_i, l_i, r_i, _) in zip(lanes, lane_inputs):
                with torch.no_grad():
                    alone = head.get_detections_fixed(head(list(p_i), l_i, r_i, [dict(m) for m in m_i]))
                torch.cuda.synchronize()
                lanes_match.append(bool(torch.equal(cap.det, alone)))
            # ... and the same K steps with ONE plan in flight: the latency-bound figure beside the throughput one
            t1 = time.perf_counter()
            for _ in range(args.steps):
                captured.replay(img_metas=metas)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            # GPU-side duration of ONE replay alone on the chip (first kernel's start to last kernel's end, HIP events on the replay's
            # stream, median of 20 single replays): what a sample's kernels take without another sample beside them
            evs = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                captured.replay(img_metas=metas)
                e1.record()
                torch.cuda.synchronize()
                evs.append(e0.elapsed_time(e1))
            single = {"value": args.steps / dt1, "unit": "samples/s", "ms_per_step": 1e3 * dt1 / args.steps,
                      "gpu_ms_per_replay_alone": statistics.median(evs),
                      "note": "one captured plan in flight (each sample's kernels run alone): per-sample latency"}
        _lib.timer = _lib.KernelTimer(only=("sampling4d_fwd", "mixing_sampled_fwd", "msmv_fwd"))
        for _ in range(args.steps):
            eager_step()
        fence()
    timer, _lib.timer = _lib.timer, _lib.KernelTimer()
    for _ in range(3):
        eager_step()
    fence()
    aux, _lib.timer = _lib.timer, None
    per_rank_ms = [1e3 * elapsed / args.steps]
    collective = None
    if use_pg:
        # the gathered block of this rank's slot is its local block, bit for bit
        with torch.no_grad():
            local = head.get_detections_fixed(head(list(pyramid), lss, radar, [dict(m) for m in metas]))
            gathered = dp.all_gather_detections(local, force_collective=args.force_collective)
        torch.cuda.synchronize()
        collective = {"backend": dist.get_backend(), "world": world, "forced_at_world_1": bool(args.force_collective and world == 1),
                      "gathered_shape": list(gathered.shape), "own_slot_equals_local": bool(torch.equal(gathered[rank], local))}
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [1e3 * float(t.item()) / args.steps for t in allr]
        elapsed = max(float(t.item()) for t in allr)      # max over ranks
        merged = dp.merge_interleaved(out, world)           # one sample per rank per step, dataset order
        assert tuple(merged.shape) == (world, 300, 11)
    # the dominant kernel: the stand-alone sampling kernel (rac_sampling4d_fwd); with decoder_layer.fuse_sampling_mixing = True
    # (RAC_FUSE_SAMPLING_MIXING=1: measured and rejected in round 4) the sampling runs inside the mixing kernel
    fused_sm = bool(timer.mean_ms("mixing_sampled_fwd"))
    msmv_ms = timer.mean_ms("mixing_sampled_fwd") or timer.mean_ms("sampling4d_fwd") or timer.mean_ms("msmv_fwd")
    bev_streams = 2 if aux.mean_ms("bev_sampling_x2_fwd") else 1        # radar + LSS in one launch
    msda_ms = aux.mean_ms("bev_sampling_x2_fwd") or aux.mean_ms("bev_sampling_fwd") or aux.mean_ms("msda_fwd")

    # algorithmic bytes of the msmv launches of one forward (untimed, instrumented pass)
    cap = _lib.KernelTimer()
    cap.capture_inputs, cap.captured = True, []
    _lib.timer = cap
    eager_step()
    torch.cuda.synchronize()
    _lib.timer = None
    elt = 4 if fdt == torch.float32 else 2
    S = cfg.batch * cfg.num_frames * cfg.num_groups
    P = cfg.num_points * cfg.img_depth_num
    out_elems = S * cfg.num_query * cfg.channels * P
    # fused kernel: the gather's bytes as SURVEY 8(d) counts them, WITHOUT the sampled-feature tensor (it is never written), plus what the
    # mixing half moves: the item's generated parameters (fp32) in, the out_proj operand image (f16 hi + lo = 4 bytes per value) out
    Pin_ = cfg.num_points * cfg.num_frames * cfg.img_depth_num
    mix_bytes = cfg.batch * cfg.num_query * cfg.num_groups * ((cfg.channels * cfg.channels + 128 * Pin_) * 4 + 128 * cfg.channels * 4)
    # The unit the roofline prices is ONE DECODER LAYER's sampling.  The layer may issue it as several launches over ranges of the
    # queries (the two-half pipeline of the decoder layer: the second half's gather runs beside the first half's mixing): their
    # locations are put together again and the SURVEY 8(d) formula -- whose per-level cap is "the level once" -- is applied to the
    # layer as a whole, against the SUM of the launches' durations.  (Applied per half-launch the caps of the coarse levels would count
    # twice and inflate the algorithmic bytes by a third.)
    launches_per_layer = max(1, len(cap.captured) // cfg.num_layers)
    assert len(cap.captured) == launches_per_layer * cfg.num_layers, (len(cap.captured), cfg.num_layers)
    layers_loc = [(torch.cat([cap.captured[i * launches_per_layer + j][0] for j in range(launches_per_layer)], dim=1),
                   cap.captured[i * launches_per_layer][1]) for i in range(cfg.num_layers)]
    per_launch = [msmv_algorithmic_bytes(loc, shapes, elt, 0 if fused_sm else out_elems) for loc, shapes in layers_loc]
    if fused_sm:
        per_launch = [(b_ + mix_bytes, f_) for b_, f_ in per_launch]
    b_alg = float(np.mean([b for b, _ in per_launch]))
    in_frac = [float(np.mean(f)) for _, f in per_launch]
    full_shapes = cap.captured[0][1]
    n_pts = S * cfg.num_query * P
    b_alg_closed = sum(min(n_pts * 4 * c * elt, s_ * n_ * h * w * c * elt) for (s_, n_, h, w, c) in full_shapes) \
        + n_pts * 3 * 4 + n_pts * cfg.num_levels * 4 + (mix_bytes if fused_sm else out_elems * 4)
    launch_ms = msmv_ms                                     # mean duration of ONE launch
    msmv_ms = msmv_ms * launches_per_layer if msmv_ms else None      # ... and of one layer's sampling
    achieved = b_alg / (msmv_ms * 1e-3) / 1e9 if msmv_ms else None
    traffic, traffic_src = pmc_traffic("mixing_c64_f16x3_kernel<4>", "racformer_amd/csrc/mixing.hip", args.config) if fused_sm else \
        pmc_traffic("sampling4d_c64_kernel", "racformer_amd/csrc/sampling_fused.hip", args.config)
    bev_traffic, _ = pmc_traffic("bev_sampling", "racformer_amd/csrc/bev_fused.hip", args.config)
    # MSDA algorithmic bytes of SURVEY 8(d) for one BEV launch (value stream read once + loc + weights + output)
    bev_h, bev_w = cfg.bev_hw
    bev_pts = cfg.batch * cfg.num_frames * cfg.num_query * 4 * cfg.num_points_bev * cfg.bev_depth_num
    velt = 2 if args.value_storage == "i16" else 4     # bytes per stored value of a BEV stream (+ one 4-byte scale per 64 with int16 blocks)
    vstream = cfg.batch * cfg.num_frames * bev_h * bev_w * (256 * velt + (16 if velt == 2 else 0))
    bev_alg = bev_streams * (min(bev_pts * 4 * 64 * velt, vstream) + bev_pts * 2 * 4
                             + bev_pts * 4 + cfg.batch * cfg.num_frames * cfg.num_query * 256 * 4)

    # Dense contractions of the path on the matrix cores.  The big ones run as split-precision products: operands
    # split into 16-bit terms (f16 hi/lo; bf16 x3 for the sampled features), 3 (6) cross products accumulated in
    # fp32 -- fp32-GEMM accuracy.  `executed` counts every MFMA product issued (incl. K padding) and is priced
    # against the dense 16-bit MFMA peak of gfx950 (2.5 PFLOP/s); `algorithmic` is the fp32 GEMM they replace.
    Qn, E, G_, C_ = cfg.num_query, cfg.embed_dims, cfg.num_groups, cfg.channels
    Pin = cfg.num_points * cfg.num_frames * cfg.img_depth_num
    gen_cols = G_ * (C_ * C_ + 128 * Pin)
    layer = head.transformer.decoder.decoder_layer
    split = bool(layer._pack_cache.get("split_packs", (None, {}))[1])
    PEAK16, PEAK32 = 2500.0, 157.3
    mfma = {}
    for key, name, alg, executed, peak in layer.mfma_report(cfg, split):
        ms = aux.mean_ms(key)
        if ms:
            tf = executed / (ms * 1e-3) / 1e12
            mfma[key] = {"kernel": name, "avg_launch_ms": ms, "gflop_algorithmic": alg / 1e9, "gflop_executed": executed / 1e9,
                         "achieved_tflops_executed": tf, "fp32_equivalent_tflops": alg / (ms * 1e-3) / 1e12,
                         "peak_tflops": PEAK16 if peak == 16 else PEAK32, "frac": tf / (PEAK16 if peak == 16 else PEAK32)}
    sasa_ms = aux.mean_ms("sasa_fwd")

    result = {
        "metric": "samples/sec (6-cam 704x256, 900 queries, f8)" if args.config == "f8"
                  else "samples/sec (3-cam 704x256, 900 queries, f8)",
        "value": world * args.steps / elapsed,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "per_rank_ms_per_step": per_rank_ms,
        # host-side time to enqueue one step (Python + launches, rank 0): close to ms_per_step means the host is the limiter
        "host_enqueue_ms_per_step": 1e3 * host_elapsed / args.steps,
        "submission": ("eager: one launch per kernel from Python" + (f" (graph capture failed: {capture_error})" if not args.no_graph and capture_error else ""))
                      if captured is None else
                      "HIP graph: the step's kernels captured once (racformer_amd/graph.py), one replay per step; metas staged in "
                      "front of it, all-gather behind it" + (f"; {len(lanes)} plans on streams of their own, replayed round-robin, so "
                      "that up to that many samples are in flight per GPU (the latency-bound launches of one sample run beside the "
                      "bandwidth-bound ones of another)" if lanes else ""),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "collective": collective,
        "one_sample_in_flight": single,
        # per lane: detections of the lane's last replay (run beside the other lanes) == the single-plan eager result on that lane's sample
        "lanes_match_single_plan_bitwise": lanes_match,
        "dtype": "f32" if fdt == torch.float32 else "bf16-features/f32-math", "data": "synthetic",
        "arithmetic_note": "fp32 arithmetic and fp32-accurate results throughout; inputs (pyramid, BEV maps), queries and every output in "
                           "fp32; the largest contractions run on the 16-bit matrix cores as split-precision products (operands = sums "
                           "of f16/bf16 terms, fp32 accumulate, truncation <= 2^-22 relative); "
                           + ("the two hoisted BEV value streams (internal tensors: value_proj's outputs) are stored as int16 mantissas "
                              "with one power-of-two scale per (pixel, head) block of 64 channels (opt-in mode, NOT the product default)"
                              if args.value_storage == "i16" else "fp32 storage everywhere (BEV value streams as the reference's fp32 maps)"),
        "config": {"workload": f"racformer_r50_nuimg_704x256_{args.config} query-decoder hot path: regroup + 6 decoder "
                               "layers + NMS-free decode, 1 sample/GPU/step",
                   "queries": cfg.num_query, "cams": cfg.num_cams, "frames": cfg.num_frames,
                   "levels": cfg.num_levels, "samples_per_gpu": 1, "samples_in_flight_per_gpu": max(1, args.in_flight if captured is not None else 1),
                   # each lane holds a sample of its own (pyramid, BEV stacks, metas: synthetic seeds rank * lanes + lane)
                   "distinct_inputs_per_lane": bool(lanes) and not args.same_inputs_per_lane,
                   "sample_seeds_this_rank": [seed + i for i in range(len(lanes) or 1)] if not args.same_inputs_per_lane else [seed],
                   "parallelism": f"dp{world}", "bev_value_stream_storage": "int16-block" if args.value_storage == "i16" else "f32",
                   # the figures a reader of the driver's record needs beside `value` (the driver keeps this dict whole):
                   # the reference's own evaluation semantics are one sample at a time (val.py:133-136)
                   "one_sample_in_flight_samples_per_s": single["value"] if single else None,
                   "one_sample_in_flight_ms_per_sample": single["ms_per_step"] if single else None,
                   "kernel_ms_per_sample_one_plan": single.get("gpu_ms_per_replay_alone") if single else None,
                   "lanes_match_single_plan_bitwise": lanes_match,
                   "pyramid_layout": "pregrouped [B*T*G,N,H,W,C]" if args.pregrouped else "reference [B,T*N,G*C,H,W] (regroup timed)"},
        "roofline": {"bound": "hbm", "kernel": "mixing_c64_f16x3_kernel<4> (rac_mixing_sampled_fwd: keypoints + projection + view select + gather "
                               "of the item's 96 points AND both adaptive mixings + LayerNorms, one launch per layer)" if fused_sm else
                               "sampling4d_c64_kernel (rac_sampling4d_fwd: keypoints + projection + view select + gather)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                     "traffic_note": traffic_src,
                     # real HBM-side bytes / time / peak: what the memory system itself sustained.  Below `frac` because L2 and
                     # the Infinity Cache absorb re-sampled pixels: the kernel is limited by its tap requests through L1/L2
                     # (gather request rate), not by HBM bandwidth.
                     "hbm_frac_measured": (traffic / (msmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and msmv_ms else None,
                     "limiter": "the CU's texture path (1.41 GB of tap requests through L1 / L2) beside the 0.35 GB parameter / output stream of the "
                                "mixing half" if fused_sm else "L1/L2 gather-request rate (HBM-side traffic is below the algorithmic bytes)",
                     "sampling_inside_mixing_kernel": fused_sm,
                     "avg_launch_ms": launch_ms, "launches_per_step": len(cap.captured), "launches_per_layer": launches_per_layer,
                     "avg_layer_ms": msmv_ms,
                     "roofline_unit_note": "one decoder layer's sampling (all its launches: algorithmic bytes of the layer / sum of their durations)",
                     "timed_in": "HIP events on the launch stream inside the timed region" if captured is None else
                                 f"HIP events on the launch stream over {args.steps} eager steps of the same work right behind the "
                                 "timed region (a captured graph holds no event brackets)",
                     "algorithmic_bytes_per_launch": b_alg, "algorithmic_bytes_all_in_range": b_alg_closed,
                     "in_range_fraction_per_layer": in_frac,
                     "bev_sampling": {"avg_launch_ms": msda_ms, "streams_per_launch": bev_streams, "algorithmic_bytes_per_launch": bev_alg,
                                      "achieved": bev_alg / (msda_ms * 1e-3) / 1e9 if msda_ms else None,
                                      "frac": bev_alg / (msda_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if msda_ms else None,
                                      "traffic": bev_traffic},
                     "sasa_avg_launch_ms": sasa_ms},
        "mfma": mfma,
    }
    if world > 1:
        # the CPU oracle and the operator stress set are timed on rank 0 of a ONE-rank run only (they would sit inside the other
        # ranks' barrier otherwise); the line says so instead of omitting the keys
        result["cpu_baseline"] = None
        result["cpu_baseline_note"] = "measured at --gpus 1 only (rank 0 of a single-rank run); see the N = 1 line of the same build"
        result["roofline_stress"] = None
    if rank == 0 and world == 1 and not args.no_stress:
        result["roofline_stress"] = stress_block(cfg, device)

    if rank == 0 and world == 1 and not args.pregrouped and captured is not None:
        # SECOND line, never the headline (row f2): the same step when the producer hands over the pyramid already in the
        # sampling layout [B*T*G, N, H, W, C] (racformer_amd/fpn_writer.py writes it from the neck's last convolution), i.e.
        # without the 1.47 GB regroup of racformer_transformer.py:112-124
        from racformer_amd.graph import CapturedStep
        from racformer_amd.transformer import regroup_pyramid
        dec = head.transformer.decoder
        grouped = regroup_pyramid(pyramid, cfg.num_cams, 4, fdt)
        dec.pregrouped = True
        try:
            cap2 = CapturedStep(head, grouped, lss, radar, metas)
            for _ in range(args.warmup):
                cap2.replay(img_metas=metas)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                cap2.replay(img_metas=metas)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            result["pregrouped_producer_layout"] = {
                "value": args.steps / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / args.steps,
                "note": "secondary figure: pyramid handed over as [B*T*G,N,H,W,C] (producer-side layout, SURVEY 8 row f2), no "
                        "regroup in the step; the headline `value` times the reference layout WITH the regroup"}
            cap2.close()
            del cap2
        finally:
            dec.pregrouped = False

    if rank == 0 and world == 1 and not args.pregrouped and captured is not None and args.value_storage == "f32":
        # THIRD line, never the headline: the same step, the same lanes and samples, with the two hoisted BEV value streams in the
        # opt-in int16 block storage (decoder_layer.value_storage = "i16": csrc/quant.hip; written by their producers' own epilogues,
        # rac_conv3x3_q16_fwd / rac_value_proj_q16_fwd, read by rac_bev_sampling_multi_q16_fwd) -- half the bytes through the texture
        # path that bounds the BEV kernel.  fp32 arithmetic, fp32 pyramid.  Parity of this mode: both reference-initialised
        # rigs literal, the f8 random rig inside the fp32 path's tail budget, the reduced head fixture NOT literal (one query at 1.1e-3)
        # -- tests/test_lowprec_storage_gpu.py, DESIGN 3.11 -- which is why it is not the default.
        from racformer_amd.graph import CapturedStep
        layer.value_storage = "i16"
        try:
            caps = [(CapturedStep(head, p_i, l_i, r_i, m_i, own_scratch=True), torch.cuda.Stream(device=device), m_i)
                    for (p_i, l_i, r_i, m_i) in lane_inputs]

            def run(k, only_first=False):
                use = caps[:1] if only_first else caps
                for i in range(k):
                    cap_i, st_i, m_i = use[i % len(use)]
                    with torch.cuda.stream(st_i):
                        cap_i.replay(img_metas=m_i)
            run(args.warmup * len(caps))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            t1 = time.perf_counter()
            run(args.steps, only_first=True)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            for cap_i, _, _ in caps:
                cap_i.close()
            del caps
            _lib.timer = _lib.KernelTimer(only=("bev_sampling_x2_fwd", "value_proj_fwd", "temporal_fusion_conv"))
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            kt, _lib.timer = _lib.timer, None
            result["bev_values_int16_block"] = {
                "value": args.steps / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / args.steps,
                "samples_in_flight_per_gpu": len(lane_inputs),
                "one_sample_in_flight": {"value": args.steps / dt1, "ms_per_step": 1e3 * dt1 / args.steps},
                "bev_sampling_avg_launch_ms": kt.mean_ms("bev_sampling_x2_fwd"), "bev_sampling_fp32_avg_launch_ms": msda_ms,
                "value_proj_avg_launch_ms": kt.mean_ms("value_proj_fwd"), "value_proj_fp32_avg_launch_ms": aux.mean_ms("value_proj_fwd"),
                "temporal_fusion_conv_avg_launch_ms": kt.mean_ms("temporal_fusion_conv"),
                "quantiser_launches_per_step": 0,
                "note": "secondary figure (same lanes and samples as the headline; compare value with value, one_sample_in_flight with "
                        "one_sample_in_flight): BEV value streams stored as int16 mantissas + one power-of-two scale per (pixel, head) "
                        "block of 64 channels, written by the producers' own epilogues (no quantiser launch); fp32 arithmetic; the "
                        "headline `value` keeps fp32 value streams"}
        finally:
            layer.value_storage = args.value_storage

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import restate as R  # the checker, timed as the reported CPU baseline only
        from parity import flipped_points, head_boxes_normalised
        torch.set_num_threads(host_threads())
        sd = {k: v.detach().cpu() for k, v in head.transformer.state_dict().items()}
        hsd = {k: v.detach().cpu() for k, v in head.state_dict().items() if not k.startswith("transformer.")}
        cpu_pyr = [f.cpu() for f in pyramid]
        warm, reps = 3, 5      # ~35 s of CPU work on the 16-core share (bounded sample; SURVEY 8d plans 3 warm-ups + 10 in the container)
        times = []
        R.LOC_TAP = []
        with torch.no_grad():
            for i in range(warm + reps):
                if i == warm + reps - 1:
                    R.LOC_TAP = []
                c0 = time.perf_counter()
                ref = R.head_forward(hsd, sd, cpu_pyr, lss.cpu(), radar.cpu(), syn.make_img_metas(cfg, sample=seed), cfg)
                times.append(time.perf_counter() - c0)
        oviews = torch.stack([R.views_of(l, cfg.num_cams) for l in R.LOC_TAP])
        R.LOC_TAP = None
        cpu_s = statistics.median(times[warm:])
        # parity of the benchmarked step against the oracle, reported beside the numbers (normalised box space, flips shown)
        taps = layer.sampling.capture_loc = []
        with torch.no_grad():
            preds = head(list(pyramid), lss, radar, [dict(m) for m in metas])
        torch.cuda.synchronize()
        layer.sampling.capture_loc = None
        views = torch.stack([R.views_of(l.cpu(), cfg.num_cams) for l in taps])
        got_n = head_boxes_normalised(preds["all_bbox_preds"].cpu(), cfg.pc_range)
        ref_n = head_boxes_normalised(ref["all_bbox_preds"], cfg.pc_range)
        nflips = flipped_points(views, oviews)
        eb = (got_n - ref_n).abs().amax(-1)
        mism = preds["all_cls_scores"].cpu().argmax(-1) != ref["all_cls_scores"].argmax(-1)
        result["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"{warm} warm-up + median of {reps} forwards of the same synthetic sample (regroup + 6 decoder layers + "
                      "head) through oracle/restate.py (torch-CPU + OpenMP C gathers)",
            "seconds": cpu_s, "seconds_all": times}
        result["parity_vs_oracle"] = {
            "space": "decoder-normalised boxes (xyz / pc_range span, log sizes, sin, cos, v)",
            "box_abs_err_median": float(eb.median()), "box_abs_err_max": float(eb.max()),
            "queries_over_1e-3_per_layer": (eb > 1e-3).flatten(1).sum(1).tolist(),
            "argmax_mismatches_per_layer": mism.flatten(1).sum(1).tolist(),
            "differing_camera_choices_per_layer": nflips,
            "note": "free-running six layers with each side's own camera choices (tests/parity.py explains the criterion the "
                    "tests apply: differing choices equalised, per-layer tail budget)"}

    bad_lanes = lanes_match is not None and not all(lanes_match)
    if bad_lanes:
        # a lane that does not reproduce the single-plan result is a wrong result, not a throughput: no headline number
        result["invalid"] = f"lanes_match_single_plan_bitwise = {lanes_match}: value withheld (was {result['value']:.2f})"
        result["value"] = None
    if rank == 0:
        print(json.dumps(result))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    if bad_lanes:
        sys.exit(3)


if __name__ == "__main__":
    main()
