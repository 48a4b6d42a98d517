#!/usr/bin/env python3
"""Where the BEV kernel's time goes, phase by phase (diagnostic; GPU box).

usage:  tools/build_variant.sh bevstamps "-DBEV_STAMPS" bev_fused.hip          (container)
        RACFORMER_HIP_LIB=build/lib_bevstamps.so python3 tools/bev_phase_split.py [out.json]     (GPU box)

The diagnostic build of bev_fused.hip stamps s_memtime at the phase boundaries of every workgroup (wave 0): start | A: box
table, ray offsets, the two softmaxes | B: 640 keypoints -> tap lists | C: gather | D: LDS sum + store.  One eager forward of
the f8 rig runs; the stamps of its LAST bev_sampling launch (layer 5, radar + LSS streams, 1800 workgroups) are read back
through rac_dbg_bev_stamps and summarised: per-phase ticks (mean / p50 / p90), the share of the workgroup's lifetime, how
the workgroups' start times fall into rounds, and the tick rate calibrated against the launch's HIP-event time."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib, synthetic as syn  # noqa: E402
import bench  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "bev_phase_split.json")
    dev = torch.device("cuda", 0)
    h = _lib.lib()
    fn = getattr(h, "rac_dbg_bev_stamps", None)
    if fn is None:
        raise SystemExit("bev_phase_split: this library has no rac_dbg_bev_stamps -- build the -DBEV_STAMPS variant and set RACFORMER_HIP_LIB")
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    cfg = syn.F8
    head = bench.build_head(cfg, dev)
    pyramid = [f.to(dev) for f in syn.make_pyramid(cfg, 0)]
    lss, radar = syn.make_bev(cfg, 0, 0).to(dev), syn.make_bev(cfg, 0, 1).to(dev)
    metas = syn.make_img_metas(cfg)
    _lib.timer = _lib.KernelTimer(only=("bev_sampling_x2_fwd",))
    with torch.no_grad():
        for _ in range(3):
            head(list(pyramid), lss, radar, [dict(m) for m in metas])
    torch.cuda.synchronize()
    ev_ms = _lib.timer.mean_ms("bev_sampling_x2_fwd")
    _lib.timer = None
    n = cfg.num_query * 2
    buf = np.zeros((n, 8), dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p), n)
    if rc != 0:
        raise SystemExit(f"rac_dbg_bev_stamps rc={rc}")
    t = buf[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    merged = bool((t[:, 6] > 0).all())      # the tap-merging variant stamps the end of phase B2 (slot 6; slot 5 is the placement word)
    t[:, [0, 1, 2, 3, 4, 6]] -= t0
    span = int(t[:, 4].max())
    if merged:
        phases = {"A_prologue": t[:, 1] - t[:, 0], "B_keypoints_records": t[:, 2] - t[:, 1], "B2_merge_taps": t[:, 6] - t[:, 2],
                  "C_gather": t[:, 3] - t[:, 6], "D_sum_store": t[:, 4] - t[:, 3]}
    else:
        phases = {"A_prologue": t[:, 1] - t[:, 0], "B_keypoints_taplists": t[:, 2] - t[:, 1], "C_gather": t[:, 3] - t[:, 2],
                  "D_sum_store": t[:, 4] - t[:, 3]}
    life = t[:, 4] - t[:, 0]
    hw = buf[:, 5]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    cu = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 8) & 0xF
    se = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 13) & 0x7
    starts = np.sort(t[:, 0])
    res = {
        "kernel": "bev_sampling_d64_kernel<float>, last launch of an eager f8 forward (layer 5; radar + LSS, 1800 workgroups of 256 threads)",
        "hip_event_ms_per_launch_stamped_build": ev_ms,
        "ticks_first_start_to_last_end": span,
        "ticks_per_us_if_span_equals_event_time": span / (ev_ms * 1e3) if ev_ms else None,
        "workgroup_lifetime_ticks": {"mean": float(life.mean()), "p50": float(np.median(life)), "p90": float(np.percentile(life, 90))},
        "phases_ticks": {k: {"mean": float(v.mean()), "p50": float(np.median(v)), "p90": float(np.percentile(v, 90)),
                             "share_of_lifetime": float(v.sum() / life.sum())} for k, v in phases.items()},
        "first_round": {"workgroups_started_within_5pct_of_span": int((starts < 0.05 * span).sum()),
                        "start_tick_percentiles_of_span": {str(p): float(np.percentile(starts, p) / span) for p in (10, 25, 50, 57, 75, 90, 99)}},
        "placement": {"distinct_xcc": int(len(set(xcc.tolist()))), "distinct_(xcc,se,cu)": int(len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))},
        "note": "s_memtime ticks (wave 0 of each workgroup); a stamped build runs a few % slower than the product kernel",
    }
    # first-round workgroups only (those that started with the launch): their phases are what a lock-step launch sees
    first = t[:, 0] < 0.05 * span
    res["phases_ticks_first_round"] = {k: float(v[first].mean()) for k, v in phases.items()}
    res["phases_ticks_later_rounds"] = {k: float(v[~first].mean()) for k, v in phases.items()} if (~first).any() else None
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
