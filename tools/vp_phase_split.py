#!/usr/bin/env python3
"""Where a stage of value_proj_kernel spends its time (diagnostic; GPU box).

usage:  tools/build_variant.sh vp_stamps "-DVP_STAMPS" value_proj.hip                          (container)
        RACFORMER_HIP_LIB=build/lib_vp_stamps.so python3 tools/vp_phase_split.py [out.json]    (GPU box)

The diagnostic build stamps s_memtime at nine points of every stage (waves 0 and 5 of every workgroup; value_proj.hip VP_STAMP)
and s_memrealtime (100 MHz) at the end of each stage.  Printed: the median length of every phase over workgroups and stages
(first stage of a workgroup left out), in shader-clock ticks and -- calibrated by the two clocks -- in nanoseconds."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib  # noqa: E402
from racformer_amd.fused import SPLIT_ACT_SCALE, pack_gemm_split_weight, value_proj_fused  # noqa: E402

PHASES = ["top wait (pieces of this stage)", "raw reads + in-wave maxima", "barrier A", "maxima, scale, hi/lo image stores", "barrier B",
          "additive loads + piece issue", "image reads + MFMA", "epilogue + stores"]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 256, 128, 128, generator=g).to(dev)
    add = torch.randn(128 * 128, 256, generator=g).to(dev)
    w_img, alpha = pack_gemm_split_weight((torch.randn(256, 256, generator=g) * 0.06).to(dev))
    junk = torch.empty(300 * 1024 * 1024 // 4, device=dev)
    fn = getattr(_lib.lib(), "rac_dbg_vp_stamps", None)
    if fn is None:
        raise SystemExit("vp_phase_split: this library has no rac_dbg_vp_stamps -- build the -DVP_STAMPS variant and set RACFORMER_HIP_LIB")
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    out = {}
    for q16 in (False, True):
        for _ in range(5):
            junk.fill_(1.0)
            value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, add=add, q16=q16)
        torch.cuda.synchronize()
        buf = np.zeros((256, 2, 32, 10), dtype=np.uint64)
        rc = fn(buf.ctypes.data_as(ctypes.c_void_p), 256)
        if rc != 0:
            raise SystemExit(f"rac_dbg_vp_stamps rc={rc}")
        t = buf.astype(np.int64)
        n = int((t[0, 0, :, 0] != 0).sum())
        t = t[:, :, :n]
        # ticks per ns from the two clocks over the whole workgroup
        tick = (t[:, 0, n - 1, 8] - t[:, 0, 0, 8]) / ((t[:, 0, n - 1, 9] - t[:, 0, 0, 9]) * 10.0)
        ghz = float(np.median(tick))
        rec = {"stages_per_workgroup": n, "shader_clock_ghz": ghz, "waves": {}}
        for wi, wname in enumerate(("wave0", "wave5")):
            d = np.diff(t[:, wi, 1:, :9], axis=-1).reshape(-1, 8)            # phases of stages 1..n-1
            stage = (t[:, wi, 2:, 0] - t[:, wi, 1:-1, 0]).reshape(-1)
            rec["waves"][wname] = {"stage_ns_median": float(np.median(stage) / ghz),
                                   "phases_ns_median": {PHASES[i]: float(np.median(d[:, i]) / ghz) for i in range(8)},
                                   "phases_ns_p90": {PHASES[i]: float(np.percentile(d[:, i], 90) / ghz) for i in range(8)}}
        span = (t[:, :, n - 1, 9].max() - t[:, :, 0, 9].min()) * 10.0
        rec["first_stage_end_to_last_stage_end_us"] = float(span / 1e3)
        out["q16" if q16 else "fp32"] = rec
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
