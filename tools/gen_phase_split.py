#!/usr/bin/env python3
"""Where a stage of generator_ws_kernel<8, 32> spends its time (diagnostic; GPU box).

usage:  tools/build_variant.sh gw_stamps "-DGW_STAMPS" gemm_split.hip                          (container)
        RACFORMER_HIP_LIB=build/lib_gw_stamps.so python3 tools/gen_phase_split.py [out.json]   (GPU box)

Stamps (s_memtime; waves 0 and 5 of every workgroup of the wide launch): stage top | after the MFMA block | after the epilogue's stores |
after the counted wait for the next stage's pieces | after the barrier; s_memrealtime at the end of the stage calibrates the tick."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib  # noqa: E402
from racformer_amd.fused import SPLIT_ACT_SCALE, generator_fused, pack_gemm_split_weight, row_gemm, row_seg, rowgemm_launch  # noqa: E402

PH = ["fragment reads + 192 MFMAs per SIMD", "epilogue: alpha / bias FMAs + stores", "counted wait for the next stage's pieces", "barrier"]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    M, N, K = 900, 65536, 256
    w_img, alpha = pack_gemm_split_weight((torch.randn(N, K, generator=g) * 0.05).to(dev))
    x = torch.randn(M, K, generator=g).to(dev)
    # the X line image as the decoder produces it: finished rows of a rowgemm prologue (plain copy of x) beside a throw-away 16-column GEMM
    x_img = torch.empty(M, 512, device=dev, dtype=torch.float16)
    dummy_w, dummy_out = torch.zeros(16, 256, device=dev), torch.empty(M, 16, device=dev)
    rowgemm_launch([row_gemm([row_seg(x, split_out=x_img, split_lines=True)], dummy_w, None, dummy_out)], M)
    x_alpha = 1.0 / SPLIT_ACT_SCALE
    bias = torch.randn(N, generator=g).to(dev)
    fn = getattr(_lib.lib(), "rac_dbg_gw_stamps", None)
    if fn is None:
        raise SystemExit("gen_phase_split: build the -DGW_STAMPS variant and set RACFORMER_HIP_LIB")
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    junk = torch.empty(300 * 1024 * 1024 // 4, device=dev)
    for _ in range(5):
        junk.fill_(1.0)
        generator_fused(x_img, w_img, bias, alpha)
    torch.cuda.synchronize()
    buf = np.zeros((256, 2, 32, 6), dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise SystemExit(f"rac_dbg_gw_stamps rc={rc}")
    t = buf.astype(np.int64)
    n = int((t[0, 0, :, 0] != 0).sum())
    t = t[:, :, :n]
    ghz = float(np.median((t[:, 0, n - 1, 4] - t[:, 0, 0, 4]) / ((t[:, 0, n - 1, 5] - t[:, 0, 0, 5]) * 10.0)))
    out = {"stages": n, "shader_clock_ghz": ghz, "waves": {}}
    for wi, name in enumerate(("wave0", "wave5")):
        d = np.diff(t[:, wi, 1:, :5], axis=-1).reshape(-1, 4)
        stage = (t[:, wi, 2:, 0] - t[:, wi, 1:-1, 0]).reshape(-1)
        out["waves"][name] = {"stage_ns_median": float(np.median(stage) / ghz),
                              "phases_ns_median": {PH[i]: float(np.median(d[:, i]) / ghz) for i in range(4)},
                              "phases_ns_p90": {PH[i]: float(np.percentile(d[:, i], 90) / ghz) for i in range(4)}}
    out["first_stage_to_last_us"] = float((t[:, :, n - 1, 5].max() - t[:, :, 0, 5].min()) * 10.0 / 1e3)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
