#!/usr/bin/env python3
"""The clock the three matrix-core kernels hold under their own load (diagnostic; GPU box).

usage:  tools/build_variant.sh clocks "-DRAC_CLOCK_STAMPS" conv3x3.hip gemm_split.hip          (container)
        RACFORMER_HIP_LIB=build/lib_clocks.so python3 tools/kernel_clock.py [out.json]         (GPU box)

The diagnostic build stamps s_memtime (shader-clock ticks) and s_memrealtime (100 MHz) around the main loop of every workgroup of
conv3x3_f16x3_kernel, generator_ws_kernel and gemm_split_kernel (rac_common.h, RAC_CLOCK_*).  In-kernel clock =
d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups (MI355X_MICROARCH.md, 'DVFS give-back' item 6).  Two settings
per kernel: (a) back to back -- the kernel alone, launched for >= 2 s on random operands of the f8 shapes, stamps of the last
launch; (b) in situ -- >= 2 s of eager forwards of the whole decoder step, stamps of the step's last launch of that kernel.
Beside the clock: the loop's duration and the executed MFMA rate it implies, against the matrix pipes' rate AT THAT CLOCK
(1024 flop / clock / SIMD for the 16-bit MFMAs, 256 CUs x 4 SIMDs)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib, synthetic as syn  # noqa: E402
from racformer_amd.fused import ConvImage, generator_fused, outproj_fused, pack_conv3x3_weight, pack_gemm_split_weight  # noqa: E402
import bench  # noqa: E402

DEV = torch.device("cuda", 0)
WGS = 1024


def read(name, n):
    fn = getattr(_lib.lib(), "rac_dbg_clock_" + name, None)
    if fn is None:
        raise SystemExit("kernel_clock: this library has no rac_dbg_clock_* -- build the -DRAC_CLOCK_STAMPS variant and set RACFORMER_HIP_LIB")
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((n, 4), dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p), n)
    if rc != 0:
        raise SystemExit(f"rac_dbg_clock_{name} rc={rc}")
    t = buf.astype(np.int64)
    ok = (t[:, 3] > t[:, 1]) & (t[:, 2] > t[:, 0])
    t = t[ok]
    ghz = (t[:, 2] - t[:, 0]) / (t[:, 3] - t[:, 1]) * 0.1
    us = (t[:, 3] - t[:, 1]) / 100.0
    span_us = (t[:, 3].max() - t[:, 1].min()) / 100.0
    return {"workgroups": int(len(t)), "clock_ghz_median": float(np.median(ghz)), "clock_ghz_p10": float(np.percentile(ghz, 10)),
            "clock_ghz_p90": float(np.percentile(ghz, 90)), "loop_us_median": float(np.median(us)), "launch_span_us": float(span_us)}


def rate(rec, executed_flop, loops_share=1.0):
    """executed MFMA rate over the launch's span, and the pipes' rate at the measured clock"""
    tf = executed_flop / (rec["launch_span_us"] * 1e-6) / 1e12
    peak_at_clock = 1024 * 4 * 256 * rec["clock_ghz_median"] * 1e9 / 1e12
    rec.update(executed_tflops_over_span=tf, pipe_tflops_at_measured_clock=peak_at_clock, frac_of_pipe_at_clock=tf / peak_at_clock)
    return rec


def spin(fn, seconds):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return n


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "kernel_clock.json")
    g = torch.Generator().manual_seed(3)
    res = {"note": __doc__.split("\n\n")[1].replace("\n", " ")}
    # ---- (a) back to back, f8 shapes, random operands
    x = torch.randn(8, 320, 128, 128, generator=g).to(DEV)
    w = (torch.randn(256, 320, 3, 3, generator=g) * 0.02).to(DEV)
    ws, alpha = pack_conv3x3_weight(w)
    img = ConvImage(8, 128, 128, 320, DEV).begin([x]).pack(x, 0)
    bias = torch.randn(256, generator=g).to(DEV)
    n = spin(lambda: img.conv(ws, alpha, bias), 2.5)
    res["conv3x3_back_to_back"] = rate(read("conv3x3", 512), 3 * 2.0 * 8 * 128 * 128 * 256 * 320 * 9)
    res["conv3x3_back_to_back"]["launches"] = n
    del x, img
    xg, _ = pack_gemm_split_weight(torch.randn(900, 256, generator=g).to(DEV))
    wg, ag = pack_gemm_split_weight((torch.randn(65536, 256, generator=g) * 0.05).to(DEV))
    bg = torch.randn(65536, generator=g).to(DEV)
    n = spin(lambda: generator_fused(xg.view(900, -1), wg, bg, ag, timer_name=None), 2.5)
    res["generator_back_to_back"] = rate(read("generator", 256), 3 * 2.0 * 928 * 256 * 65536)
    res["generator_back_to_back"]["launches"] = n
    zo, _ = pack_gemm_split_weight(torch.randn(900, 32768, generator=g).to(DEV))
    wo, _ = pack_gemm_split_weight((torch.randn(256, 32768, generator=g) * 0.05).to(DEV))
    n = spin(lambda: outproj_fused(zo, wo, 32), 2.5)
    res["outproj_back_to_back"] = rate(read("outproj", 256), 3 * 2.0 * 900 * 32768 * 256)
    res["outproj_back_to_back"]["launches"] = n
    del xg, wg, zo, wo
    torch.cuda.empty_cache()
    # ---- (b) in situ: eager forwards of the whole step
    cfg = syn.F8
    head = bench.build_head(cfg, DEV)
    pyramid = [f.to(DEV) for f in syn.make_pyramid(cfg, 0)]
    lss, radar = syn.make_bev(cfg, 0, 0).to(DEV), syn.make_bev(cfg, 0, 1).to(DEV)
    metas = syn.make_img_metas(cfg)

    def fwd():
        with torch.no_grad():
            head(list(pyramid), lss, radar, [dict(m) for m in metas])
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.5:
        fwd()
        n += 1
    torch.cuda.synchronize()
    res["in_situ_forwards"] = n
    res["conv3x3_in_situ"] = rate(read("conv3x3", 512), 3 * 2.0 * 8 * 128 * 128 * 256 * 320 * 9)
    res["generator_in_situ_last_launch"] = read("generator", 256)        # (the step's last generator launch is the 2189-wide one)
    res["outproj_in_situ"] = rate(read("outproj", 256), 3 * 2.0 * 900 * 32768 * 256)
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res.items():
        if isinstance(v, dict):
            print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
