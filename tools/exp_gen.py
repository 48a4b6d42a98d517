#!/usr/bin/env python3
"""A/B timing of the split-precision GEMM kernels (generator / out_proj) across library builds, one child process per
build (RACFORMER_HIP_LIB), interleaved rounds, HIP-event medians.  usage: python tools/exp_gen.py lib1.so lib2.so ..."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("EXP_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    from racformer_amd.fused import generator_fused, outproj_fused
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    x_img = (torch.randn(900, 512, generator=g) * 100).to(torch.float16).to(dev)
    w_img = (torch.randn(65536, 8, 64, generator=g) * 100).to(torch.float16).to(dev)
    bias = torch.randn(65536, generator=g).to(dev)
    z_img = (torch.randn(900, 1024, 64, generator=g) * 100).to(torch.float16).to(dev)
    wo_img = (torch.randn(256, 1024, 64, generator=g) * 100).to(torch.float16).to(dev)
    spoil = torch.empty(512 * 1024 * 1024 // 4, device=dev)

    def timed(fn, reps=15):
        out = []
        for _ in range(reps):
            spoil.fill_(1.0)                       # cold caches, as inside the layer
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) * 1e3)
        return statistics.median(out), min(out)
    for _ in range(3):
        generator_fused(x_img, w_img, bias, 1e-6)
        outproj_fused(z_img, wo_img, 32)
    print("generator us (median, min): %.1f %.1f | outproj: %.1f %.1f" % (timed(lambda: generator_fused(x_img, w_img, bias, 1e-6))
                                                                          + timed(lambda: outproj_fused(z_img, wo_img, 32))), flush=True)
    sys.exit(0)

libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, EXP_CHILD="1", RACFORMER_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(f"round {rnd} {os.path.basename(lib):28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
