#!/usr/bin/env python3
"""Where rac_layer_tail_fwd's time goes, phase by phase (diagnostic; GPU box).
usage:  tools/build_variant.sh ltstamps "-DLT_STAMPS" layer_tail.hip                       (container)
        RACFORMER_HIP_LIB=build/lib_ltstamps.so python3 tools/layer_tail_phases.py [out.json]   (GPU box)
Wave 0 of every workgroup stamps s_memtime (100 MHz on gfx950: 10 ns ticks) at the phase boundaries; one launch at 900 rows."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racformer_amd import _lib, synthetic as syn  # noqa: E402
from racformer_amd.fused import layer_tail_fused  # noqa: E402
from racformer_amd.transformer import RaCFormerTransformer  # noqa: E402

NAMES = ["0 load bev", "1 bev out_proj x2", "2 LN radar/lss", "3 fusion K=768", "4 LN fusion", "5 FFN1 256->512", "6 FFN2 K=512",
         "7 LN3", "8 c0r0 256->512", "9 LN c1", "10 cls3 || reg2", "11 LN c4", "12 cls6 || reg4"]


def main():
    h = _lib.lib()
    fn = getattr(h, "rac_dbg_layer_tail_stamps", None)
    if fn is None:
        raise SystemExit("build the -DLT_STAMPS variant and set RACFORMER_HIP_LIB")
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    dev = "cuda:0"
    tr = RaCFormerTransformer(**syn.F8.transformer_kwargs()).eval()
    syn.fill_params(tr, 3)
    lg = tr.to(dev).decoder.decoder_layer
    rows = 900
    bev, x1, x2 = torch.randn(2, rows, 256, device=dev), torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
    c0, r0 = lg.cls_branch[0], lg.reg_branch[0]
    w, b = torch.cat([c0.weight, r0.weight], 0).contiguous(), torch.cat([c0.bias, r0.bias], 0).contiguous()
    with torch.no_grad():
        for _ in range(5):
            layer_tail_fused(bev, x1, x2, lg, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            layer_tail_fused(bev, x1, x2, lg, w, b)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    buf = np.zeros((64, 16), dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf[:57, :14].astype(np.int64)
    d = np.diff(t, axis=1)
    life = t[:, 13] - t[:, 0]
    tick_us = 0.01
    out = {"launch_us_back_to_back": us, "workgroup_life_us_mean": float(life.mean() * tick_us), "phases_us_mean": {}}
    print(f"launch {us:.1f} us back to back; workgroup life mean {life.mean() * tick_us:.1f} us (min {life.min() * tick_us:.1f}, max {life.max() * tick_us:.1f})")
    for i, n in enumerate(NAMES):
        out["phases_us_mean"][n] = float(d[:, i].mean() * tick_us)
        print(f"  {n:22s} {d[:, i].mean() * tick_us:7.2f} us  (p90 {np.percentile(d[:, i], 90) * tick_us:.2f})")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
