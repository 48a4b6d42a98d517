"""Host side of round 5's radar-encoder kernels, on the CPU: (1) the algebra of the temporal fold -- the per-pixel map that stands in
for the constant hidden channels of the frames past the ConvGRU's live ones (racformer_amd.transformer.dead_frame_bias_map,
models/racformer_transformer.py:657-693) against the convolution evaluated explicitly in float64, border included; (2) the ctypes
mirror of `rac_conv_direct` / `rac_cd_frames` / `rac_cd_scale` against the C structs of include/racformer_hip.h as gcc lays them out
(sizes and the offset of every field: a descriptor struct crosses the C-ABI by pointer, a shifted field would be silent)."""
import ctypes
import os
import re
import subprocess
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_dead_frame_bias_map_is_the_convolution_of_the_constant():
    from racformer_amd.transformer import dead_frame_bias_map
    g = torch.Generator().manual_seed(11)
    Cx, Ch, Cout, H, W = 8, 6, 5, 7, 9
    w = torch.randn(Cout, Cx + Ch, 3, 3, generator=g, dtype=torch.float64)
    b_up = torch.randn(Ch, generator=g, dtype=torch.float64)
    x = torch.randn(1, Cx, H, W, generator=g, dtype=torch.float64)
    full = torch.cat([x, b_up.view(1, Ch, 1, 1).expand(1, Ch, H, W)], dim=1)
    want = F.conv2d(full, w, None, padding=1) - F.conv2d(x, w[:, :Cx], None, padding=1)          # [1, Cout, H, W]
    got = dead_frame_bias_map(w[:, Cx:], b_up, H, W)                                              # [H*W, Cout]
    assert tuple(got.shape) == (H * W, Cout) and got.dtype == torch.float64
    assert (got.t().reshape(Cout, H, W) - want[0]).abs().max().item() < 1e-12
    # interior pixels see all nine taps, a corner four, an edge six: the map is not a constant
    full_sum = (w[:, Cx:] * b_up.view(1, Ch, 1, 1)).sum(dim=(1, 2, 3))
    assert torch.allclose(got[(H // 2) * W + W // 2], full_sum, atol=1e-12)
    corner = (w[:, Cx:, 1:, 1:] * b_up.view(1, Ch, 1, 1)).sum(dim=(1, 2, 3))                      # pixel (0, 0): taps (dy, dx) >= 0 only
    assert torch.allclose(got[0], corner, atol=1e-12) and not torch.allclose(got[0], full_sum, atol=1e-6)


def _c_layout(struct, fields):
    """sizeof + offsetof of every field of a struct of include/racformer_hip.h, from a program compiled with gcc"""
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "racformer_hip.h"', "int main(void) {",
            f'  printf("%zu\\n", sizeof({struct}));']
    prog += [f'  printf("%zu\\n", offsetof({struct}, {f}));' for f in fields]
    prog += ["  return 0;", "}"]
    src = os.path.join("/tmp", f"rac_layout_{struct}_{os.getpid()}.c")
    exe = src[:-2]
    with open(src, "w") as fh:
        fh.write("\n".join(prog))
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    os.remove(src)
    os.remove(exe)
    return int(out[0]), [int(v) for v in out[1:]]


def test_ctypes_descriptors_match_the_header_layout():
    from racformer_amd import _lib
    header = open(os.path.join(ROOT, "include", "racformer_hip.h")).read()
    for struct, mirror in (("rac_cd_scale", _lib.CdScale), ("rac_cd_frames", _lib.CdFrames), ("rac_conv_direct", _lib.ConvDirect),
                           ("rac_rowseg", _lib.RowSeg), ("rac_rowgemm", _lib.RowGemm)):
        end = header.index("} " + struct + ";")
        start = header.rindex("typedef struct", 0, end)
        text = re.sub(r"/\*.*?\*/", "", header[header.index("{", start) + 1:end], flags=re.S)
        names = []
        for decl in text.split(";"):                      # "const float *a, *b" -> a, b
            if decl.strip():
                names += [re.search(r"(\w+)\s*(?:\[\w*\])?\s*$", piece.strip()).group(1) for piece in decl.split(",")]
        assert names == [f[0] for f in mirror._fields_], (struct, names, [f[0] for f in mirror._fields_])
        size, offsets = _c_layout(struct, names)
        assert size == ctypes.sizeof(mirror), (struct, size, ctypes.sizeof(mirror))
        assert offsets == [getattr(mirror, n).offset for n in names], struct
