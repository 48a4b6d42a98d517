"""The built library carries no packed-FP32 VALU instruction.

Several samples are in flight per GPU (racformer_amd/graph.py): a gather kernel's waves then share a SIMD with another
stream's MFMA kernels, and in that situation ``v_pk_fma_f32`` accumulation produced wrong sums (round 3).  The library is
therefore built with ``-target-feature -packed-fp32-ops`` (csrc/Makefile); hipcc's HOST pass prints "not a recognized
feature" for that flag and ignores it, so nothing but the generated device code says whether the flag took effect.  This test
disassembles the gfx950 code objects inside libracformer_hip.so and fails on any v_pk_{fma,mul,add}_f32."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "racformer_amd", "csrc", "libracformer_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(lib, workdir):
    """Disassembly of every gfx950 code object bundled in ``lib`` (llvm-objdump --offloading unbundles next to its input,
    so it works on a copy in ``workdir``)."""
    copy = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, copy)
    subprocess.run([OBJDUMP, "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=workdir)
    objs = sorted(glob.glob(copy + ".*gfx950"))
    text = []
    for o in objs:
        text.append(subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout)
    return objs, "\n".join(text)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_library_has_no_packed_fp32_instructions(tmp_path):
    if not os.path.exists(LIB):
        pytest.fail("libracformer_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    objs, dis = device_disassembly(LIB, str(tmp_path))
    assert len(objs) >= 15, f"expected one gfx950 code object per source file, found {len(objs)}"
    # the disassembly is real: the gather kernels' scalar accumulation and the matrix-core kernels are in it
    assert len(re.findall(r"\bv_fma_f32\b", dis)) > 1000 and "v_mfma_f32_16x16x32_f16" in dis
    for kernel in ("sampling4d_c64_kernel", "bev_sampling_d64_kernel", "msmv_fwd_c64_kernel", "msda_fwd_d64_kernel"):
        assert kernel in dis, kernel
    packed = re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)
    assert not packed, f"{len(packed)} packed-FP32 instructions in the device code (csrc/Makefile's NOPK flag did not take effect)"
