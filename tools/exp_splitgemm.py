"""Experiment: fp32-accurate GEMMs on the bf16/f16 matrix cores by operand splitting.
a = a_hi + a_lo (+ a_lo2): C ~= sum of the leading cross products, accumulated in fp32.
Measures error vs float64 and time vs the fp32 rocBLAS GEMM for the shapes on the hot path."""
import sys
import time

import torch

dev = "cuda"
torch.manual_seed(0)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def split(x, dt, terms):
    parts, r = [], x
    for _ in range(terms):
        p = r.to(dt)
        parts.append(p)
        r = r - p.float()
    return parts


def run(M, K, N, tag):
    A = torch.randn(M, K, device=dev) * 0.5
    W = torch.randn(N, K, device=dev) * 0.06
    ref = (A.double() @ W.double().t())
    scale = ref.abs().mean().item()

    def report(name, fn, flops_mult=1.0):
        try:
            out = fn()
        except Exception as e:  # noqa: BLE001
            print(f"{tag} {name}: FAILED {type(e).__name__}: {str(e)[:200]}")
            return
        err = (out.double() - ref).abs()
        us = bench(fn)
        print(f"{tag} {name}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF(alg)  max_err/mean|C| {err.max().item()/scale:.2e} "
              f"mean {err.mean().item()/scale:.2e}")

    report("fp32 F.linear", lambda: torch.nn.functional.linear(A, W))
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        a2, w2 = split(A, dt, 2), split(W, dt, 2)
        A3 = torch.cat([a2[0], a2[0], a2[1]], 1).contiguous()
        W3 = torch.cat([w2[0], w2[1], w2[0]], 1).contiguous()
        report(f"{name} 1-term (plain)", lambda: torch.mm(a2[0], w2[0].t(), out_dtype=torch.float32))
        report(f"{name} 2-term/3-product Kcat", lambda: torch.mm(A3, W3.t(), out_dtype=torch.float32))
        W3t = W3.t().contiguous()
        report(f"{name} 2-term/3-product Kcat (W pre-transposed)", lambda: torch.mm(A3, W3t, out_dtype=torch.float32))
        if dt == torch.bfloat16:
            a3, w3 = split(A, dt, 3), split(W, dt, 3)
            A6 = torch.cat([a3[0], a3[0], a3[1], a3[0], a3[1], a3[2]], 1).contiguous()
            W6 = torch.cat([w3[0], w3[1], w3[0], w3[2], w3[1], w3[0]], 1).contiguous()
            report(f"{name} 3-term/6-product Kcat", lambda: torch.mm(A6, W6.t(), out_dtype=torch.float32))


print(torch.__version__, torch.cuda.get_device_name(0))
run(900, 256, 65536, "generator")
run(900, 32768, 256, "out_proj")
run(16384 * 8, 256, 256, "value_proj")
run(900, 256, 2189, "wide")
sys.stdout.flush()

# conv: MIOpen find mode
import torch.nn.functional as F  # noqa: E402
x = torch.randn(8, 320, 128, 128, device=dev)
w = torch.randn(256, 320, 3, 3, device=dev) * 0.02
for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    us = bench(lambda: F.conv2d(x, w, padding=1), n=5)
    print(f"conv 320->256 3x3 fp32 benchmark={bm}: {us:.0f} us  {2*8*128*128*256*320*9/us/1e6:.1f} TF")
xb, wb = x.bfloat16(), w.bfloat16()
us = bench(lambda: F.conv2d(xb, wb, padding=1), n=5)
print(f"conv bf16: {us:.0f} us")
xb = xb.to(memory_format=torch.channels_last)
wb = wb.to(memory_format=torch.channels_last)
us = bench(lambda: F.conv2d(xb, wb, padding=1), n=5)
print(f"conv bf16 NHWC: {us:.0f} us")
