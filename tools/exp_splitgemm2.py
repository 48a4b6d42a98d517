"""Experiment: K padding of the K-concatenated generator GEMM, split-K slice size of out_proj."""
import time
import torch
dev = "cuda"
torch.manual_seed(0)


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for K in (768, 776, 784, 800, 832, 896, 1024):
    A = torch.randn(900, K, device=dev).half()
    W = torch.randn(65536, K, device=dev).half()
    print(f"generator K={K}: {bench(lambda: torch.mm(A, W.t(), out_dtype=torch.float32)):.1f} us")
    A2 = torch.randn(928, K, device=dev).half()[:900]
    print(f"   (A rows padded to 928 alloc): {bench(lambda: torch.mm(A2, W.t(), out_dtype=torch.float32)):.1f} us")
for slice_ in (512, 1024, 2048, 4096):
    S = 32768 // slice_
    A = torch.randn(900, S, 3 * slice_, device=dev).half()
    W = torch.randn(S, 256, 3 * slice_, device=dev).half()
    Z = torch.zeros(S, 900, 256, device=dev)
    print(f"out_proj slice={slice_} S={S}: bmm {bench(lambda: torch.bmm(A.transpose(0, 1), W.transpose(1, 2), out_dtype=torch.float32)):.1f} us, "
          f"baddbmm(beta=0) {bench(lambda: torch.baddbmm(Z, A.transpose(0, 1), W.transpose(1, 2), beta=0, alpha=0.5, out_dtype=torch.float32)):.1f} us")
# wide GEMM and small ones: fp32 vs f16x3
for (M, K, N) in ((900, 256, 2189), (900, 256, 776), (900, 256, 256), (900, 768, 256), (900, 512, 256), (900, 256, 512)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    A3 = torch.randn(M, 3 * K + 8, device=dev).half()
    W3 = torch.randn(N, 3 * K + 8, device=dev).half()
    print(f"linear {M}x{K}x{N}: fp32 {bench(lambda: torch.nn.functional.linear(A, W, b)):.1f} us, f16x3 {bench(lambda: torch.mm(A3, W3.t(), out_dtype=torch.float32)):.1f} us")
