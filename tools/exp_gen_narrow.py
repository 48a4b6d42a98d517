import sys, torch
sys.path.insert(0, ".")
from racformer_amd.fused import SPLIT_ACT_SCALE, generator_fused, pack_gemm_split_weight, row_gemm, row_seg, rowgemm_launch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
M, N, K = 900, 2189, 256
w_img, alpha = pack_gemm_split_weight((torch.randn(N, K, generator=g) * 0.05).to(dev))
x = torch.randn(M, K, generator=g).to(dev)
x_img = torch.empty(M, 512, device=dev, dtype=torch.float16)
dummy_w, dummy_out = torch.zeros(16, 256, device=dev), torch.empty(M, 16, device=dev)
rowgemm_launch([row_gemm([row_seg(x, split_out=x_img, split_lines=True)], dummy_w, None, dummy_out)], M)
bias = torch.randn(N, generator=g).to(dev)
junk = torch.empty(64 * 1024 * 1024 // 4, device=dev)
for _ in range(60):
    junk.fill_(1.0)
    generator_fused(x_img, w_img, bias, alpha, ld_out=2192)
torch.cuda.synchronize()
