import torch, time
dev='cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
shapes={'gen':(900,256,65536),'valproj':(131072,256,256),'wide':(900,256,2189),'inproj':(900,256,776)}
for lib in ('default','hipblaslt','hipblas'):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib,'unsupported',e); continue
    for name,(M,K,N) in shapes.items():
        a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)/K**0.5; b=torch.randn(N,device=dev)
        t=bench(lambda: torch.nn.functional.linear(a,w,b))
        print(lib,name,'%.1f us  %.1f TF'%(t,2*M*K*N/t/1e6))
torch.backends.cuda.preferred_blas_library('default')
M,K,N=900,256,65536
a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)/K**0.5; b=torch.randn(N,device=dev)
for ch in (4,16):
    ws=w.view(ch,N//ch,K)
    f=lambda: torch.baddbmm(b.view(ch,1,N//ch), a.expand(ch,M,K), ws.transpose(1,2))
    print('gen bmm chunks',ch,bench(f))
