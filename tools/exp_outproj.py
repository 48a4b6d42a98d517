import torch, time
dev='cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
M,K,N=900,32768,256
a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)/K**0.5; b=torch.randn(N,device=dev)
ref=(a.double()@w.double().t()+b.double())
print('linear', bench(lambda: torch.nn.functional.linear(a,w,b)))
wt=w.t().contiguous()
print('a@wt contiguous', bench(lambda: torch.addmm(b,a,wt)))
for S in (4,8,16,32,64):
    a3=a.view(M,S,K//S).transpose(0,1)            # [S,M,k] strided view
    w3=w.view(N,S,K//S).permute(1,2,0).contiguous()  # [S,k,N]
    f=lambda: torch.bmm(a3,w3).sum(0)+b
    o=f(); print('bmm split',S, bench(f), (o.double()-ref).abs().max().item())
    w3b=w.view(N,S,K//S).permute(1,0,2).contiguous()  # [S,N,k]
    f2=lambda: torch.bmm(a3,w3b.transpose(1,2)).sum(0)+b
    print('   bmm split NT',S, bench(f2))
# transposed problem: (w @ a^T)^T
print('w@a.t', bench(lambda: (w@a.t()).t()+b))
