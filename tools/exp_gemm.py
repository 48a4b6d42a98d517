import torch, time
torch.manual_seed(0)
dev='cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
def split(x):
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16); return hi, lo
for (M,K,N) in ((900,256,65536),(900,32768,256),(131072,256,256)):
    a = torch.randn(M,K,device=dev); w = torch.randn(N,K,device=dev)/K**0.5
    ref = (a.double() @ w.double().t())
    o32 = a @ w.t()
    t32 = bench(lambda: a @ w.t())
    print(M,K,N,'fp32 us', t32, 'TF', 2*M*K*N/t32/1e6, 'err', (o32.double()-ref).abs().max().item(), 'refmax', ref.abs().max().item())
    ah, al = split(a); wh, wl = split(w)
    try:
        def f3():
            o = torch.mm(ah, wh.t(), out_dtype=torch.float32)
            o += torch.mm(ah, wl.t(), out_dtype=torch.float32)
            o += torch.mm(al, wh.t(), out_dtype=torch.float32)
            return o
        o3 = f3(); t3 = bench(f3)
        print('   bf16x3 us', t3, 'err', (o3.double()-ref).abs().max().item())
        t1 = bench(lambda: torch.mm(ah, wh.t(), out_dtype=torch.float32))
        o1 = torch.mm(ah, wh.t(), out_dtype=torch.float32)
        print('   bf16x1 us', t1, 'err', (o1.double()-ref).abs().max().item())
        # K-concatenated single GEMM: [ah|ah|al] x [wh|wl|wh]
        ac = torch.cat([ah,ah,al],1).contiguous(); wc = torch.cat([wh,wl,wh],1).contiguous()
        tc = bench(lambda: torch.mm(ac, wc.t(), out_dtype=torch.float32))
        oc = torch.mm(ac, wc.t(), out_dtype=torch.float32)
        print('   bf16x3 K-concat us', tc, 'err', (oc.double()-ref).abs().max().item())
    except Exception as e:
        print('   bf16 out_dtype failed:', repr(e)[:200])
    # fp16 variant? skip
