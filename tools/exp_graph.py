"""Experiment: host enqueue time vs GPU time per step; whole-forward capture in a HIP graph (torch.cuda.CUDAGraph)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racformer_amd import synthetic as syn, _lib
import bench

dev = torch.device("cuda", 0)
cfg = syn.F8
head = bench.build_head(cfg, dev)
pyramid = [f.to(dev) for f in syn.make_pyramid(cfg, 0)]
lss, radar = syn.make_bev(cfg, 0, 0).to(dev), syn.make_bev(cfg, 0, 1).to(dev)
metas = syn.make_img_metas(cfg)


def step(m=None):
    with torch.no_grad():
        preds = head(list(pyramid), lss, radar, m if m is not None else [dict(x) for x in metas])
        return head.get_detections_fixed(preds)


for _ in range(3):
    ref = step()
torch.cuda.synchronize()
# host enqueue time vs total
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step")

# graph capture with pre-staged metas
staged = [dict(x) for x in metas]
head.transformer.decoder.stage_metas(staged, 1, dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step(staged)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = step(staged)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("graph vs eager max diff", (out - ref).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(N):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {1e3*(time.perf_counter()-t0)/N:.2f} ms/step")
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:500])
# external events
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
    g2 = torch.cuda.CUDAGraph()
    x = torch.randn(4096, 4096, device=dev)
    with torch.cuda.graph(g2):
        e0.record()
        y = x @ x
        e1.record()
    g2.replay()
    torch.cuda.synchronize()
    print("external events in graph: elapsed", e0.elapsed_time(e1), "ms")
except Exception as e:  # noqa: BLE001
    print("external events failed:", type(e).__name__, str(e)[:300])
