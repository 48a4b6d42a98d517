"""Host-side profile of one bench step (cProfile over 20 steps after warm-up): where the Python time per step goes."""
import cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from racformer_amd import _lib, dp, synthetic as syn  # noqa: E402

cfg = syn.F8
dev = torch.device("cuda", 0)
head = bench.build_head(cfg, dev)
pyramid = [f.to(dev) for f in syn.make_pyramid(cfg, 0)]
lss, radar = syn.make_bev(cfg, 0, 0).to(dev), syn.make_bev(cfg, 0, 1).to(dev)
metas = syn.make_img_metas(cfg)


def step():
    with torch.no_grad():
        fresh = [dict(m) for m in metas]
        preds = head(list(pyramid), lss, radar, fresh)
        det = head.get_detections_fixed(preds)
        return dp.all_gather_detections(det)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:6000])
