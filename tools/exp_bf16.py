"""bf16 feature-storage mode: decoder parity vs the reference golden (normalised box space) and kernel time."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from racformer_amd import _lib, synthetic as syn
from racformer_amd.transformer import RaCFormerTransformer
from parity import decoder_parity
dev = "cuda:0"
cfg = syn.F8
g = np.load(os.path.join(ROOT, "tests/golden/decoder_f8.npz"))
for fdt in (torch.float32, torch.bfloat16):
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval(); syn.fill_params(tr, 0); tr = tr.to(dev)
    tr.decoder.feature_dtype = fdt
    qb, qf = syn.make_queries(cfg, 0)
    pyr = [f.to(dev) for f in syn.make_pyramid(cfg, 0)]
    lss, radar = syn.make_bev(cfg, 0, 0).to(dev), syn.make_bev(cfg, 0, 1).to(dev)
    _lib.timer = _lib.KernelTimer()
    with torch.no_grad():
        for _ in range(3):
            cls, box = tr(qb.to(dev), qf.to(dev), list(pyr), lss, radar, None, syn.make_img_metas(cfg))
    torch.cuda.synchronize()
    print(fdt, "sampling4d ms", _lib.timer.mean_ms("sampling4d_fwd"))
    _lib.timer = None
    try:
        decoder_parity(cls, box, g["cls"], g["box"], what=str(fdt))
    except AssertionError as e:
        print("PARITY FAIL", str(e)[:1500])
