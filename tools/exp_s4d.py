import ctypes, glob, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racformer_amd import _lib, synthetic as syn, fused
from racformer_amd.transformer import RaCFormerTransformer
from oracle import restate as R
cfg = syn.F8
dev = 'cuda:0'
tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval(); syn.fill_params(tr, 0); tr = tr.to(dev)
layer = tr.decoder.decoder_layer
qb, qf = syn.make_queries(cfg, 0); qf = (qf * 5).to(dev); qb = qb.to(dev)
metas = syn.make_img_metas(cfg); tr.decoder.stage_metas(metas, 1, torch.device(dev))
feats = [f.to(dev) for f in R.regroup_pyramid(syn.make_pyramid(cfg, 0), cfg.num_cams)]
lin = (layer.sampling.sampling_offset(qf), layer.sampling.ray_points_offset(qf), layer.sampling.scale_weights(qf))
def run():
    return layer.sampling(qb, qf, feats, metas, d_region=0.06, linear_out=lin)
ref = None
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'variants', 'lib_s4d_*.so'))):
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype, fn.argtypes = res, args
    _lib._lib = h
    with torch.no_grad():
        out = run(); torch.cuda.synchronize()
        if ref is None: ref = out
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(os.path.basename(path), 'us per call (incl. box_prep + launch): min %.1f med %.1f' % (min(ts), sorted(ts)[2]), 'maxdiff', (out - ref).abs().max().item())
