import ctypes, glob, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racformer_amd import _lib, synthetic as syn
from racformer_amd.transformer import RaCFormerTransformer
cfg = syn.F8; dev = 'cuda:0'
tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval(); syn.fill_params(tr, 0); tr = tr.to(dev)
layer = tr.decoder.decoder_layer
qb, qf = syn.make_queries(cfg, 0); qf = (qf * 5).to(dev); qb = qb.to(dev)
metas = syn.make_img_metas(cfg); tr.decoder.stage_metas(metas, 1, torch.device(dev))
mod = layer.sampling_lss_bev
with torch.no_grad():
    value, hw = mod.prepare_value(syn.make_bev(cfg, 0, 0).to(dev))
    value2, _ = layer.sampling_radar_bev.prepare_value(syn.make_bev(cfg, 0, 1).to(dev))
    lin = (mod.sampling_offset(qf), mod.ray_points_offset(qf), mod.scale_weights(qf), mod.attention.bev_queue_weight(qf))
def run(v):
    return mod.attend_prepared(qb, qf, v, hw, metas[0]['time_diff'], 0.06, lin)
ref = None
for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'variants', 'lib_bev_*.so'))) * 2:
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype, fn.argtypes = res, args
    _lib._lib = h
    with torch.no_grad():
        out = run(value); torch.cuda.synchronize()
        if ref is None: ref = out
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(value); run(value2)      # alternate the two 134 MB streams like the decoder does
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(os.path.basename(path), 'us per call (incl. box_prep + output_proj): min %.1f med %.1f' % (min(ts), sorted(ts)[2]), 'maxdiff', (out - ref).abs().max().item())
