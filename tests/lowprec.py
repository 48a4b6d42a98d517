"""One forward of the product decoder with 16-bit STORAGE of the tensors it samples from (arithmetic stays fp32), against a reference
fixture -- shared by tests/test_lowprec_storage_gpu.py and tools/exp_lowprec.py (which prints the whole table)."""
import os

import numpy as np
import torch

from parity import fill_rig_module
from racformer_amd import synthetic as syn
from racformer_amd.transformer import RaCFormerTransformer, regroup_pyramid

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rig_inputs(cfg, seed):
    qb, qf = syn.make_queries(cfg, seed)
    return (qb.to(DEV), qf.to(DEV), [f.to(DEV) for f in syn.make_pyramid(cfg, seed)], syn.make_bev(cfg, seed, 0).to(DEV),
            syn.make_bev(cfg, seed, 1).to(DEV))


def round_values(v, how):
    """A value stream [B*T, H*W, heads, 64] as it would read back from 16-bit storage: a torch dtype (bf16 / f16: plain rounding),
    or "i16b64" / "i16b256": int16 mantissas with one power-of-two scale per (pixel, head) block of 64 / per pixel (256 values) --
    block floating point, 14-15 significant bits relative to the block's largest value."""
    if not isinstance(how, str):
        return v.to(how).to(torch.float32)
    blk = v if how == "i16b64" else v.reshape(v.shape[0], v.shape[1], 1, -1)            # (i16b64: blocks along the last dimension)
    m = blk.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(14.0 - torch.floor(torch.log2(m)))                 # |v| * scale < 2^15
    q = torch.round(blk * scale).clamp_(-32767, 32767)
    return (q / scale).reshape(v.shape)


def run(cfg, g, inputs, init_rig, pyramid_dtype=torch.float32, round_levels=(), value_dtype=None, level_rounding=torch.bfloat16):
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    if init_rig:
        fill_rig_module(tr, cfg, g, GOLD)
    else:
        syn.fill_params(tr, int(g["weight_seed"]))
    tr = tr.to(DEV)
    layer = tr.decoder.decoder_layer
    layer.value_storage = "f32"       # (the formats measured here are emulated on the fp32 streams)
    layer.sampling.force_views = [torch.as_tensor(np.asarray(v)).to(DEV).contiguous() for v in g["views"]]
    tr.decoder.feature_dtype = pyramid_dtype
    qb, qf, pyr, lss, radar = inputs
    feats = [f.clone() for f in pyr]
    if round_levels:
        tr.decoder.pregrouped = True
        feats = regroup_pyramid(feats, cfg.num_cams)
        for l in round_levels:
            feats[l] = round_values(feats[l], level_rounding)     # (grouped layout [S,N,H,W,64]: a block = one pixel's 64 channels of one group)
    vmax = {}
    if value_dtype is not None:
        orig = layer.prepare

        def prep(lss_, radar_):
            p = orig(lss_, radar_)
            for k in ("radar_value", "lss_value"):
                vmax[k] = float(p[k].abs().max())
                p[k] = round_values(p[k], value_dtype)
            return p
        layer.prepare = prep
    with torch.no_grad():
        cls, box = tr(qb, qf, feats, lss, radar, None, syn.make_img_metas(cfg))
    torch.cuda.synchronize()
    gb, gc = torch.from_numpy(np.asarray(g["box"])), torch.from_numpy(np.asarray(g["cls"]))
    eb = (box.cpu() - gb).abs().amax(-1).flatten(1)
    mism = (cls.cpu().argmax(-1) != gc.argmax(-1)).flatten(1)
    return {"box_max": [float(x) for x in eb.max(1).values], "box_p50": [float(x) for x in eb.median(1).values],
            "queries_over_1e-3": [int(x) for x in (eb > 1e-3).sum(1)], "argmax_mismatches": [int(x) for x in mism.sum(1)],
            **({"value_abs_max": vmax} if vmax else {})}
