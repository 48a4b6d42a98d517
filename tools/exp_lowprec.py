#!/usr/bin/env python3
"""What 16-bit STORAGE of the sampled tensors costs against the reference's fp32 CPU forward (GPU box; round-3 verdict item 4).

usage: python3 tools/exp_lowprec.py [out.json]

Rigs: the reference-initialised ones (decoder_f8_init.npz, decoder_f8_3cam_init.npz: the model as the reference's own
init_weights() leaves it -- what BASELINE configs 3 / 5 describe) and two of the random-everything fixtures (decoder_f8.npz,
decoder_f8_s1.npz: the rig that amplifies rounding 4-5x per layer).  Variants, arithmetic in fp32 throughout:
  fp32                      -- the product as it stands
  pyramid bf16              -- decoder.feature_dtype = bfloat16 (the gather kernel's bf16 path, half its bytes)
  pyramid c2 bf16 / c2+c3   -- only the fine levels rounded (emulated on the regrouped fp32 pyramid)
  values bf16 / values f16  -- the two hoisted BEV value streams rounded to bf16 / f16 (emulated: rounded once after
                               prepare(), fp32 kernels), fp32 pyramid
  values f16 + pyramid bf16
The reference's camera choices are imposed on every run (so the one discontinuous step of the path is out of the comparison);
reported per layer: max / p50 box error, queries over 1e-3, class-argmax mismatches."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity import fill_rig_module  # noqa: E402
from racformer_amd import synthetic as syn  # noqa: E402
from racformer_amd.transformer import RaCFormerTransformer, regroup_pyramid  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")


def run(cfg, g, inputs, init_rig, pyramid_dtype=torch.float32, round_levels=(), value_dtype=None):
    tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
    if init_rig:
        fill_rig_module(tr, cfg, g, GOLD)
    else:
        syn.fill_params(tr, int(g["weight_seed"]))
    tr = tr.to(DEV)
    layer = tr.decoder.decoder_layer
    layer.sampling.force_views = [torch.as_tensor(np.asarray(v)).to(DEV).contiguous() for v in g["views"]]
    tr.decoder.feature_dtype = pyramid_dtype
    qb, qf, pyr, lss, radar = inputs
    feats = [f.clone() for f in pyr]
    if round_levels:
        tr.decoder.pregrouped = True
        feats = regroup_pyramid(feats, cfg.num_cams)
        for l in round_levels:
            feats[l] = feats[l].to(torch.bfloat16).to(torch.float32)
    vmax = {}
    if value_dtype is not None:
        orig = layer.prepare

        def prep(lss_, radar_):
            p = orig(lss_, radar_)
            for k in ("radar_value", "lss_value"):
                vmax[k] = float(p[k].abs().max())
                p[k] = p[k].to(value_dtype).to(torch.float32)
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


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "exp_lowprec.json")
    rigs = [("decoder_f8_init.npz", syn.F8, True), ("decoder_f8_3cam_init.npz", syn.F8_3CAM, True),
            ("decoder_f8.npz", syn.F8, False), ("decoder_f8_s1.npz", syn.F8, False)]
    variants = [("fp32", {}), ("pyramid bf16", dict(pyramid_dtype=torch.bfloat16)), ("pyramid c2 bf16", dict(round_levels=(0,))),
                ("pyramid c2+c3 bf16", dict(round_levels=(0, 1))), ("values bf16", dict(value_dtype=torch.bfloat16)),
                ("values f16", dict(value_dtype=torch.float16)),
                ("values f16 + pyramid bf16", dict(value_dtype=torch.float16, pyramid_dtype=torch.bfloat16))]
    res = {}
    for name, cfg, init_rig in rigs:
        g = np.load(os.path.join(GOLD, name))
        seed = int(g["seed"])
        qb, qf = syn.make_queries(cfg, seed)
        inputs = (qb.to(DEV), qf.to(DEV), [f.to(DEV) for f in syn.make_pyramid(cfg, seed)], syn.make_bev(cfg, seed, 0).to(DEV),
                  syn.make_bev(cfg, seed, 1).to(DEV))
        res[name] = {"rig": "reference init_weights()" if init_rig else "random-everything"}
        for vname, kw in variants:
            r = run(cfg, g, inputs, init_rig, **kw)
            res[name][vname] = r
            print(f"{name:26s} {vname:28s} max {['%.1e' % x for x in r['box_max']]} >1e-3 {r['queries_over_1e-3']} argmax {r['argmax_mismatches']}",
                  flush=True)
        del inputs
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
