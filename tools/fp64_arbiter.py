#!/usr/bin/env python3
"""tools/fp64_arbiter.py -- an independent arbiter for the free-running parity criterion (TEST INFRASTRUCTURE).

tests/parity.py allows a small tail on the random-everything rig because two fp32 implementations of the same arithmetic drift
apart there (the rig amplifies rounding 4-5x per layer).  That justification compared fp32 against fp32.  This tool adds the
missing third party: the oracle (oracle/restate.py + the `_f64` instance of oracle/gather_ref.c) evaluated in FLOAT64 on the same
inputs (the fixtures' float32 inputs and weights, upcast) with the REFERENCE's camera choices imposed (the fixture's `views`: the
path's one discontinuous step is taken out, as in every free-running comparison).  Against that trajectory

    |reference fixture (fp32, the reference's own CPU forward) - fp64|     and     |GPU (fp32) - fp64|

are both plain rounding-error measurements, and the product can be asked to be no further from the truth than the reference is
(tests/test_fp64_arbiter_gpu.py: per layer, GPU error <= 1.5 x the reference's).

  python tools/fp64_arbiter.py                 writes tests/golden/fp64_arbiter.npz (float64 cls / box of every f8 decoder fixture)
                                               and the reference's side of profiles/r05_fp64_arbiter.json
Runs in the build container only (CPU, ~1 min per fixture on 8 cores); the GPU test reads the committed .npz.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIXTURES = [("decoder_f8", "F8"), ("decoder_f8_s1", "F8"), ("decoder_f8_s2", "F8"), ("decoder_f8_s3", "F8"),
            ("decoder_f8_3cam", "F8_3CAM"), ("decoder_f8_3cam_s1", "F8_3CAM"),
            ("decoder_f8_init", "F8"), ("decoder_f8_3cam_init", "F8_3CAM")]
GOLDEN = os.path.join(ROOT, "tests", "golden")


def stats(a, b):
    """per layer: max / p99 / median of the per-query error (max over the 10 components) and queries over 1e-3"""
    e = (torch.as_tensor(np.asarray(a)).double() - torch.as_tensor(np.asarray(b)).double()).abs().amax(-1).flatten(1)
    return dict(max=e.max(1).values.tolist(), p99=e.quantile(0.99, dim=1).tolist(), p50=e.median(1).values.tolist(),
                over_1e3=(e > 1e-3).sum(1).tolist())


def fp64_decoder(name, cfg_name):
    from oracle import restate as R
    from parity import load_rig_state_dict, oracle_decoder
    from racformer_amd import synthetic as syn
    cfg = getattr(syn, cfg_name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    seed = int(g["seed"])
    sd = {k: v.double() for k, v in load_rig_state_dict(cfg, g, GOLDEN).items()}
    qb, qf = syn.make_queries(cfg, seed)
    pyr = [f.double() for f in syn.make_pyramid(cfg, seed)]
    lss, radar = syn.make_bev(cfg, seed, 0).double(), syn.make_bev(cfg, seed, 1).double()
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)          # (the oracle's own constants -- zeros, linspace -- follow the default type)
    try:
        cls, box, views = oracle_decoder(R, sd, qb.double(), qf.double(), pyr, lss, radar, syn.make_img_metas(cfg), cfg,
                                         force_views=g["views"])
    finally:
        torch.set_default_dtype(prev)
    assert cls.dtype == torch.float64 and box.dtype == torch.float64
    own_differs = int((views != torch.as_tensor(g["views"])).sum())
    return g, cls, box, own_differs


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out, record = {}, {}
    for name, cfg_name in FIXTURES:
        t0 = time.time()
        g, cls, box, own = fp64_decoder(name, cfg_name)
        out[name + "_cls"], out[name + "_box"] = cls.numpy(), box.numpy()
        record[name] = dict(reference_vs_fp64=dict(box=stats(g["box"], box), cls=stats(g["cls"], cls)),
                            points_where_fp64_own_choice_differs_from_imposed=own,
                            rig="reference-initialised" if name.endswith("_init") else "random-everything")
        r = record[name]["reference_vs_fp64"]["box"]
        print(f"{name}: {time.time() - t0:.0f} s; |reference - fp64| box max per layer {['%.1e' % v for v in r['max']]} "
              f"over 1e-3: {r['over_1e3']}; fp64's own camera choice differs at {own} points", flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "fp64_arbiter.npz"), **out)
    path = os.path.join(ROOT, "profiles", "r05_fp64_arbiter.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    for k, v in record.items():
        old.setdefault(k, {}).update(v)
    old["_what"] = ("per decoder fixture and layer: error of the reference's own fp32 CPU forward (the fixture) and of the GPU's fp32 result "
                    "against the oracle evaluated in float64 with the reference's camera choices imposed (tools/fp64_arbiter.py; "
                    "gpu_vs_fp64 is written by tests/test_fp64_arbiter_gpu.py on the GPU box); error = max over a query's 10 box / "
                    "class components, statistics over the 900 queries")
    json.dump(old, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
