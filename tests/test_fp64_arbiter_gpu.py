"""An independent arbiter for the free-running parity criterion.

tests/golden/fp64_arbiter.npz holds, for every f8 decoder fixture, the oracle evaluated in FLOAT64 on the fixture's inputs with the
reference's camera choices imposed (tools/fp64_arbiter.py, run in the build container).  Against that trajectory the reference's
own fp32 CPU forward (the fixture) and the GPU's fp32 forward are both plain rounding-error measurements.  The product is asked to
be no further from the float64 result than the reference itself is, per layer, box and class logits alike:
  * the per-layer error (median over the 900 queries of the per-query error): GPU <= 1.5 x the reference's (measured 0.87-1.05 x on
    all eight fixtures: the GPU is, if anything, slightly CLOSER to float64 than the reference's CPU forward);
  * its 99th percentile: GPU <= 1.5 x the reference's + 5e-5 (a twentieth of north_star's tolerance; measured 0.73-1.71 x, the 1.71
    at 7.6e-5 against 4.5e-5);
  * queries beyond north_star's 1e-3: GPU <= 1.5 x the reference's count + 1 (measured 4 / 3 / 2, 10 against the reference's own
    5 / 3 / 1, 10 on the three chaotic seeds -- the reference itself misses 1e-3 against float64 there);
  * the single worst query: GPU <= max(4 x the reference's worst, 1e-3).  A maximum over 900 heavy-tailed errors on a rig that
    amplifies rounding 4-5 x per layer is not a robust statistic -- between these two fp32 implementations it ranges from 0.30 x to
    3.8 x with no trend -- so it is bounded loosely and the count above carries the tail.
What the GPU session measured is written to gpurun_out/fp64_arbiter_gpu.json (committed copy: profiles/r05_fp64_arbiter.json)."""
import json
import os

import numpy as np
import pytest
import torch

from racformer_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANDOM_RIG = [("decoder_f8", syn.F8), ("decoder_f8_s1", syn.F8), ("decoder_f8_s2", syn.F8), ("decoder_f8_s3", syn.F8),
              ("decoder_f8_3cam", syn.F8_3CAM), ("decoder_f8_3cam_s1", syn.F8_3CAM)]
INIT_RIG = [("decoder_f8_init", syn.F8), ("decoder_f8_3cam_init", syn.F8_3CAM)]
RATIO, RATIO_MAX = 1.5, 4.0
FLOOR = 2e-6          # absolute floor of the median bound: a few fp32 ulps of O(1) boxes / logits
FLOOR_P99 = 5e-5      # ... of the 99th-percentile bound
TOL = 1e-3            # north_star's tolerance: a worst query below it needs no comparison


def err_stats(a, b):
    e = (torch.as_tensor(np.asarray(a)).double() - torch.as_tensor(np.asarray(b)).double()).abs().amax(-1).flatten(1)
    return dict(max=e.max(1).values.tolist(), p99=e.quantile(0.99, dim=1).tolist(), p50=e.median(1).values.tolist(),
                over_1e3=(e > 1e-3).sum(1).tolist())


def test_arbiter_fixture_is_what_the_record_says(golden_dir):
    """(no GPU) the committed float64 outputs reproduce the reference-side numbers of profiles/r05_fp64_arbiter.json, and the
    reference itself misses 1e-3 against float64 on the chaotic seeds -- the independent justification of the tail allowance."""
    z = np.load(os.path.join(golden_dir, "fp64_arbiter.npz"))
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_fp64_arbiter.json")))
    worst = 0
    for name, _ in RANDOM_RIG + INIT_RIG:
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        assert z[name + "_box"].dtype == np.float64 and z[name + "_box"].shape == g["box"].shape
        s = err_stats(g["box"], z[name + "_box"])
        want = rec[name]["reference_vs_fp64"]["box"]
        assert s["over_1e3"] == want["over_1e3"] and np.allclose(s["max"], want["max"], rtol=1e-9, atol=0)
        worst = max(worst, max(s["over_1e3"]))
        if name.endswith("_init"):
            assert max(s["max"]) < 2e-4          # the reference-initialised rig does not amplify: literal criterion there
    assert worst >= 5                            # (decoder_f8_3cam_s1: 10 queries of the reference's own last layer are over 1e-3)


def _compare(golden_dir, name, cfg, rig):
    from parity import run_with_reference_views
    from test_parity_gpu import run_decoder_gpu
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    z = np.load(os.path.join(golden_dir, "fp64_arbiter.npz"))
    seed, wseed = int(g["seed"]), int(g["weight_seed"])
    kw = dict(rig=(g, golden_dir)) if rig else {}
    (cls, box, _), nflip = run_with_reference_views(lambda force: run_decoder_gpu(cfg, seed, wseed, force, **kw), g["views"],
                                                    name + " (fp64 arbiter)")
    out = {}
    for key, got, ref in (("box", box, g["box"]), ("cls", cls, g["cls"])):
        out[key] = dict(gpu_vs_fp64=err_stats(got, z[f"{name}_{key}"]), reference_vs_fp64=err_stats(ref, z[f"{name}_{key}"]))
    out["differing_camera_choices"] = nflip
    path = os.path.join(ROOT, "gpurun_out", "fp64_arbiter_gpu.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec[name] = out
    json.dump(rec, open(path, "w"), indent=1)
    bad = []
    for key in ("box", "cls"):
        gs, rs = out[key]["gpu_vs_fp64"], out[key]["reference_vs_fp64"]
        for l in range(len(gs["max"])):
            for stat, lim in (("p50", RATIO * rs["p50"][l] + FLOOR), ("p99", RATIO * rs["p99"][l] + FLOOR_P99),
                              ("max", max(RATIO_MAX * rs["max"][l], TOL))):
                if gs[stat][l] > lim:
                    bad.append(f"{key} L{l} {stat}: GPU {gs[stat][l]:.2e} vs reference {rs[stat][l]:.2e} (limit {lim:.2e})")
            if key == "box" and gs["over_1e3"][l] > RATIO * rs["over_1e3"][l] + 1:
                bad.append(f"box L{l} queries over 1e-3: GPU {gs['over_1e3'][l]} vs reference {rs['over_1e3'][l]}")
    print(name, "box max per layer  GPU", ["%.1e" % v for v in out["box"]["gpu_vs_fp64"]["max"]], " reference",
          ["%.1e" % v for v in out["box"]["reference_vs_fp64"]["max"]], " over 1e-3  GPU", out["box"]["gpu_vs_fp64"]["over_1e3"],
          " reference", out["box"]["reference_vs_fp64"]["over_1e3"])
    assert not bad, f"{name}: the GPU is further from float64 than the reference is:\n" + "\n".join(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", RANDOM_RIG)
def test_gpu_no_further_from_fp64_than_the_reference_random_rig(golden_dir, name, cfg):
    _compare(golden_dir, name, cfg, rig=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", INIT_RIG)
def test_gpu_no_further_from_fp64_than_the_reference_init_rig(golden_dir, name, cfg):
    _compare(golden_dir, name, cfg, rig=True)
