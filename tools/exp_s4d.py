#!/usr/bin/env python3
"""A/B timing of rac_sampling4d_fwd across library builds (RACFORMER_HIP_LIB), one child per build, on the decoder's own
layer-0 / layer-3 inputs of the f8 (6-cam) and f8_3cam rigs.  usage: python tools/exp_s4d.py lib1.so lib2.so ..."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("EXP_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    from racformer_amd import synthetic as syn
    from racformer_amd.transformer import RaCFormerTransformer, regroup_pyramid
    dev = "cuda:0"
    out = []
    for name, cfg in (("6cam", syn.F8), ("3cam", syn.F8_3CAM)):
        tr = RaCFormerTransformer(**cfg.transformer_kwargs()).eval()
        syn.fill_params(tr, 0)
        tr = tr.to(dev)
        layer = tr.decoder.decoder_layer
        qb, qf = syn.make_queries(cfg, 0)
        qb, qf = qb.to(dev), (qf * 5).to(dev)
        metas = syn.make_img_metas(cfg)
        tr.decoder.stage_metas(metas, 1, torch.device(dev))
        feats = regroup_pyramid([f.to(dev) for f in syn.make_pyramid(cfg, 0)], cfg.num_cams)
        spoil = torch.empty(300 * 1024 * 1024 // 4, device=dev)
        ts = []
        with torch.no_grad():
            for i in range(13):
                spoil.fill_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                layer.sampling(qb, qf, feats, metas, d_region=cfg.d_region_list[0])
                b.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ts.append(a.elapsed_time(b) * 1e3)
        out.append(f"{name} {statistics.median(ts):.1f}")
        del tr, feats
    print(" | ".join(out), flush=True)
    sys.exit(0)

libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, EXP_CHILD="1", RACFORMER_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(f"round {rnd} {os.path.basename(lib):28s} {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
