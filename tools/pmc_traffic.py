#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs: per-kernel average FETCH_SIZE / WRITE_SIZE (separate passes, as
MI355X_MICROARCH.md prescribes) -> profiles/<tag>_pmc_traffic.json.
usage: pmc_traffic.py <tag> <dir_with_fetch_pass> <dir_with_write_pass>      (tag ends in the bench config: r02_pmc_traffic_f8)
The summary records the sha256 of the gather kernels' source files: bench.py reports `roofline.traffic` only from a summary
captured from the very kernel version it is timing."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["racformer_amd/csrc/sampling_fused.hip", "racformer_amd/csrc/bev_fused.hip", "racformer_amd/csrc/mixing.hip",
           "racformer_amd/csrc/gemm_split.hip"]


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    tag, dfetch, dwrite = sys.argv[1:4]
    fetch, write = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    out = {"_source_sha": {rel: hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16] for rel in SOURCES
                           if os.path.exists(os.path.join(ROOT, rel))}}
    for k in sorted(set(fetch) | set(write)):
        if not any(s in k for s in ("sampling4d", "bev_sampling", "msmv_fwd", "msda_fwd", "regroup", "sasa", "mixing", "conv3x3", "conv_pack", "rowgemm", "absmax", "generator", "gemm_split", "decode", "value_proj")):
            continue
        f = fetch.get(k, [])
        w = write.get(k, [])
        # steady-state launches only: drop the first third (cold caches / warm-up)
        f2, w2 = f[len(f) // 3:], w[len(w) // 3:]
        fk = sum(f2) / len(f2) if f2 else None
        wk = sum(w2) / len(w2) if w2 else None
        out[k[:80]] = {
            "launches": len(f), "FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk,
            # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) reads -> x2
            "read_bytes_corrected": fk * 1024 * 2 if fk is not None else None,
            "write_bytes": wk * 1024 if wk is not None else None,
        }
    path = os.path.join(ROOT, "profiles", f"{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
