"""The LSS stream's value projection alone at the f8 shape (8 frames x 128 x 128 x 256 -> 256), for A/B builds of value_proj.hip.

  run    : python tools/exp_value_proj.py run [reps]                 (under rocprofv3 --kernel-trace; RACFORMER_HIP_LIB picks the build)
  report : python tools/exp_value_proj.py report <run_results.db>... (per-kernel min / quartiles of the launch durations)

Between two launches a 300 MB fill goes through the caches, so every launch reads its maps from HBM as in the step.
"""
import sys


def run(reps):
    import torch
    from racformer_amd.fused import SPLIT_ACT_SCALE, pack_gemm_split_weight, value_proj_fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 256, 128, 128, generator=g).to(dev)
    add = torch.randn(128 * 128, 256, generator=g).to(dev)
    w_img, alpha = pack_gemm_split_weight((torch.randn(256, 256, generator=g) * 0.06).to(dev))
    junk = torch.empty(300 * 1024 * 1024 // 4, device=dev)
    for q16 in (False, True):
        for _ in range(reps):
            junk.fill_(1.0)
            value_proj_fused(x, w_img, alpha * SPLIT_ACT_SCALE, add=add, q16=q16)
    torch.cuda.synchronize()


def report(paths):
    import sqlite3
    for p in paths:
        db = sqlite3.connect(p)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "kernel_symbol" in t][0]
        rows = db.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id "
                          "where s.kernel_name like '%value_proj%'").fetchall()
        by = {}
        for n, d in rows:
            by.setdefault(n, []).append(d)
        for n, d in sorted(by.items()):
            d.sort()
            print(p, n.split("EEv")[0][-8:], "n", len(d), "min %.1f p25 %.1f med %.1f p75 %.1f us" %
                  (d[0] / 1e3, d[len(d) // 4] / 1e3, d[len(d) // 2] / 1e3, d[3 * len(d) // 4] / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        sys.path.insert(0, ".")
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    else:
        report(sys.argv[2:])
