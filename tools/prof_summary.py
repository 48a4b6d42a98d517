"""Per-step kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): only the dispatches of the last
`--steps` bench steps are counted (warm-up, MIOpen find passes and the capture pass are cut off by time)."""
import argparse
import csv
import sqlite3
import sys

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--steps", type=int, required=True, help="timed steps of the profiled bench run")
ap.add_argument("--extra", type=int, default=1, help="untimed forwards after the timed region (bench: 1 capture pass)")
ap.add_argument("--marker", default="regroup_kernel", help="kernel that runs a fixed number of times per step")
ap.add_argument("--marker-per-step", type=int, default=4)
ap.add_argument("--csv", default="")
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = list(c.execute("select name, start, end from kernels order by start"))
marks = [r[1] for r in rows if a.marker in r[0]]
nsteps_total = len(marks) // a.marker_per_step
first = nsteps_total - a.extra - a.steps
t0 = marks[first * a.marker_per_step]
t1 = marks[(first + a.steps) * a.marker_per_step]
agg = {}
for n, s, e in rows:
    if t0 <= s < t1:
        d = agg.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += e - s
tot = sum(v[1] for v in agg.values())
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"steps {a.steps}: wall {(t1 - t0) / a.steps / 1e3:.1f} us/step, kernel time {tot / a.steps / 1e3:.1f} us/step, {sum(v[0] for v in agg.values()) / a.steps:.0f} launches/step")
for n, (cnt, t) in out[:70]:
    print(f"{cnt / a.steps:6.1f}/step {t / a.steps / 1e3:8.1f} us/step avg {t / cnt / 1e3:8.1f} us  {n[:110]}")
if a.csv:
    with open(a.csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerStep", "UsPerStep", "AverageUs", "Percentage"])
        for n, (cnt, t) in out:
            w.writerow([n, f"{cnt / a.steps:.2f}", f"{t / a.steps / 1e3:.2f}", f"{t / cnt / 1e3:.2f}", f"{100.0 * t / tot:.2f}"])
