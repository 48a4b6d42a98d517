"""bench.py's own multi-rank launcher, rehearsed on CPU: `python bench.py --gpus 2` (no RANK in the environment) must start
its two ranks itself, rendezvous on 127.0.0.1, run the barrier / all-gather / rank-interleaved merge of the real step
(racformer_amd/dp.py, here over gloo with a stand-in detection block) and print ONE JSON line from rank 0; a failing rank
must surface as a non-zero exit status of the parent."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_launches_its_own_ranks_world2():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["data"] == "plumbing-only" and out["value"] is None
    assert out["merged_sample_order"] == [0.0, 1.0, 2.0, 3.0, 4.0]            # dataset order restored, padding dropped
    assert out["gathered_shape"] == [2, 1, 300, 11] and len(out["per_rank_ms"]) == 2


def test_bench_rank_refuses_world_size_mismatch():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--plumbing-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_parent_reports_child_failure():
    # RAC_BENCH_TEST_FAIL makes the last rank exit non-zero inside torch.distributed.run (bench.py: plumbing_only); the parent
    # started by `--gpus 2` must pass that status on
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--blas", "x", "--steps", "-1", "--config", "f8",
                        "--warmup", "0"], env=dict(_env(), RAC_BENCH_TEST_FAIL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
