"""The RCCL branch of the data-parallel path (val.py:98,135 -> racformer_amd/dp.py, bench.py) on the one GPU of the test box:
a FRESH child process initialises the ``nccl`` (= RCCL) process group with ``device_id`` at world size 1, runs real bench
steps with the all-gather forced (``--force-collective`` bypasses the world-1 short-circuit of dp.all_gather_detections),
checks that the gathered block equals the local one and tears the group down; the parent only reads its exit status and
JSON line.  No scaling number can come from one GPU -- this pins that the code the driver's 8-GPU run will execute first
has executed at all."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("graph", [True, False], ids=["captured", "eager"])
def test_bench_step_with_forced_rccl_all_gather(graph):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-stress", "--force-collective"] + ([] if graph else ["--no-graph"])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["forced_at_world_1"] is True
    assert c["gathered_shape"] == [1, 1, 300, 11] and c["own_slot_equals_local"] is True
    assert line["value"] > 0 and line["n_gpus"] == 1
