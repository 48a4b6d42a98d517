import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle_clib():
    """The C restatement is test infrastructure; build it on demand (gcc, <1 s)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libgather_ref.so")
    src = os.path.join(ROOT, "oracle", "gather_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    """What the parity comparisons of this session actually used of their allowances (tests/parity.py ``USED``), written
    next to the other GPU-box outputs and keyed by the kind of session: ``parity_budget_used_gpu.json`` when a GPU took part
    (the ``-m gpu`` run), ``parity_budget_used_cpu.json`` otherwise (oracle-vs-golden comparisons here in the container) -- a CPU
    session can no longer overwrite the GPU session's record.  The committed copy of the GPU record lives under profiles/."""
    import json
    try:
        import parity
    except Exception:
        return
    if not parity.USED:
        return
    try:
        import torch
        kind = "gpu" if torch.cuda.is_available() else "cpu"
    except Exception:
        kind = "cpu"
    path = os.environ.get("RAC_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", f"parity_budget_used_{kind}.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dict(session=kind, exitstatus=int(exitstatus),
                           criteria=dict(tail_queries=parity.TAIL_QUERIES, tail_tol=parity.TAIL_TOL,
                                         argmax_margin=parity.ARGMAX_MARGIN, argmax_margin_init_rig=parity.ARGMAX_MARGIN_INIT_RIG,
                                         max_flipped_points=parity.MAX_FLIPPED_POINTS),
                           comparisons=parity.USED), f, indent=1)
    except OSError:
        pass
