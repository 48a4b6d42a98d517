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
    """What the parity comparisons of this session actually used of their allowances (tests/parity.py ``USED``): written
    next to the other GPU-box outputs; the committed copy lives under profiles/."""
    import json
    try:
        import parity
    except Exception:
        return
    if not parity.USED:
        return
    path = os.environ.get("RAC_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "parity_budget_used.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dict(criteria=dict(tail_queries=parity.TAIL_QUERIES, tail_tol=parity.TAIL_TOL,
                                         argmax_margin=parity.ARGMAX_MARGIN, max_flipped_points=parity.MAX_FLIPPED_POINTS),
                           comparisons=parity.USED), f, indent=1)
    except OSError:
        pass
