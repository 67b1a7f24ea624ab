import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's GEMMs on at most 16 torch threads: a GPU box shows 256 logical CPUs of a shared host, and on all of them the oracle
    # runs 5x SLOWER than on 16 (bench.py's cpu_baseline sweep: 0.12 frames/s on 128 threads against 0.6 on 16) -- round 6 found the
    # full-size tests spending most of their time there.  MDQE_TEST_THREADS overrides.
    try:
        import torch
        torch.set_num_threads(max(1, min(int(os.environ.get("MDQE_TEST_THREADS", "16")), os.cpu_count() or 1)))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Achieved parity margins of this session (tests/_golden.record_margin) -> gpurun_out/parity_margins.txt."""
    try:
        from _golden import MARGINS, margins_table
    except Exception:
        return
    if not MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.txt"), "w") as f:
        f.write("# achieved error of the product (HIP, through the C ABI) against the CPU oracle, worst case per stage; rel err = abs / scale;\n"
                "# the bar (tol) is relative to `scale`.  Written by tests/conftest.py at the end of a pytest session (exit status %s).\n" % exitstatus)
        f.write(margins_table())
