import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
