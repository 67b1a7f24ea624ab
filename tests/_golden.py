"""Fixture loading helpers shared by CPU and GPU tests (data only; see oracle/make_golden.py)."""
import os

import numpy as np
import torch

from synth import synth_state  # oracle/synth.py (test infrastructure)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __contains__(self, k):
        return k in self.z.files

    def t(self, k):
        a = self.z[k]
        return torch.from_numpy(a) if a.dtype.kind in "fiub" else a

    def i(self, k):
        return int(self.z[k])

    def f(self, k):
        return float(self.z[k])

    def state(self):
        return synth_state(self.z["manifest_names"], self.z["manifest_shapes"], int(self.z["synth_seed"]))

    def shapes(self, k="shapes"):
        return [tuple(int(v) for v in r) for r in self.z[k]]


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


# ---- achieved parity margins ("margins are invisible") ------------------------------------------------------------------
# Tests that hold the product to the oracle record what they ACHIEVED, not only that it was under the bar: max |difference|, the scale
# it is judged against and the tolerance.  conftest.pytest_sessionfinish writes the table to gpurun_out/parity_margins.txt; the round's
# copy is committed under profiles/.
MARGINS = {}          # (group, stage) -> dict(abs=, rel=, scale=, tol=, n=)


def record_margin(group, stage, abs_err, scale=1.0, tol=None):
    """Keep the worst achieved error per (group, stage).  rel = abs_err / scale (the scale the tolerance is relative to)."""
    key = (str(group), str(stage))
    rel = float(abs_err) / max(float(scale), 1e-30)
    m = MARGINS.get(key)
    if m is None or rel > m["rel"]:
        MARGINS[key] = {"abs": float(abs_err), "rel": rel, "scale": float(scale), "tol": tol, "n": (m["n"] if m else 0) + 1}
    else:
        m["n"] += 1
    return float(abs_err)


def margins_table():
    rows = ["%-66s %-56s %12s %12s %10s %8s %5s" % ("group", "stage", "max abs err", "scale", "rel err", "tol", "n")]
    for (g, st), m in sorted(MARGINS.items()):
        rows.append("%-66s %-56s %12.3e %12.3e %10.2e %8s %5d" % (g, st, m["abs"], m["scale"], m["rel"],
                                                                     "%.0e" % m["tol"] if m["tol"] is not None else "-", m["n"]))
    return "\n".join(rows) + "\n"
