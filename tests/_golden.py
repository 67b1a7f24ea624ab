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
