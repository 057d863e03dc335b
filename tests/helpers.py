"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UPDATE_CASES = ["c1_cartpole", "c2_lunarlander", "c3_bipedal", "c5_halfcheetah_small",
                "edge_a1_continuous", "edge_offclip", "edge_saturated_softmax", "edge_std01"]


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def case_params(d, prefix="param/"):
    return {k[len(prefix):]: np.array(v, dtype=np.float32) for k, v in d.items()
            if k.startswith(prefix)}


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor) with floor defaulting to the mean magnitude of b."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if floor is None:
        floor = max(float(np.abs(b).mean()), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max()) if a.size else 0.0


def max_abs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
