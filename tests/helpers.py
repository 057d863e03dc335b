"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UPDATE_CASES = ["c1_cartpole", "c2_lunarlander", "c3_bipedal", "c5_halfcheetah_small",
                "edge_a1_continuous", "edge_offclip", "edge_saturated_softmax", "edge_std01"]


TWO_UPDATE_CASES = ["two_update_discrete", "two_update_continuous"]
TOL_PARAM = 1e-4   # parameters after an update (north star)


def param_tolerance(name, K):
    """Bound on max |param - reference param| after K epochs.  Up to 4 epochs: 1e-4.  Beyond that
    the reference is only defined up to its own summation-order spread: on identical data its
    update() lands up to 1.2e-4 (C1, K=80) / 3.1e-4 (C3, K=80) away from itself when the intra-op
    thread count or the sample order changes (reference_spread.json, recorded by gen_golden.py
    from the reference's own runs), because Adam's early steps move a weight by ~lr*sign(g) and a
    gradient component near fp32 noise flips sign with the order of the sum.  The bound for K > 4
    is therefore 1e-4 + that recorded spread of the case."""
    if K <= 4:
        return TOL_PARAM
    with open(os.path.join(GOLDEN, "reference_spread.json")) as f:
        spread = json.load(f)["cases"]
    return TOL_PARAM + float(spread[name]["spread"])


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
