"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UPDATE_CASES = ["c1_cartpole", "c2_lunarlander", "c3_bipedal", "c5_halfcheetah_small",
                "edge_a1_continuous", "edge_offclip", "edge_saturated_softmax", "edge_std01"]


TWO_UPDATE_CASES = ["two_update_discrete", "two_update_continuous"]
TOL_PARAM = 1e-4   # parameters after an update (north star)


def load_branches(name):
    """End points of the reference's own update() on this case under re-ordered summations
    (gen_golden.py: reference_spread): float32 [n_branches, P], flat state_dict order."""
    return np.load(os.path.join(GOLDEN, name + "_branches.npz"))["finals"]


def load_trajectory(name):
    """Golden run's flat parameter vector after selected epochs + the reference's per-epoch spread."""
    z = np.load(os.path.join(GOLDEN, name + "_traj.npz"))
    return [int(k) for k in z["epochs"]], {int(k): z[f"K{int(k)}"] for k in z["epochs"]}, z["spread_curve"]


def nearest_branch(flat_params, name):
    """(max |param - reference end point|, index) for the closest of the reference's recorded end points
    (index 0 is the golden run's own end point)."""
    fin = load_branches(name).astype(np.float64)
    dist = np.abs(fin - np.asarray(flat_params, np.float64)[None, :]).max(axis=1)
    return float(dist.min()), int(dist.argmin())


def nearest_branch_distance(flat_params, name):
    return nearest_branch(flat_params, name)[0]


def load_branch_state(name, index):
    """The reference optimizer's (exp_avg, exp_avg_sq), flat, at recorded end point `index`."""
    z = np.load(os.path.join(GOLDEN, name + "_branches.npz"))
    return z["exp_avg"][index], z["exp_avg_sq"][index]


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
