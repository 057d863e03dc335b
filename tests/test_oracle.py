"""Pins the CPU restatement (oracle/) against vectors captured from the reference itself.

CPU-only.  The goldens were produced by tests/golden/gen_golden.py running the reference's
own PPO.update / select_action / decay_action_std (PPO.py:124-258).
"""
import json
import os

import numpy as np
import pytest

from helpers import (GOLDEN, TWO_UPDATE_CASES, UPDATE_CASES, case_params, load_case, load_trajectory,
                     max_abs, nearest_branch_distance, rel_err)
from oracle import ppo_oracle as O

# tolerances (north star: 1e-5 relative on returns/advantages/log-probs/loss, 1e-4 on params)
TOL_REL = 1e-5
TOL_PARAM = 1e-4

# After more than a handful of epochs the reference's result is not a single point: on identical data
# its own update() ends up to 1.5e-4 (C1, K=80) / 4.1e-4 (C3, K=80) away from its golden run when only
# the order of its fp32 sums changes (intra-op threads, sample order), on a small number of distinct
# end points (tests/golden/reference_spread.json, <case>_branches.npz, recorded by gen_golden.py from
# the reference's own runs).  So for K > 4 the checks are: (1) within 1e-4 of the NEAREST recorded
# end point of the reference, and (2) along the golden run's trajectory, within 1e-4 + the
# reference's own spread at that epoch.  K <= 4 keeps the plain 1e-4 bound against the golden run.


def _run(d, K, record=False):
    cont = bool(d["continuous"])
    std = float(d["action_std"]) if cont else None
    P = case_params(d)
    st = O.adam_init(P)
    rec = O.update(P, st, d["states"], d["actions"], d["old_logprobs"], d["old_values"],
                   d["rewards"], d["terminals"], float(d["gamma"]), float(d["eps_clip"]), K,
                   float(d["lr_actor"]), float(d["lr_critic"]), cont, std, record=record)
    return P, st, rec


@pytest.mark.parametrize("name", UPDATE_CASES)
def test_first_epoch_intermediates(name):
    d = load_case(name)
    K = int(d["K_list"][0])
    P, st, rec = _run(d, K, record=True)
    # returns: fp64 accumulation is the same serial recurrence -> exact; fp32 rounding exact
    assert max_abs(rec["returns64"], d["returns64"]) == 0.0
    assert np.array_equal(rec["returns32"], d["returns32"])
    assert rel_err(rec["ret_mean"], d["ret_mean"]) < TOL_REL
    assert rel_err(rec["ret_std"], d["ret_std"]) < TOL_REL
    assert rel_err(rec["returns_norm"], d["returns_norm"]) < TOL_REL
    assert rel_err(rec["advantages"], d["advantages"]) < TOL_REL
    assert rel_err(rec["logprobs_e1"], d["logprobs_e1"]) < TOL_REL
    assert rel_err(rec["state_values_e1"], d["state_values_e1"]) < TOL_REL
    assert rel_err(rec["entropy_e1"], d["entropy_e1"]) < TOL_REL
    assert rel_err(rec["ratios_e1"], d["ratios_e1"]) < TOL_REL
    assert rel_err(rec["loss_e1"], d["loss_per_sample_e1"]) < TOL_REL
    assert rel_err(rec["loss_mean_e1"], d["loss_mean_e1"]) < TOL_REL
    for k in O.PARAM_KEYS:
        g = d["grad_e1/" + k]
        # gradients: relative to the largest component of the tensor (tiny components are
        # rounding noise of an fp32 sum over T samples in both implementations)
        assert rel_err(rec["grads_e1"][k], g, floor=float(np.abs(g).max())) < 1e-4, k
    for k in O.PARAM_KEYS:
        assert max_abs(P[k], d[f"K{K}/param/{k}"]) < TOL_PARAM, k
        assert max_abs(st["m"][k], d[f"K{K}/exp_avg/{k}"]) < 1e-6, k
    assert st["step"] == int(d[f"K{K}/step"])


@pytest.mark.parametrize("name", UPDATE_CASES)
def test_params_after_k_epochs(name):
    d = load_case(name)
    for K in d["K_list"][1:]:
        K = int(K)
        P, st, rec = _run(d, K)
        if K <= 4:
            worst = max(max_abs(P[k], d[f"K{K}/param/{k}"]) for k in O.PARAM_KEYS)
        else:
            worst = nearest_branch_distance(O.flatten_params(P), name)
        assert worst < TOL_PARAM, (K, worst)
        assert rel_err(rec["loss_mean_all"], d[f"K{K}/loss_mean_all"]) < 1e-4
        assert st["step"] == int(d[f"K{K}/step"]) == K
        if K <= 4:   # Adam moments against the reference's optimizer state
            for k in O.PARAM_KEYS:
                m, v = d[f"K{K}/exp_avg/{k}"], d[f"K{K}/exp_avg_sq/{k}"]
                assert max_abs(st["m"][k], m) < 1e-4 * max(float(np.abs(m).max()), 1e-30), k
                assert max_abs(st["v"][k], v) < 1e-4 * max(float(np.abs(v).max()), 1e-30), k


@pytest.mark.parametrize("name", ["c1_cartpole", "c2_lunarlander", "c3_bipedal"])
def test_trajectory_within_reference_spread(name):
    """Parameters after 1, 2, 4, 8, ... epochs against the golden run's trajectory."""
    d = load_case(name)
    epochs, snaps, curve = load_trajectory(name)
    cont = bool(d["continuous"])
    P = case_params(d)
    st = O.adam_init(P)
    ret64 = O.returns_scan(d["rewards"], d["terminals"], float(d["gamma"]))
    _, _, _, ret_n = O.normalize_returns(ret64)
    adv = (ret_n - d["old_values"]).astype(np.float32)
    for e in range(1, epochs[-1] + 1):
        _, grads = O.ppo_loss_and_grads(P, d["states"], d["actions"], d["old_logprobs"], adv, ret_n,
                                        float(d["eps_clip"]), cont, float(d["action_std"]) if cont else None)
        O.adam_step(P, grads, st, float(d["lr_actor"]), float(d["lr_critic"]))
        if e in snaps and e < epochs[-1]:
            assert max_abs(O.flatten_params(P), snaps[e]) < TOL_PARAM + float(curve[e - 1]), e


@pytest.mark.parametrize("name", TWO_UPDATE_CASES)
def test_two_consecutive_updates_keep_adam_state(name):
    """update, refill, update on ONE optimizer (PPO.py:139 is its only construction): step count
    and moments carry over; compared with the reference after each update."""
    d = load_case(name)
    cont, K = bool(d["continuous"]), int(d["K"])
    P = case_params(d)
    st = O.adam_init(P)
    for u in (1, 2):
        std = float(d[f"u{u}/action_std"]) if cont else None
        O.update(P, st, d[f"u{u}/states"], d[f"u{u}/actions"], d[f"u{u}/old_logprobs"],
                 d[f"u{u}/old_values"], d[f"u{u}/rewards"], d[f"u{u}/terminals"], float(d["gamma"]),
                 float(d["eps_clip"]), K, float(d["lr_actor"]), float(d["lr_critic"]), cont, std)
        assert st["step"] == int(d[f"u{u}/after/step"]) == u * K
        for k in O.PARAM_KEYS:
            assert max_abs(P[k], d[f"u{u}/after/param/{k}"]) < TOL_PARAM, (u, k)
            m, v = d[f"u{u}/after/exp_avg/{k}"], d[f"u{u}/after/exp_avg_sq/{k}"]
            assert max_abs(st["m"][k], m) < 1e-4 * max(float(np.abs(m).max()), 1e-30), (u, k)
            assert max_abs(st["v"][k], v) < 1e-4 * max(float(np.abs(v).max()), 1e-30), (u, k)
    if cont:
        assert float(d["u2/action_std"]) != float(d["u1/action_std"])   # the schedule ran in between


def test_scan_cases():
    d = np.load(os.path.join(GOLDEN, "scan_cases.npz"))
    for i in range(int(d["n_cases"])):
        r64 = O.returns_scan(d[f"{i}/rewards"], d[f"{i}/terminals"], float(d[f"{i}/gamma"]))
        assert max_abs(r64, d[f"{i}/returns64"]) == 0.0, i
        r32, mean, std, norm = O.normalize_returns(r64)
        assert np.array_equal(r32, d[f"{i}/returns32"]), i
        assert rel_err(mean, d[f"{i}/ret_mean"], floor=float(np.abs(r32).mean())) < TOL_REL, i
        assert rel_err(std, d[f"{i}/ret_std"]) < TOL_REL, i
        assert rel_err(norm, d[f"{i}/returns_norm"]) < TOL_REL, i


def test_scan_rows_equals_flat_with_row_end_terminals():
    rng = np.random.default_rng(0)
    N, T = 7, 33
    r = rng.standard_normal((N, T)).astype(np.float32).astype(np.float64)
    term = rng.random((N, T)) < 0.1
    rows = O.returns_scan_rows(r, term, 0.97)
    forced = term.copy()
    forced[:, -1] = True
    flat = O.returns_scan(r.reshape(-1), forced.reshape(-1), 0.97).reshape(N, T)
    # a terminal on the last step of a row resets the carry BEFORE that step, which is what
    # a fresh row start (carry 0) does as well
    assert np.array_equal(rows, flat)


def test_decay_sequence():
    with open(os.path.join(GOLDEN, "decay_sequence.json")) as f:
        g = json.load(f)
    s = g["std"][0]
    for want, var in zip(g["std"][1:], g["var"][1:]):
        s = O.decay_action_std(s, g["rate"], g["min"])
        assert s == want
        v, _ = O.action_sigma(s)
        assert np.float32(var[0]) == v
    s = g["std2"][0]
    for want in g["std2"][1:]:
        s = O.decay_action_std(s, g["rate2"], g["min2"])
        assert s == want


def test_shipped_checkpoints_forward():
    d = np.load(os.path.join(GOLDEN, "checkpoints.npz"))
    for env in d["envs"]:
        P = {k[len(env) + 7:]: d[k] for k in d.files if k.startswith(f"{env}/param/")}
        assert sorted(P) == sorted(O.PARAM_KEYS)
        cont = bool(d[f"{env}/continuous"])
        logp, V, ent, _ = O.evaluate(P, d[f"{env}/states"], d[f"{env}/actions"], cont, 0.6)
        assert rel_err(V, d[f"{env}/critic_out"]) < TOL_REL, env
        assert rel_err(logp, d[f"{env}/logprobs"]) < TOL_REL, env
        assert rel_err(ent, d[f"{env}/entropy"]) < TOL_REL, env


def test_param_packing_roundtrip():
    d = load_case("c2_lunarlander")
    P = case_params(d)
    flat = O.flatten_params(P)
    assert flat.shape[0] == 9797
    Q = O.unflatten_params(flat, 8, 4)
    for k in O.PARAM_KEYS:
        assert np.array_equal(P[k], Q[k])


# ------------------------------------------------------------------------------ CartPole restatement
def test_cartpole_restatement_basic_properties():
    """The CPU restatement of CartPole-v1 used to check the device env: reset range, determinism,
    the known one-step response to a push from rest, random-policy episode lengths, time limit."""
    from oracle import cartpole_oracle as C
    env = C.VecCartPole(256, seed=5)
    o = env.reset()
    assert o.shape == (256, 4) and o.dtype == np.float32 and np.abs(o).max() < 0.05
    assert np.array_equal(C.VecCartPole(256, seed=5).reset(), o)          # same seed, same start
    assert not np.array_equal(C.VecCartPole(256, seed=6).reset(), o)
    # from rest, a push to the right: x_dot += tau * F / (m_c + m_p) * (1 + small pole term), theta_dot < 0
    s0 = np.zeros((1, 4))
    s1, fell = C.step(s0, np.array([1]))
    assert not fell[0] and s1[0, 0] == 0.0 and s1[0, 2] == 0.0
    temp = 10.0 / 1.1
    thacc = -temp / (0.5 * (4.0 / 3.0 - 0.1 / 1.1))
    assert abs(s1[0, 3] - 0.02 * thacc) < 1e-15
    assert abs(s1[0, 1] - 0.02 * (temp - 0.05 * thacc / 1.1)) < 1e-15
    # a random policy keeps the pole up for roughly 10-100 steps on average (published behaviour: ~22)
    rng = np.random.default_rng(0)
    for t in range(400):
        env.step(rng.integers(0, 2, 256))
    mean_len = env.fin_sum.sum() / env.fin_cnt.sum()
    assert 15.0 < mean_len < 30.0, mean_len
    # time limit
    short = C.VecCartPole(2, seed=1, max_steps=5)
    short.reset()
    dones = [short.step(np.array([t % 2, 1 - t % 2]))[2] for t in range(5)]
    assert not np.any(dones[:4]) and np.all(dones[4])
