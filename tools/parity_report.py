#!/usr/bin/env python3
"""Prints the measured GPU-vs-reference errors for every golden case (needs a GPU)."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import (TWO_UPDATE_CASES, UPDATE_CASES, case_params, load_branch_state, load_case, max_abs, nearest_branch, rel_err)
from oracle import ppo_oracle as O

with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo
from test_gpu_parity import agent_from_case

print("GPU engine against the reference's golden vectors (tests/golden/).  param: max |param - golden run|;")
print("for K > 4 also the nearest of the reference's own recorded end points (branch 0 = the golden run's) and the")
print("distance to it; m_err / v_err: exp_avg / exp_avg_sq against the reference optimizer's state (the golden run's")
print("for K <= 4, the nearest end point's for K > 4), largest error per tensor relative to that tensor's largest entry.")
print(f"{'case':26s} {'K':>3s} {'ret_norm':>9s} {'adv':>9s} {'logp':>9s} {'value':>9s} {'entropy':>9s} "
      f"{'loss':>9s} {'grad':>9s} {'param':>9s} {'branch':>6s} {'to_branch':>9s} {'m_err':>9s} {'v_err':>9s}")


def adam_errors(agent, want_m, want_v, S, A):
    """largest per-tensor error of exp_avg / exp_avg_sq relative to the tensor's largest entry"""
    m = O.unflatten_params(agent.optimizer.exp_avg.cpu().numpy(), S, A)
    v = O.unflatten_params(agent.optimizer.exp_avg_sq.cpu().numpy(), S, A)
    em = max(max_abs(m[k], want_m[k]) / max(float(np.abs(want_m[k]).max()), 1e-30) for k in O.PARAM_KEYS)
    ev = max(max_abs(v[k], want_v[k]) / max(float(np.abs(want_v[k]).max()), 1e-30) for k in O.PARAM_KEYS)
    return em, ev


for name in UPDATE_CASES:
    d = load_case(name)
    for i, K in enumerate(d["K_list"]):
        K = int(K)
        with contextlib.redirect_stdout(io.StringIO()):
            agent = agent_from_case(ppo, d, K)
        agent.update(debug=(i == 0))
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
        pe = max(max_abs(got[k], d[f"K{K}/param/{k}"]) for k in O.PARAM_KEYS)
        le = rel_err(agent.last_loss.cpu().numpy(), d[f"K{K}/loss_mean_all"])
        S_, A_ = int(d["S"]), int(d["A"])
        branch, to_branch = "", ""
        if K > 4 and os.path.exists(os.path.join(ROOT, "tests", "golden", name + "_branches.npz")):
            dist, idx = nearest_branch(agent.policy.flat.cpu().numpy(), name)
            bm, bv = load_branch_state(name, idx)
            em, ev = adam_errors(agent, O.unflatten_params(bm, S_, A_), O.unflatten_params(bv, S_, A_), S_, A_)
            branch, to_branch = str(idx), f"{dist:9.2e}"
        else:
            em, ev = adam_errors(agent, {k: d[f"K{K}/exp_avg/{k}"] for k in O.PARAM_KEYS},
                                 {k: d[f"K{K}/exp_avg_sq/{k}"] for k in O.PARAM_KEYS}, S_, A_)
        tail = f" {branch:>6s} {to_branch:>9s} {em:9.2e} {ev:9.2e}"
        if i == 0:
            g = agent.debug_outputs
            grads = O.unflatten_params(g["grad"].cpu().numpy(), int(d["S"]), int(d["A"]))
            ge = max(rel_err(grads[k], d["grad_e1/" + k], floor=float(np.abs(d["grad_e1/" + k]).max()))
                     for k in O.PARAM_KEYS)
            print(f"{name:26s} {K:3d} {rel_err(g['returns_norm'].cpu().numpy(), d['returns_norm']):9.2e} "
                  f"{rel_err(g['advantages'].cpu().numpy(), d['advantages']):9.2e} "
                  f"{rel_err(g['logprobs'].cpu().numpy(), d['logprobs_e1']):9.2e} "
                  f"{rel_err(g['state_values'].cpu().numpy(), d['state_values_e1']):9.2e} "
                  f"{rel_err(g['entropy'].cpu().numpy(), d['entropy_e1']):9.2e} {le:9.2e} {ge:9.2e} {pe:9.2e}" + tail)
        else:
            print(f"{name:26s} {K:3d} {'':9s} {'':9s} {'':9s} {'':9s} {'':9s} {le:9.2e} {'':9s} {pe:9.2e}" + tail)


print()
print("two consecutive update() calls on one agent (Adam's step count and moments persist): after update u")
print(f"{'case':26s} {'u':>3s} {'param':>9s} {'m_err':>9s} {'v_err':>9s} {'loss':>9s}")
for name in TWO_UPDATE_CASES:
    d = load_case(name)
    cont, K, S_, A_ = bool(d["continuous"]), int(d["K"]), int(d["S"]), int(d["A"])
    with contextlib.redirect_stdout(io.StringIO()):
        agent = ppo.PPO(S_, A_, float(d["lr_actor"]), float(d["lr_critic"]), float(d["gamma"]), K,
                        float(d["eps_clip"]), cont, float(d["action_std"]) if cont else None)
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    agent.policy.load_state_dict(sd)
    agent.policy_old.load_state_dict(sd)
    for u in (1, 2):
        with contextlib.redirect_stdout(io.StringIO()):
            if cont and float(d[f"u{u}/action_std"]) != agent.action_std:
                agent.decay_action_std(0.05, 0.1)
        agent.buffer.load_rollout(d[f"u{u}/states"], d[f"u{u}/actions"], d[f"u{u}/old_logprobs"],
                                  d[f"u{u}/old_values"], d[f"u{u}/rewards"], d[f"u{u}/terminals"])
        agent.update()
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
        pe = max(max_abs(got[k], d[f"u{u}/after/param/{k}"]) for k in O.PARAM_KEYS)
        em, ev = adam_errors(agent, {k: d[f"u{u}/after/exp_avg/{k}"] for k in O.PARAM_KEYS},
                             {k: d[f"u{u}/after/exp_avg_sq/{k}"] for k in O.PARAM_KEYS}, S_, A_)
        le = rel_err(agent.last_loss.cpu().numpy(), d[f"u{u}/loss_mean_all"])
        print(f"{name:26s} {u:3d} {pe:9.2e} {em:9.2e} {ev:9.2e} {le:9.2e}")
