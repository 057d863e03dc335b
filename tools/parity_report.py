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
from helpers import UPDATE_CASES, case_params, load_case, max_abs, rel_err
from oracle import ppo_oracle as O

with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo
from test_gpu_parity import agent_from_case

print(f"{'case':26s} {'K':>3s} {'ret_norm':>9s} {'adv':>9s} {'logp':>9s} {'value':>9s} {'entropy':>9s} "
      f"{'loss':>9s} {'grad':>9s} {'param':>9s}")
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
        if i == 0:
            g = agent.debug_outputs
            grads = O.unflatten_params(g["grad"].cpu().numpy(), int(d["S"]), int(d["A"]))
            ge = max(rel_err(grads[k], d["grad_e1/" + k], floor=float(np.abs(d["grad_e1/" + k]).max()))
                     for k in O.PARAM_KEYS)
            print(f"{name:26s} {K:3d} {rel_err(g['returns_norm'].cpu().numpy(), d['returns_norm']):9.2e} "
                  f"{rel_err(g['advantages'].cpu().numpy(), d['advantages']):9.2e} "
                  f"{rel_err(g['logprobs'].cpu().numpy(), d['logprobs_e1']):9.2e} "
                  f"{rel_err(g['state_values'].cpu().numpy(), d['state_values_e1']):9.2e} "
                  f"{rel_err(g['entropy'].cpu().numpy(), d['entropy_e1']):9.2e} {le:9.2e} {ge:9.2e} {pe:9.2e}")
        else:
            print(f"{name:26s} {K:3d} {'':9s} {'':9s} {'':9s} {'':9s} {'':9s} {le:9.2e} {'':9s} {pe:9.2e}")
