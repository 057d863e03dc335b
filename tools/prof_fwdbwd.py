#!/usr/bin/env python3
"""Runs one iteration of the default product path (fused rollout + update()) on a synthetic C4/C5-shaped
buffer (for rocprofv3 passes).
usage: python3 tools/prof_fwdbwd.py [c4|c5] [epochs]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import CONFIGS

with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import _lib, ppo

cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c4"])
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if len(sys.argv) > 3:
    cfg["n_envs"] = int(sys.argv[3])
S, A, N, T = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"]
torch.manual_seed(0)
agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, epochs, 0.2, cfg["continuous"], cfg["std"])
agent.buffer.reserve(N * T)
# the product path: one persistent rollout kernel, then update() through ppo_update_epochs
agent.synthetic_rollout(N, T, cfg["reward_kind"], cfg["p_terminal"])
agent.update()
torch.cuda.synchronize()
print("loss", agent.last_loss.tolist()[-1])
