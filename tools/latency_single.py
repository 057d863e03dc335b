#!/usr/bin/env python3
"""Latency of the reference's single-env protocol (one select_action per step) on the engine."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PPO import PPO
from ppo_amd import envs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, cont, S, A in (("discrete S=4 A=2", False, 4, 2), ("continuous S=24 A=4", True, 24, 4)):
    agent = PPO(S, A, 3e-4, 1e-3, 0.99, 4, 0.2, cont, 0.6 if cont else None)
    agent.buffer.reserve(n + 64)
    x = np.random.default_rng(0).standard_normal(S)
    for _ in range(50):
        agent.select_action(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        agent.select_action(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"select_action, {name}: {dt * 1e6:.1f} us per call ({1 / dt:.0f} steps/s)")
    agent.buffer.clear()
env = envs.make("CartPole-v1")
agent = PPO(4, 2, 3e-4, 1e-3, 0.99, 4, 0.2, False, None)
agent.buffer.reserve(n + 64)
s = env.reset()
t0 = time.perf_counter()
for _ in range(n):
    a = agent.select_action(s)
    s, r, d, _ = env.step(a)
    agent.buffer.rewards.append(r)
    agent.buffer.is_terminals.append(d)
    if d:
        s = env.reset()
dt = (time.perf_counter() - t0) / n
print(f"full step (select_action + device CartPole step + appends): {dt * 1e6:.1f} us ({1 / dt:.0f} steps/s)")

# ---- where the time of one select_action goes (discrete S=4 A=2)
from ppo_amd import _lib
agent = PPO(4, 2, 3e-4, 1e-3, 0.99, 4, 0.2, False, None)
agent.buffer.reserve(n + 64)
x = np.random.default_rng(0).standard_normal(4)
agent.select_action(x)
real = agent._act_host
agent._act_host = lambda *a: 0
t0 = time.perf_counter()
for _ in range(n):
    agent.select_action(x)
dt_py = (time.perf_counter() - t0) / n
agent._act_host = real
agent.buffer.clear()
b = agent.buffer
lo = b.claim_step(n)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rows = [b.rows(c, lo, lo + n) for c in ("states", "actions", "logprobs", "state_values")]
L = _lib.lib()
torch.cuda.synchronize()
e0.record()
for i in range(n):
    L.ppo_act(agent.policy_old.flat.data_ptr(), 4, 2, 0, 0.0, rows[0].data_ptr(), 1, 1, i,
              rows[1].data_ptr(), rows[2].data_ptr(), rows[3].data_ptr(), st)
e1.record()
torch.cuda.synchronize()
print(f"breakdown: python wrapper without the library call {dt_py * 1e6:.1f} us; "
      f"act kernel back to back (n = 1) {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch")
