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

# ---- the same with host work between the calls, as an env's step() would be (the waiting kernel of a
# session is then already polling when the next state arrives)
def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass

for work_us in (10, 30):
    agent = PPO(4, 2, 3e-4, 1e-3, 0.99, 4, 0.2, False, None)
    agent.buffer.reserve(n + 64)
    x = np.random.default_rng(0).standard_normal(4)
    for _ in range(50):
        agent.select_action(x)
    tot = 0.0
    for _ in range(n):
        spin(work_us)
        t0 = time.perf_counter()
        agent.select_action(x)
        tot += time.perf_counter() - t0
    print(f"select_action with {work_us} us of host work between calls: {tot / n * 1e6:.1f} us per call")
    agent._act_drain()

# ---- the host wrapper alone: everything select_action does except the library call
agent = PPO(4, 2, 3e-4, 1e-3, 0.99, 4, 0.2, False, None)
agent.buffer.reserve(n + 64)
x = np.random.default_rng(0).standard_normal(4)
agent.select_action(x)
agent._act_drain()
from ppo_amd import _lib
L = _lib.lib()
real = L.ppo_act_session_step_bound, L.ppo_act_session_bind
L.ppo_act_session_step_bound = lambda *a: 0        # (instance attributes shadow the library's functions)
L.ppo_act_session_bind = lambda *a: 0
agent._host_step = agent._make_session_step()
t0 = time.perf_counter()
for _ in range(n):
    agent.select_action(x)
dt_py = (time.perf_counter() - t0) / n
L.ppo_act_session_step_bound, L.ppo_act_session_bind = real
agent._host_step = agent._make_session_step()
agent.buffer.clear()
print(f"of which the Python wrapper (no library call): {dt_py * 1e6:.1f} us")
