#!/usr/bin/env python3
"""update() latency on the reference-shaped buffers (one flat sequence of a few thousand steps)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import contextlib
import io
import numpy as np
import torch
with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo

for name, S, A, cont, T, K in (("C1 CartPole", 4, 2, False, 1600, 80), ("C2 LunarLander", 8, 4, False, 900, 30),
                               ("C3 BipedalWalker", 24, 4, True, 6000, 80)):
    rng = np.random.default_rng(0)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, K, 0.2, cont, 0.6 if cont else None)
    x = rng.standard_normal((T, S)).astype(np.float32)
    acts = rng.standard_normal((T, A)).astype(np.float32) if cont else rng.integers(0, A, T)
    ts = []
    for rep in range(6):
        agent.buffer.load_rollout(x, acts, -np.ones(T, np.float32), np.zeros(T, np.float32),
                                  [1.0] * T, [bool(i % 200 == 199) for i in range(T)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.update()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    print(f"{name:18s} T={T:5d} K={K:3d}: update() {best * 1e3:7.2f} ms  ({best / K * 1e6:6.1f} us per epoch)")
