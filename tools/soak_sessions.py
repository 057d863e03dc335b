#!/usr/bin/env python3
"""Soak test of the single-state act sessions: many steps with irregular pauses (waiting kernels
give up, buffer storage moves), then every stored row is checked against ppo_evaluate of the stored
(state, action) - log-prob and value must match bit for bit - and the stored states against the
inputs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PPO import PPO

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(0)
for cont, S, A in ((False, 8, 4), (True, 17, 6)):
    agent = PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
    xs = rng.standard_normal((n, S)).astype(np.float32)
    pauses = set(rng.integers(0, n, size=60).tolist())
    t0 = time.perf_counter()
    acts = []
    for i in range(n):
        if i in pauses:
            time.sleep(0.0025)
        acts.append(agent.select_action(xs[i]))
    dt = time.perf_counter() - t0
    agent._act_drain()
    b = agent.buffer
    st, ac = b.rows("states", 0, n), b.rows("actions", 0, n)
    lp, val, _ = agent.policy_old.evaluate(st, ac)
    ok_s = torch.equal(st.cpu(), torch.from_numpy(xs))
    ok_a = np.array_equal(np.asarray(acts, dtype=np.float32 if cont else np.int64).reshape(n, -1),
                          ac.cpu().numpy().reshape(n, -1))
    ok_l = torch.equal(lp, b.rows("logprobs", 0, n))
    ok_v = torch.equal(val.reshape(-1), b.rows("state_values", 0, n))
    print(f"{'continuous' if cont else 'discrete'} S={S} A={A}: {n} steps in {dt:.2f} s "
          f"({dt / n * 1e6:.1f} us/step incl. {len(pauses)} pauses): states {ok_s}, actions {ok_a}, "
          f"logprobs {ok_l}, values {ok_v}")
    assert ok_s and ok_a and ok_l and ok_v
print("soak ok")
