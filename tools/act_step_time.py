#!/usr/bin/env python3
"""Device time of one batched ppo_act step (launches back to back) for a few batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ppo, _lib
for S, A, cont in ((4, 2, False), (26, 6, True)):
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
    for n in (64, 512, 4096, 8192, 16384, 65536):
        x = torch.randn(n, S, device="cuda")
        out = (torch.empty((n, A) if cont else (n,), dtype=torch.float32 if cont else torch.int32, device="cuda"),
               torch.empty(n, device="cuda"), torch.empty(n, device="cuda"))
        for _ in range(5):
            agent.policy_old.act(x, _seed=1, _offset=0, _out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            agent.policy_old.act(x, _seed=1, _offset=i, _out=out)
        e1.record()
        torch.cuda.synchronize()
        print(f"S={S} A={A} n={n}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per act step (tile {os.environ.get('PPO_ACT_TILE', 'auto')})")
