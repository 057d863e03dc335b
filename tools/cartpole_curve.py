#!/usr/bin/env python3
"""Learning curve + throughput of the vectorised driver on the device-side CartPole-v1."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import train_synth

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 12
extra = sys.argv[3:]
tmp = tempfile.mkdtemp()
args = train_synth.parse_args(["--n-envs", str(n_envs), "--horizon", "128", "--k-epochs", "80",
                               "--max-training-timesteps", str(n_envs * 128 * updates - 1),
                               "--log-dir", tmp + "/l", "--checkpoint-dir", tmp + "/c", "--random-seed", "1",
                               "--save-model-freq", str(10 ** 12), "--quiet"] + extra)
torch.cuda.synchronize()
t0 = time.perf_counter()
agent, history, ck, log = train_synth.train(args)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = n_envs * 128 * updates
print(f"n_envs={n_envs} updates={updates} {' '.join(extra)}: {steps} env-steps in {dt:.2f} s = {steps / dt:.3e} env-steps/s")
for ts, avg in history:
    print(f"   timestep {ts:9d}  average episode return {avg:8.2f}")
