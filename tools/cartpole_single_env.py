#!/usr/bin/env python3
"""The reference's own protocol (one env, one select_action per step, 1600-step updates, K = 40:
the CartPole settings of the reference's README) on the device-side CartPole: learning curve + rate."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train_synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tmp = tempfile.mkdtemp()
args = train_synth.parse_args(["--n-envs", "1", "--max-ep-len", "400", "--k-epochs", "40",
                               "--max-training-timesteps", str(steps), "--log-dir", tmp + "/l",
                               "--checkpoint-dir", tmp + "/c", "--random-seed", "1", "--quiet"])
t0 = time.perf_counter()
agent, history, ck, log = train_synth.train(args)
dt = time.perf_counter() - t0
print(f"single env, {steps} steps in {dt:.1f} s = {steps / dt:.0f} env-steps/s (act + env step + appends + updates)")
for ts, avg in history[::4]:
    print(f"   timestep {ts:7d}  average episode return {avg:8.2f}")
