"""Diagnostic: run update() on a golden case for several K and save the engine's parameter vectors
under gpurun_out/ for offline comparison with the reference's own run-to-run spread."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import case_params, load_case  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo

name = sys.argv[1]
Ks = [int(k) for k in sys.argv[2:]]
d = load_case(name)
cont = bool(d["continuous"])
out = {}
for K in Ks:
    agent = ppo.PPO(int(d["S"]), int(d["A"]), float(d["lr_actor"]), float(d["lr_critic"]), float(d["gamma"]), K,
                    float(d["eps_clip"]), cont, float(d["action_std"]) if cont else None)
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    agent.policy.load_state_dict(sd)
    agent.policy_old.load_state_dict(sd)
    agent.buffer.load_rollout(d["states"], d["actions"], d["old_logprobs"], d["old_values"], d["rewards"],
                              d["terminals"])
    agent.update()
    torch.cuda.synchronize()
    out[f"K{K}"] = agent.policy.flat.cpu().numpy()
    out[f"K{K}/loss"] = agent.last_loss.cpu().numpy()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"dump_{name}{os.environ.get('PPO_DUMP_TAG', '')}.npz"), **out)
print("saved", name, Ks)
