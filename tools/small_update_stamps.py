#!/usr/bin/env python3
"""Diagnostic: where one launch of the fwd/bwd kernel spends its time on a SMALL buffer (one tile per
wave pair), from wall-clock marks inside the kernel (a -DPPO_PHASE_STAMPS build: tools/phase_stamps.py
--build).  usage: python3 tools/small_update_stamps.py [c1|c2|c3]"""
import contextlib
import ctypes
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import CONFIGS
from ppo_amd import _lib

LIB = os.environ.get("PPO_STAMPS_LIB", os.path.join(ROOT, "ppo_amd", "libppo_hip_stamps.so"))
_lib.LIB_PATH = LIB
with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo
cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c1"])
S, A, T = cfg["S"], cfg["A"], cfg["horizon"]
torch.manual_seed(0)
agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 4, 0.2, cfg["continuous"], cfg["std"])
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((T, S)).astype(np.float32)).cuda()
agent.select_action(x)
for t in range(T):
    agent.buffer.rewards.append(1.0)
    agent.buffer.is_terminals.append(bool(t % 200 == 199))
stamps = torch.zeros(2048 * 16, dtype=torch.int64, device="cuda")
ctypes.CDLL(LIB).ppo_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
agent.update()
torch.cuda.synchronize()
st = stamps.view(256, 8, 16).cpu().numpy().astype(np.int64)      # the LAST epoch's marks
used = st[:, 0, 12] != 0
m = st[used][:, :, 12:16].astype(float)                             # [wg, wave, mark]
t0 = m[:, :, 0].min()
us = (m - t0) / 100.0
print(f"{int(used.sum())} workgroups; microseconds since the first wave's entry (mean over waves; min .. max):")
for i, name in enumerate(["entry", "weights staged (after the barrier)", "tile loop + in-workgroup reduction done",
                          "partial row written"]):
    print(f"   {name:44s} {us[:, :, i].mean():7.2f}   {us[:, :, i].min():7.2f} .. {us[:, :, i].max():7.2f}")
fw, bw = us[:, :4], us[:, 4:]
print(f"   forward waves leave the wave function at  {fw[:, :, 2].mean():7.2f}, backward waves at {bw[:, :, 2].mean():7.2f}")
cyc = st[used][:, :, :12].astype(float)
print("per-phase cycles (s_memtime), forward waves:", np.round(cyc[:, :4].mean((0, 1))[:7]).tolist())
print("per-phase cycles (s_memtime), backward waves:", np.round(cyc[:, 4:].mean((0, 1))[:8]).tolist())
