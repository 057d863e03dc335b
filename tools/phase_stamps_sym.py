#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of the symmetric-wave fwd/bwd kernel (update_sym.hip) from a
-DPPO_PHASE_STAMPS build (python -m ppo_amd.build --variant stamps -DPPO_PHASE_STAMPS).  A wave's
phase time includes the cycles in which its SIMD partner held the issue slot."""
import contextlib
import ctypes
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "ppo_amd", "libppo_hip_stamps.so")
os.environ["PPO_HIP_LIB"] = LIB
import torch
from bench import CONFIGS

with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import _lib, ppo
cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c4"])
S, A, N, T = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"]
torch.manual_seed(0)
agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 3, 0.2, cfg["continuous"], cfg["std"])
agent.buffer.reserve(N * T)
agent.synthetic_rollout(N, T, cfg["reward_kind"], cfg["p_terminal"])
stamps = torch.zeros(2048 * 16, dtype=torch.int64, device="cuda")
ctypes.CDLL(LIB).ppo_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
agent.kernel_events = []
agent.update()
torch.cuda.synchronize()
print("ms per launch:", [round(a.elapsed_time(b), 4) for a, b in agent.kernel_events])
st = stamps.view(256, 8, 16).cpu().numpy().astype(float)
names = ["X tile to LDS, prefetch, head inputs", "layer 1 + tanh", "layer 2 + tanh", "output layer + head + loss",
         "W3 wgrad, dZ2, staging writes", "dgrad + dZ1", "wgrad 2", "dZ1 staging + wgrad 1"]
mfma = [0, 8, 64, 0, 0, 64, 64, 16]       # 16x16x4 MFMAs per phase (S + 1 <= 16)
nz = st[:, :, :8].sum(2) > 0
ws = int(st[:, 0, 1].nonzero()[0].size)
for role in ("actor", "critic"):
    sel = [g for g in range(256) if nz[g].any()]
    # actor workgroups come first; find the split from the head phase cost
    break
tot = st[:, :, :8].sum(2)
print("workgroups with stamps:", int(nz.any(1).sum()))
for half, wsl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    rows = st[nz.any(1)][:, wsl, :8].reshape(-1, 8)
    n_tiles = (N * T / 16) * 2 / (rows.shape[0] * 2)
    print(f"{half}: mean cycles per wave {rows.sum(1).mean():.0f}, {rows.sum(1).mean() / n_tiles:.0f} per tile ({n_tiles:.1f} tiles per wave)")
    for i, nme in enumerate(names):
        c = rows[:, i].mean() / n_tiles
        print(f"   {nme:40s} {c:8.0f} cycles/tile  {100 * rows[:, i].mean() / rows.sum(1).mean():5.1f}%   MFMA issue {mfma[i] * 32:5d}")
per_wg = tot.max(1)
edges = [0, 64, 100, 120, 130, 138, 146, 160, 200, 256]
print("max over waves of loop cycles, mean over workgroup ranges:")
for a_, b_ in zip(edges[:-1], edges[1:]):
    print(f"   wg {a_:3d}-{b_ - 1:3d}: {per_wg[a_:b_].mean():9.0f}   (waves 0-3 {tot[a_:b_, :4].mean():9.0f}, waves 4-7 {tot[a_:b_, 4:].mean():9.0f})")
print(f"prologue (image fill, first prefetch): {st[:, :, 8].mean():.0f} cycles; epilogue (wait for the workgroup's other waves + reduction + partial row): waves 0-3 {st[:, :4, 9].mean():.0f}, waves 4-7 {st[:, 4:, 9].mean():.0f}")
