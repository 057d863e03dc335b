#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fwd/bwd kernel (builds a -DPPO_PHASE_STAMPS library
next to the product one; the stamps serialise the schedule, so read SHARES, not totals)."""
import contextlib
import ctypes
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CONFIGS
from ppo_amd import build as B

LIB = os.environ.get("PPO_STAMPS_LIB", os.path.join(ROOT, "ppo_amd", "libppo_hip_stamps.so"))
if "--build" in sys.argv:
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES + B.SYM_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DPPO_PHASE_STAMPS", "-DPPO_WITH_SYM", "-shared"] + srcs + ["-o", LIB])
    sys.exit(0)

from ppo_amd import _lib
_lib.LIB_PATH = LIB
with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import ppo
cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c4"])
S, A, N, T = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"]
L = _lib.lib()
torch.manual_seed(0)
agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 3, 0.2, cfg["continuous"], cfg["std"])
agent.buffer.reserve(N * T)
for t in range(T):
    xs = agent.next_state_slot(N)
    rs, ds = agent.buffer.claim_env_slots(N)
    _lib.check(L.ppo_synth_env_step(xs.data_ptr(), rs.data_ptr(), ds.data_ptr(), N, S, cfg["reward_kind"],
                                    cfg["p_terminal"], t, T, 1234, _lib.stream_ptr()))
    agent.act_inplace(N)
stamps = torch.zeros(2048 * 16, dtype=torch.int64, device="cuda")
ctypes.CDLL(LIB).ppo_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
agent.kernel_events = []
agent.update()
torch.cuda.synchronize()
print("ms per launch:", [round(a.elapsed_time(b), 4) for a, b in agent.kernel_events])
st = stamps.view(256, 8, 16).cpu().numpy().astype(float)
fwd_names = ["L1 + tanh", "L2 + tanh", "out layer + head + L3 wgrad", "WAIT barrier (partner done)",
             "dZ2 + hand-off writes", "WAIT barrier (written)", "epilogue"]
bwd_names = ["x loads issue", "WAIT barrier (release)", "deferred wgrad1 + WAIT barrier (written)", "dgrad",
             "dZ1", "wgrad2", "dZ1 store (+ wgrad1 if S >= 32)", "epilogue"]
def actor_wgs(A, n):                                  # actor_wgs() of update.hip
    r = 1.0 + (0.051 if A > 4 else (0.062 if A > 2 else 0.06)) * A
    tiles = (n + 31) // 32
    best, best_cost = 128, 1e300
    for na in range(64, 193):
        ia, ic = -(-tiles // (4 * na)), -(-tiles // (4 * (256 - na)))
        cost = max(ia * r, ic) * (1.0 + 1e-6 * abs(na - 256 * r / (1.0 + r)))
        if cost < best_cost:
            best, best_cost = na, cost
    return best


n_actor_wg = actor_wgs(A, N * T)
for role, sel in (("actor", slice(0, n_actor_wg)), ("critic", slice(n_actor_wg, 256))):
    for kind, ws, names in (("forward wave", slice(0, 4), fwd_names), ("backward wave", slice(4, 8), bwd_names)):
        rows = st[sel][:, ws].reshape(-1, 16)
        tot = rows[:, :len(names)].sum(1).mean()
        n_wg = n_actor_wg if role == "actor" else 256 - n_actor_wg
        tiles = (N * T / 32) / (n_wg * 4)
        print(f"{role} {kind}: mean cycles/wave {tot:.0f} ({tot / tiles:.0f} per tile, {tiles:.1f} tiles/wave)")
        for i, nme in enumerate(names):
            print(f"   {nme:42s} {rows[:, i].mean():10.0f}  {100 * rows[:, i].mean() / tot:5.1f}%  {rows[:, i].mean() / tiles:7.0f}/tile")
