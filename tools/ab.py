#!/usr/bin/env python3
"""Same-box A/B of library builds: loads every given build of libppo_hip into ONE process, puts the
same synthetic rollout into an agent per build, and times update() of the builds in turn (interleaved
rounds, default product path), so that device-to-device and clock differences cancel.

  python tools/ab.py [--cfg c4] [--rounds 8] [--k 20] name=path/to/lib.so [name=path ...]

Prints per build: median ms per epoch of update() (wall, events around the whole call) and the median
ms per fwd/bwd launch (events inside the library around every launch, in a separate call)."""
import contextlib
import ctypes
import io
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import CONFIGS


def main():
    args = sys.argv[1:]
    cfg_name, rounds, k = "c4", 8, 20
    libs = []
    i = 0
    while i < len(args):
        if args[i] == "--cfg":
            cfg_name = args[i + 1]; i += 2
        elif args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--k":
            k = int(args[i + 1]); i += 2
        else:
            name, path = args[i].split("=", 1)
            libs.append((name, os.path.abspath(path))); i += 1
    if cfg_name.startswith("wide"):     # wideS_A: a 524 288-sample buffer with S-wide observations (not a BASELINE config)
        S_, A_ = (int(v) for v in cfg_name[4:].split("_"))
        cfg = dict(CONFIGS["c4"], S=S_, A=A_)
    else:
        cfg = dict(CONFIGS[cfg_name])
    S, A, N, T = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"]
    from ppo_amd import _lib
    agents = []
    for name, path in libs:
        _lib.LIB_PATH = path
        _lib._lib = None
        with contextlib.redirect_stdout(io.StringIO()):
            from ppo_amd import ppo
        handle = _lib.lib()
        torch.manual_seed(0)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, k, 0.2, cfg["continuous"], cfg["std"])
        agent.buffer.reserve(N * T)
        agents.append((name, handle, agent))

    def one(name, handle, agent, timed):
        _lib._lib = handle          # the agent's calls go to this build
        agent.synthetic_rollout(N, T, cfg["reward_kind"], cfg["p_terminal"])
        handle.ppo_kernel_timings_enable(1 if timed else 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        agent.update()
        e1.record()
        torch.cuda.synchronize()
        out = (ctypes.c_float * 3)()
        if timed:
            _lib.check(handle.ppo_kernel_timings(out, 3))
            handle.ppo_kernel_timings_enable(0)
        return e0.elapsed_time(e1) / k, out[0], out[1]

    res = {name: ([], [], []) for name, _, _ in agents}
    for name, handle, agent in agents:      # warm-up
        one(name, handle, agent, False)
    for r in range(rounds):
        for name, handle, agent in (agents if r % 2 == 0 else agents[::-1]):
            res[name][0].append(one(name, handle, agent, False)[0])
            _, fb, ra = one(name, handle, agent, True)
            res[name][1].append(fb)
            res[name][2].append(ra)
    base = None
    for name, _, _ in agents:
        ep, fb, ra = (statistics.median(v) for v in res[name])
        base = base or ep
        print(f"{cfg_name} {name:16s} update {ep * 1e3:8.2f} us/epoch ({ep / base:6.4f} of first)   "
              f"fwdbwd {fb * 1e3:8.2f} us   reduce+adam {ra * 1e3:6.2f} us   "
              f"[min {min(res[name][0]) * 1e3:.2f} max {max(res[name][0]) * 1e3:.2f}]", flush=True)


if __name__ == "__main__":
    main()
