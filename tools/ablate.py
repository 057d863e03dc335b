#!/usr/bin/env python3
"""Diagnostic: timing ablations of the fwd/bwd kernel.  Builds variants of the library from PATCHED
COPIES of the sources (in a scratch directory; the product sources are not touched) in which one
part of the kernel is switched off, and times one epoch with each.  Results are WRONG numbers by
construction - only the launch durations mean anything.

  python tools/ablate.py --build        # here (hipcc cross-compiles), variants land in ppo_amd/abl/
  python tools/ablate.py [c4|c5]        # on the GPU box
"""
import ctypes
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "ppo_amd", "abl")

# name -> list of (file, regex, replacement)
VARIANTS = {
    "base": [],
    "no_barrier": [("update.hip", r"__syncthreads\(\);  // the partner has finished[^\n]*", ""),
                   ("update.hip", r"__syncthreads\(\);  // buffers of this tile are complete", ""),
                   ("update.hip", r"__syncthreads\(\);  // \(forward wave: previous buffers may be overwritten\)", "")],
    "fwd_only": [("update.hip", r"(        STAMP\(2\)\n\n        // -+ dgrad of layer 2)", r"        continue;\n\1")],
    "bwd_only": [("update.hip", r"(        STAMP_START\n\n        // -+ layer 1 \(bias rides)",
                  r"        __syncthreads(); __syncthreads(); continue;\n\1")],
    "no_tanh": [("ppo_common.h", r"    const float r = __builtin_amdgcn_rcpf\(__builtin_amdgcn_exp2f\(xs\) \+ 1\.0f\);\n    return fmaf\(-2\.0f, r, 1\.0f\);\n#endif",
                 "    return xs * 0.01f;\n#endif")],
}


def build():
    from ppo_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    variants = dict(VARIANTS)
    if "--stamps" in sys.argv:      # the two single-role variants with per-phase cycle stamps
        variants = {k + "_stamps": v for k, v in VARIANTS.items() if k in ("fwd_only", "bwd_only")}
    for name, patches in variants.items():
        work = os.path.join("/tmp", "ppo_abl", name)
        shutil.rmtree(work, ignore_errors=True)
        shutil.copytree(B.CSRC, os.path.join(work, "ppo_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
        for fname, pat, rep in patches:
            p = os.path.join(work, "ppo_amd", "csrc", fname)
            src = open(p).read()
            new, n = re.subn(pat, rep, src)
            if n == 0:
                raise SystemExit(f"{name}: pattern not found in {fname}: {pat[:60]}")
            open(p, "w").write(new)
        srcs = [os.path.join(work, "ppo_amd", "csrc", s) for s in B.SOURCES]
        lib = os.path.join(OUT, f"libppo_hip_{name}.so")
        extra = ["-DPPO_PHASE_STAMPS"] if name.endswith("_stamps") else []
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + extra + ["-shared"] + srcs + ["-o", lib])
        print("built", lib, flush=True)


def run(cfg_name):
    import contextlib
    import io
    import torch
    from bench import CONFIGS
    from ppo_amd import _lib
    cfg = dict(CONFIGS[cfg_name])
    S, A, N, T = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"]
    base = None
    for name in VARIANTS:
        path = os.path.join(OUT, f"libppo_hip_{name}.so")
        if not os.path.exists(path):
            continue
        _lib.LIB_PATH = path
        _lib._lib = None
        with contextlib.redirect_stdout(io.StringIO()):
            from ppo_amd import ppo
        torch.manual_seed(0)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 6, 0.2, cfg["continuous"], cfg["std"])
        agent.buffer.reserve(N * T)
        agent.synthetic_rollout(N, T, cfg["reward_kind"], cfg["p_terminal"])
        agent.kernel_events = []
        agent.update()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in agent.kernel_events)[len(agent.kernel_events) // 2]
        base = base or ms
        print(f"{cfg_name} {name:12s} {ms:8.4f} ms/epoch  ({ms / base:5.2f} of base)", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        run(sys.argv[1] if len(sys.argv) > 1 else "c4")
