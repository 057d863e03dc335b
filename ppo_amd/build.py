"""Builds ppo_amd/libppo_hip.so (gfx950) from csrc/*.hip with hipcc.  In-tree, no JIT cache."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppo_hip.so")
SOURCES = ["scan.hip", "act.hip", "act_one.hip", "update.hip", "env.hip", "rollout.hip", "update_sym.hip"]
HEADERS = ["ppo_common.h", "update_common.h", os.path.join("..", "..", "include", "ppo_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
         "-fhip-fp32-correctly-rounded-divide-sqrt"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, extra_flags=()):
    """Compile every HIP source for gfx950 and link the C-ABI shared library.
    variant / extra_flags: diagnostic builds (e.g. -DPPO_TANH_ACCURATE) go to
    libppo_hip_<variant>.so, selected at run time with the PPO_HIP_LIB environment variable."""
    lib = LIB if variant is None else os.path.join(HERE, f"libppo_hip_{variant}.so")
    if variant is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" if variant is None else f"build_{variant}")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), obj, cmd))
    objs = []
    for p, obj, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    # python -m ppo_amd.build [--force] [--variant NAME -DFLAG ...]
    args = sys.argv[1:]
    if "--variant" in args:
        i = args.index("--variant")
        build(variant=args[i + 1], extra_flags=[a for a in args[i + 2:] if a.startswith("-")])
    else:
        build(force="--force" in args)
