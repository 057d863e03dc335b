"""Builds ppo_amd/libppo_hip.so (gfx950) from csrc/*.hip with hipcc.  In-tree, no JIT cache.

Two libraries come out of one set of objects:
  libppo_hip.so      the product
  libppo_hip_sym.so  the same plus the symmetric-wave form of the fused update kernel (update_sym.hip,
                     never the default on any shape; selected with PPO_HIP_LIB=... PPO_FWDBWD=sym).  Kept
                     under test as the alternative form, kept out of the product binary.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppo_hip.so")
LIB_SYM = os.path.join(HERE, "libppo_hip_sym.so")
FASTCALL = os.path.join(HERE, "_fastcall.so")          # CPython extension (csrc/fastcall.c), plain C
SOURCES = ["scan.hip", "act.hip", "act_one.hip", "update.hip", "update_vjp.hip", "env.hip", "rollout.hip"]
SYM_SOURCES = ["update_sym.hip"]             # + update.hip compiled with -DPPO_WITH_SYM
HEADERS = ["ppo_common.h", "update_common.h", os.path.join("..", "..", "include", "ppo_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
         "-fhip-fp32-correctly-rounded-divide-sqrt"]


def _stale(lib, sources):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in sources + HEADERS] + [os.path.abspath(__file__)]   # (update_vjp.hip includes update.hip)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(jobs, verbose):
    """jobs: (source, object, extra flags); all at once (six or seven files)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for src, obj, extra in jobs:
        cmd = [hipcc] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))


def _link(objs, lib, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_fastcall(force=False, verbose=True):
    """The ctypes-free per-step binding (csrc/fastcall.c): a CPython extension compiled with the host C
    compiler against this interpreter's headers; it links against nothing else."""
    import sysconfig
    src = os.path.join(CSRC, "fastcall.c")
    if not force and os.path.exists(FASTCALL) and os.path.getmtime(FASTCALL) >= os.path.getmtime(src):
        return FASTCALL
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-Wall", "-I" + sysconfig.get_paths()["include"],
           src, "-o", FASTCALL]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return FASTCALL


def build(force=False, verbose=True, variant=None, extra_flags=()):
    """Compile every HIP source for gfx950 and link the C-ABI shared libraries.  Returns the product
    library's path; build.last_action says what happened ("compiled" or "up to date").
    variant / extra_flags: diagnostic builds (e.g. -DPPO_TANH_ACCURATE) go to
    libppo_hip_<variant>.so (with the symmetric kernel inside), selected at run time with the
    PPO_HIP_LIB environment variable."""
    if variant is not None:
        objdir = os.path.join(HERE, f"build_{variant}")
        os.makedirs(objdir, exist_ok=True)
        srcs = SOURCES + SYM_SOURCES
        obj = lambda s: os.path.join(objdir, s.replace(".hip", ".o"))
        _compile([(s, obj(s), list(extra_flags) + ["-DPPO_WITH_SYM"]) for s in srcs], verbose)
        lib = os.path.join(HERE, f"libppo_hip_{variant}.so")
        _link([obj(s) for s in srcs], lib, verbose)
        return lib
    build_fastcall(force, verbose)
    if not force and not _stale(LIB, SOURCES) and not _stale(LIB_SYM, SOURCES + SYM_SOURCES):
        build.last_action = "up to date"
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    obj = lambda s: os.path.join(objdir, s.replace(".hip", ".o"))
    upd_sym = os.path.join(objdir, "update_withsym.o")
    _compile([(s, obj(s), []) for s in SOURCES] + [(s, obj(s), []) for s in SYM_SOURCES] +
             [("update.hip", upd_sym, ["-DPPO_WITH_SYM"])], verbose)
    _link([obj(s) for s in SOURCES], LIB, verbose)
    _link([obj(s) for s in SOURCES if s != "update.hip"] + [upd_sym] + [obj(s) for s in SYM_SOURCES],
          LIB_SYM, verbose)
    build.last_action = "compiled"
    return LIB


build.last_action = None


if __name__ == "__main__":
    # python -m ppo_amd.build [--force] [--variant NAME -DFLAG ...]
    args = sys.argv[1:]
    if "--variant" in args:
        i = args.index("--variant")
        build(variant=args[i + 1], extra_flags=[a for a in args[i + 2:] if a.startswith("-")])
    else:
        build(force="--force" in args)
        print("libppo_hip:", build.last_action)
