"""ctypes binding of libppo_hip.so (the C ABI declared in include/ppo_hip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint, c_uint64, c_void_p

import torch  # noqa: F401  (loads the HIP runtime the library binds to before dlopen)

HERE = os.path.dirname(os.path.abspath(__file__))
# PPO_HIP_LIB: diagnostic builds of the same sources (python -m ppo_amd.build --variant ...)
LIB_PATH = os.environ.get("PPO_HIP_LIB") or os.path.join(HERE, "libppo_hip.so")
ABI_VERSION = 10

_SIGNATURES = {
    "ppo_hip_abi_version": (c_int, []),
    "ppo_hip_strerror": (c_char_p, [c_int]),
    "ppo_param_count": (c_int64, [c_int, c_int, ctypes.POINTER(c_int64)]),
    "ppo_workspace_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "ppo_act": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_uint64,
                        c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_evaluate": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int64,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_evaluate_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ppo_returns_scan_flat": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_double, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "ppo_returns_scan_flat_noalloc": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_double, c_void_p,
                                              c_void_p, c_void_p]),
    "ppo_returns_scan_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "ppo_normalize_advantage": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "ppo_epoch_fwdbwd": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ppo_grad_reduce": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ppo_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                              c_double, c_double, c_double, c_double, c_double, c_void_p,
                              c_void_p]),
    "ppo_synth_env_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float,
                                   c_int64, c_int64, c_uint64, c_void_p]),
    "ppo_act_host": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_uint64, c_uint64,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_void_p]),
    "ppo_kernel_timings_enable": (c_int, [c_int]),
    "ppo_kernel_timings": (c_int, [c_void_p, c_int]),
    "ppo_act_session_create": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "ppo_act_session_step": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_void_p, c_uint64, c_uint64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_act_session_bind": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_void_p, c_void_p, c_uint64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_act_session_step_bound": (c_int, [c_void_p, c_int64, c_uint64]),
    "ppo_act_session_drain": (c_int, [c_void_p]),
    "ppo_act_session_destroy": (c_int, [c_void_p]),
    "ppo_cartpole_reset": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_uint64, c_void_p]),
    "ppo_cartpole_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_int, c_uint64, c_void_p]),
    "ppo_cartpole_step_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p, ctypes.c_uint, c_int, c_uint64, c_void_p]),
    "ppo_update_epochs": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_double, c_double, c_double, c_double, c_double, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "ppo_cartpole_rollout": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int64, c_int, c_uint64, c_uint64, c_uint64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_synth_rollout": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int64, c_int64, c_int,
                                  c_float, c_int64, c_int64, c_uint64, c_uint64, c_uint64, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppo_reduce_adam_step": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_double, c_double, c_double, c_double,
                                     c_double, c_void_p, c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def lib():
    """The loaded library (dlopen on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m ppo_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        got = handle.ppo_hip_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libppo_hip.so ABI {got} != binding {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise RuntimeError(f"libppo_hip call failed ({code}): "
                           f"{lib().ppo_hip_strerror(code).decode()}")


def ptr(t):
    """Device (or host) address of a tensor, or NULL for None."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
