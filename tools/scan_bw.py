"""Achieved bandwidth of the HBM-bound phase kernels (batched return scan, normalise/advantage, flat
scan) at the C4 and C5 buffer sizes: algorithmic bytes (9 and 16 per step) / average time of 20
back-to-back calls between two events on the launch stream."""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
with contextlib.redirect_stdout(io.StringIO()):
    from ppo_amd import _lib
L = _lib.lib()
st = _lib.stream_ptr()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for name, N, T in (("C4", 4096, 128), ("C5", 65536, 128)):
    n = N * T
    r = torch.randn(n, device="cuda")
    t = (torch.rand(n, device="cuda") < 0.005).to(torch.uint8)
    v = torch.randn(n, device="cuda")
    out = torch.empty(n, device="cuda")
    adv = torch.empty(n, device="cuda")
    mom = torch.zeros(3, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.ppo_workspace_bytes(4, 2, n), dtype=torch.uint8, device="cuda")
    s_rows = timed(lambda: _lib.check(L.ppo_returns_scan_rows(r.data_ptr(), t.data_ptr(), T, N, 0.99, out.data_ptr(),
                                                              mom.data_ptr(), ws.data_ptr(), ws.numel(), st)))
    s_flat = timed(lambda: _lib.check(L.ppo_returns_scan_flat(r.data_ptr(), 0, t.data_ptr(), n, 0.99, out.data_ptr(),
                                                              mom.data_ptr(), ws.data_ptr(), ws.numel(), st)))
    s_norm = timed(lambda: _lib.check(L.ppo_normalize_advantage(out.data_ptr(), v.data_ptr(), n, mom.data_ptr(),
                                                                out.data_ptr(), adv.data_ptr(), st)))
    for what, sec, bps in (("scan_rows (+moments_finalize)", s_rows, 9), ("scan_flat (3 launches + moments_finalize; reads twice: 14 B/step moved)", s_flat, 9),
                           ("normalize_advantage", s_norm, 16)):
        gbs = bps * n / sec / 1e9
        print(f"{name} {n:9d} steps  {what:75s} {sec * 1e6:8.1f} us  {gbs:7.0f} GB/s algorithmic  {gbs / 8000:.3f} of 8 TB/s")
