"""The allocation-free flat return scan (static device scratch) against the workspace-taking entry
point it is a variant of: bit-identical returns and moments, on two calls in a row, the second on
a smaller input (the scratch keeps the first call's contents)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scan(_lib, r, t, gamma, f64, noalloc):
    L = _lib.lib()
    n = r.numel()
    out = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    mom = torch.full((3,), float("nan"), dtype=torch.float64, device="cuda")
    if noalloc:
        _lib.check(L.ppo_returns_scan_flat_noalloc(r.data_ptr(), int(f64), t.data_ptr(), n, gamma,
                                                   out.data_ptr(), mom.data_ptr(), _lib.stream_ptr()))
    else:
        ws = torch.empty(L.ppo_workspace_bytes(4, 2, n), dtype=torch.uint8, device="cuda")
        _lib.check(L.ppo_returns_scan_flat(r.data_ptr(), int(f64), t.data_ptr(), n, gamma, out.data_ptr(),
                                           mom.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy(), mom.cpu().numpy()


def test_noalloc_scan_equals_workspace_scan_twice_second_smaller():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ppo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(42)
    for n, f64, p in [(1_000_003, False, 1e-3), (70_001, True, 0.02), (5, False, 0.5)]:   # shrinking inputs
        r_np = rng.standard_normal(n)
        r = torch.from_numpy(r_np if f64 else r_np.astype(np.float32)).cuda()
        t = torch.from_numpy((rng.random(n) < p).astype(np.uint8)).cuda()
        want, want_mom = _scan(_lib, r, t, 0.99, f64, noalloc=False)
        got, got_mom = _scan(_lib, r, t, 0.99, f64, noalloc=True)
        assert np.array_equal(got, want), n
        assert np.array_equal(got_mom, want_mom), n
        assert got_mom[0] == n and np.isfinite(got).all()
    # on a stream of its own (the stream's device is checked against the current one: same device here)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        got, got_mom = _scan(_lib, r, t, 0.99, f64, noalloc=True)
    assert np.array_equal(got, want) and np.array_equal(got_mom, want_mom)
    # capacity: one element beyond the fixed maximum is refused with a clear error, nothing is launched
    cap = 1 << 24
    code = L.ppo_returns_scan_flat_noalloc(1, 0, 1, cap + 1, 0.99, 1, 1, None)
    assert code == -4 and b"capacity" in L.ppo_hip_strerror(code)
