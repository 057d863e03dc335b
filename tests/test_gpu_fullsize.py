"""The update kernels at the benchmark's full buffer size (524 288 samples, the C4 shape, and a
HalfCheetah-shaped continuous buffer): parity with the CPU restatement on the full gradient, and
the size-independent properties the path offers - linearity of the mean gradient over shards,
bit-reproducibility, and the full iteration against its own per-step form."""
import numpy as np
import pytest
import torch

from helpers import max_abs, rel_err
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from ppo_amd import _lib, ppo
    return ppo, _lib


def _problem(S, A, cont, T, seed):
    rng = np.random.default_rng(seed)
    P = {k: (rng.standard_normal(s) * (0.3 if k.endswith("weight") else 0.1)).astype(np.float32)
         for k, s in O.param_shapes(S, A).items()}
    x = rng.standard_normal((T, S)).astype(np.float32)
    std = 0.5 if cont else None
    if cont:
        acts = rng.standard_normal((T, A)).astype(np.float32)
    else:
        acts = rng.integers(0, A, T).astype(np.int64)
    logp, V, _, _ = O.evaluate(P, x, acts, cont, std)
    old = (logp + rng.standard_normal(T).astype(np.float32) * 0.15).astype(np.float32)   # ratios in and out of the clip range
    adv = rng.standard_normal(T).astype(np.float32)
    ret = rng.standard_normal(T).astype(np.float32)
    return P, x, acts, old, adv, ret, std


def _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, lo, hi, n_global, outs=False):
    L = _lib.lib()
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    flat = torch.from_numpy(O.flatten_params(P)).cuda()
    n = hi - lo
    xs, a = dev(x[lo:hi], torch.float32), dev(acts[lo:hi], torch.float32 if cont else torch.int32)
    o, ad, r = (dev(v[lo:hi], torch.float32) for v in (old, adv, ret))
    ws = torch.empty(L.ppo_workspace_bytes(S, A, n), dtype=torch.uint8, device="cuda")
    grad = torch.empty(flat.numel() + 2, dtype=torch.float32, device="cuda")
    extra = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)] if outs else [None] * 4
    ptr = lambda t: None if t is None else t.data_ptr()
    _lib.check(L.ppo_epoch_fwdbwd(flat.data_ptr(), S, A, int(cont), float(np.float32(std) ** 2) if cont else 0.0,
                                  xs.data_ptr(), a.data_ptr(), o.data_ptr(), ad.data_ptr(), r.data_ptr(),
                                  n, n_global, 0.2, ptr(extra[0]), ptr(extra[1]), ptr(extra[2]), ptr(extra[3]),
                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    _lib.check(L.ppo_grad_reduce(S, A, ws.data_ptr(), ws.numel(), grad.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return grad.cpu().numpy(), [None if e is None else e.cpu().numpy() for e in extra]


@pytest.mark.parametrize("S,A,cont,T", [(4, 2, False, 524288), (26, 6, True, 262144)])
def test_full_size_gradient_matches_restatement_and_is_linear_over_shards(eng, S, A, cont, T):
    _, _lib = eng
    P, x, acts, old, adv, ret, std = _problem(S, A, cont, T, 7)
    g_full, (lp, v, ent, surr) = _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, 0, T, T, outs=True)
    # (1) parity with the CPU restatement at this size: per-sample outputs and the whole gradient
    out_w, grads_w = O.ppo_loss_and_grads(P, x, acts, old, adv, ret, 0.2, cont, std)
    assert rel_err(lp, out_w["logprobs"]) < 1e-5 and rel_err(v, out_w["state_values"]) < 1e-5
    assert rel_err(ent, out_w["entropy"]) < 1e-5
    want = O.flatten_params(grads_w)
    got = g_full[:-2]
    assert max_abs(got, want) < 1e-4 * np.abs(want).max()
    assert rel_err(g_full[-2] + g_full[-1], out_w["loss_mean"]) < 1e-5
    # (2) linearity: the gradient of the mean over the buffer is the sum of the shards' gradients
    #     when every shard scales by 1/n_global (what the data-parallel path relies on)
    cut = (T * 3) // 8 + 5                                   # uneven, not a multiple of the tile
    g_a, _ = _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, 0, cut, T)
    g_b, _ = _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, cut, T, T)
    assert max_abs(g_a + g_b, g_full) < 2e-6 * np.abs(g_full).max()
    # (3) bit-reproducible
    g_again, _ = _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, 0, T, T)
    assert np.array_equal(g_again, g_full)


@pytest.mark.parametrize("S,A,cont,T", [(1, 1, False, 777), (15, 5, False, 2050), (16, 1, True, 3001),
                                        (31, 2, True, 2000), (32, 8, False, 1500), (40, 3, False, 5000),
                                        (63, 8, True, 4099)])
def test_every_shape_class_matches_restatement(eng, S, A, cont, T):
    """One case per kernel instantiation class (layer-1 width classes at S = 15/16, 31/32 and the
    maximum 63; head widths 1-2, 3-4, 5-8; both heads; buffer sizes that are not multiples of the
    32-sample tile): per-sample outputs and the gradient against the CPU restatement."""
    _, _lib = eng
    P, x, acts, old, adv, ret, std = _problem(S, A, cont, T, 100 + S)
    g, (lp, v, ent, surr) = _fwdbwd(_lib, P, S, A, cont, std, x, acts, old, adv, ret, 0, T, T, outs=True)
    out_w, grads_w = O.ppo_loss_and_grads(P, x, acts, old, adv, ret, 0.2, cont, std)
    assert rel_err(lp, out_w["logprobs"]) < 1e-5 and rel_err(v, out_w["state_values"]) < 1e-5
    assert rel_err(ent, out_w["entropy"]) < 1e-5
    want = O.flatten_params(grads_w)
    assert max_abs(g[:-2], want) < 1e-4 * np.abs(want).max()
    assert rel_err(g[-2] + g[-1], out_w["loss_mean"]) < 1e-5


def test_full_size_iteration_equals_its_per_step_form(eng):
    """One benchmark iteration (C-side rollout loop + fused reduce/Adam) leaves exactly the state
    that the per-step public calls + the two-launch reduce/Adam leave."""
    ppo, _lib = eng
    N, T, K = 4096, 128, 3
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, K, 0.2, False, None)
        agent.buffer.reserve(N * T)
        if fused:
            agent.synthetic_rollout(N, T, 0, 1.0 / 200.0, step0=0, env_seed=1234)
            agent.update()
        else:
            b = agent.buffer
            for t in range(T):
                slot = agent.next_state_slot(N)
                r, term = b.claim_env_slots(N)
                _lib.check(_lib.lib().ppo_synth_env_step(slot.data_ptr(), r.data_ptr(), term.data_ptr(), N, 4, 0,
                                                         1.0 / 200.0, t, T, 1234, _lib.stream_ptr()))
                agent.act_inplace(N)
            agent.update(debug=True)
        torch.cuda.synchronize()
        res.append((agent.policy.flat.clone(), agent.last_loss.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[0][1]).all()


def test_c5_full_size_rollout_and_update(eng):
    """The large configuration at its real size: 65 536 envs x 128 steps of a HalfCheetah-shaped
    continuous policy (8 388 608 samples, observations 26 wide: 872 MB of states).  One fused rollout and
    one update epoch: finite and bit-reproducible; the first epoch re-evaluates the rollout's log-probs
    to rounding; the full gradient is the sum of two uneven shards' gradients; and on a 262 144-sample
    slice the gradient equals the CPU restatement's."""
    ppo, _lib = eng
    L = _lib.lib()
    S, A, N, T = 26, 6, 65536, 128
    n = N * T
    res = []
    keep = None
    for rep in range(2):
        torch.manual_seed(0)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 1, 0.2, True, 0.6)
        agent.buffer.reserve(n)
        agent.synthetic_rollout(N, T, 1, -1.0, step0=0, env_seed=1234)
        b = agent.buffer
        if rep == 0:
            flat0 = agent.policy.flat.clone()
            keep = [b.rows(k, 0, n).clone() for k in ("states", "actions", "logprobs")]
        agent.update(debug=(rep == 0))
        torch.cuda.synchronize()
        res.append((agent.policy.flat.clone(), agent.last_loss.clone()))
        if rep == 0:
            dbg = agent.debug_outputs
        del agent
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])    # bit-reproducible
    assert not torch.equal(res[0][0], flat0)                                           # and it did step
    xs, acts, old = keep
    # the update kernel re-evaluates the rollout's log-probs (its output layer sums in another order
    # than act's, so to rounding, not to the bit): ratios of the first epoch are 1 within 1e-5
    assert (dbg["logprobs"] - old).abs().max().item() < 1e-5
    adv, ret, g_full = dbg["advantages"], dbg["returns_norm"], dbg["grad"]
    assert torch.isfinite(adv).all() and torch.isfinite(ret).all() and torch.isfinite(g_full).all()
    var = float(np.float32(0.6 * 0.6))
    ws = torch.empty(L.ppo_workspace_bytes(S, A, n), dtype=torch.uint8, device="cuda")

    def grad_of(lo, hi, n_global):
        g = torch.empty(flat0.numel() + 2, dtype=torch.float32, device="cuda")
        _lib.check(L.ppo_epoch_fwdbwd(flat0.data_ptr(), S, A, 1, var, xs[lo:hi].data_ptr(), acts[lo:hi].data_ptr(),
                                      old[lo:hi].data_ptr(), adv[lo:hi].data_ptr(), ret[lo:hi].data_ptr(),
                                      hi - lo, n_global, 0.2, None, None, None, None, ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr()))
        _lib.check(L.ppo_grad_reduce(S, A, ws.data_ptr(), ws.numel(), g.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return g.cpu().numpy()

    P = flat0.numel()
    g_again = grad_of(0, n, n)
    assert np.array_equal(g_again[:P], g_full.cpu().numpy())
    cut = (n * 3) // 8 + 5                                    # uneven, not a multiple of the tile
    g_sum = grad_of(0, cut, n) + grad_of(cut, n, n)
    assert max_abs(g_sum, g_again) < 2e-6 * np.abs(g_again).max()
    # a 262 144-sample slice against the CPU restatement
    m = 262144
    Pd = O.unflatten_params(flat0.cpu().numpy(), S, A)
    h = lambda t: t[:m].cpu().numpy()
    out_w, grads_w = O.ppo_loss_and_grads(Pd, h(xs), h(acts), h(old), h(adv), h(ret), 0.2, True, 0.6)
    want = O.flatten_params(grads_w)
    got = grad_of(0, m, m)
    assert max_abs(got[:P], want) < 1e-4 * np.abs(want).max()
    assert rel_err(got[-2] + got[-1], out_w["loss_mean"]) < 1e-5


def test_symmetric_wave_kernel_passes_the_same_parity_cases():
    """The alternative fwd/bwd kernel (symmetric waves on 16-sample tiles; in libppo_hip_sym.so, selected
    with PPO_FWDBWD=sym; the variable is read once per process) against the same restatement / golden cases, in a child
    process: every shape class it supports (S <= 31), the golden first-epoch and K-epoch cases."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), "ppo_amd", "libppo_hip_sym.so")   # the variant that carries it
    assert os.path.exists(lib), "python -m ppo_amd.build makes libppo_hip_sym.so next to the product library"
    env = dict(os.environ, PPO_FWDBWD="sym", PPO_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_fullsize.py"), os.path.join(here, "test_gpu_parity.py"),
                        "-k", "every_shape_class or first_epoch_golden or k_epochs_golden or full_size_gradient"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
