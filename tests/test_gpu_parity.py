"""Parity of the HIP engine (through the C ABI) with the reference: golden vectors captured from
the reference itself, and the CPU restatement in oracle/ on seeded inputs.  Needs an MI355X."""
import ctypes
import json
import os

import time

import numpy as np
import pytest
import torch

from helpers import (GOLDEN, TWO_UPDATE_CASES, UPDATE_CASES, case_params, load_case, load_trajectory,
                     load_branch_state, max_abs, nearest_branch, nearest_branch_distance, rel_err)
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu

TOL_REL = 1e-5     # returns / advantages / log-probs / loss (north star)
TOL_PARAM = 1e-4   # parameters after an update

# After more than a handful of epochs the reference's result is not a single point: on identical data
# its own update() ends up to 1.5e-4 (C1, K=80) / 4.1e-4 (C3, K=80) away from its golden run when only
# the order of its fp32 sums changes (intra-op threads, sample order), on a small number of distinct
# end points (tests/golden/reference_spread.json, <case>_branches.npz, recorded by gen_golden.py from
# the reference's own runs).  So for K > 4 the checks are: (1) within 1e-4 of the NEAREST recorded
# end point of the reference, and (2) along the golden run's trajectory, within 1e-4 + the
# reference's own spread at that epoch.  K <= 4 keeps the plain 1e-4 bound against the golden run.


def tol_rel(name):
    """edge_saturated_softmax scales the actor's output weights by 60 on purpose, so 1-ulp
    differences in the hidden activations reach the logits 60x larger: two fp32 CPU implementations
    (torch vs numpy) already differ by 5.5e-6 on its entropy.  Its bound is 5e-5; every other case
    keeps 1e-5."""
    return 5e-5 if name == "edge_saturated_softmax" else TOL_REL


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ppo_amd import _lib, ppo
    assert os.path.exists(_lib.LIB_PATH)
    return ppo, _lib


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def scan_flat(_lib, rewards, terminals, gamma, f64=True):
    n = len(rewards)
    L = _lib.lib()
    r = dev(rewards, torch.float64 if f64 else torch.float32)
    t = dev(np.asarray(terminals, dtype=np.uint8))
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    mom = torch.zeros(3, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.ppo_workspace_bytes(4, 2, n), dtype=torch.uint8, device="cuda")
    _lib.check(L.ppo_returns_scan_flat(r.data_ptr(), int(f64), t.data_ptr(), n, float(gamma),
                                       out.data_ptr(), mom.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out, mom


def normalize(_lib, returns, values, mom):
    n = returns.numel()
    rn = torch.empty_like(returns)
    adv = torch.empty_like(returns)
    _lib.check(_lib.lib().ppo_normalize_advantage(returns.data_ptr(), values.data_ptr(), n,
                                                  mom.data_ptr(), rn.data_ptr(), adv.data_ptr(),
                                                  _lib.stream_ptr()))
    torch.cuda.synchronize()
    return rn, adv


# ------------------------------------------------------------------------------ returns
def test_scan_flat_golden(eng):
    _, _lib = eng
    d = np.load(os.path.join(GOLDEN, "scan_cases.npz"))
    for i in range(int(d["n_cases"])):
        out, mom = scan_flat(_lib, d[f"{i}/rewards"], d[f"{i}/terminals"], float(d[f"{i}/gamma"]))
        want = d[f"{i}/returns32"]
        assert np.array_equal(out.cpu().numpy(), want), f"case {i}: returns not bit-exact"
        n, mean, m2 = mom.cpu().numpy()
        assert n == len(want)
        std = np.sqrt(m2 / (n - 1))
        assert rel_err(mean, d[f"{i}/ret_mean"], floor=float(np.abs(want).mean())) < TOL_REL
        assert rel_err(std, d[f"{i}/ret_std"]) < TOL_REL
        rn, _ = normalize(_lib, out, torch.zeros_like(out), mom)
        assert rel_err(rn.cpu().numpy(), d[f"{i}/returns_norm"]) < TOL_REL


@pytest.mark.parametrize("T,p", [(1, 0.0), (2, 0.5), (2047, 0.01), (2048, 0.0), (2049, 1.0),
                                 (524288, 1 / 200), (1000003, 1e-4)])
def test_scan_flat_vs_oracle(eng, T, p):
    _, _lib = eng
    rng = np.random.default_rng(T)
    r = rng.standard_normal(T).astype(np.float32)
    term = rng.random(T) < p
    want64 = O.returns_scan(r, term, 0.99)
    out, mom = scan_flat(_lib, r, term, 0.99, f64=False)
    got = out.cpu().numpy()
    want = want64.astype(np.float32)
    # carries cross blocks through re-associated fp64 arithmetic: identical after fp32 rounding
    # except for (rare) double-rounding ties
    assert rel_err(got, want) < 1e-6
    assert (got != want).mean() < 1e-4
    r32, mean, std, _ = O.normalize_returns(want64)
    n, gmean, m2 = mom.cpu().numpy()
    assert n == T
    if T > 1:
        assert abs(gmean - mean) <= 1e-5 * max(abs(float(mean)), float(std))
        assert rel_err(np.sqrt(m2 / (T - 1)), std) < TOL_REL


@pytest.mark.parametrize("N,T", [(1, 5), (7, 33), (300, 128), (4096, 128)])
def test_scan_rows_vs_oracle(eng, N, T):
    _, _lib = eng
    L = _lib.lib()
    rng = np.random.default_rng(N * 1000 + T)
    r = rng.standard_normal((N, T)).astype(np.float32)
    term = rng.random((N, T)) < 0.02
    want = O.returns_scan_rows(r, term, 0.99).astype(np.float32)          # [N, T]
    r_tn = dev(r.T.copy())
    t_tn = dev(term.T.copy().astype(np.uint8))
    out = torch.empty((T, N), dtype=torch.float32, device="cuda")
    mom = torch.zeros(3, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.ppo_workspace_bytes(4, 2, N * T), dtype=torch.uint8, device="cuda")
    _lib.check(L.ppo_returns_scan_rows(r_tn.data_ptr(), t_tn.data_ptr(), T, N, 0.99, out.data_ptr(),
                                       mom.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().T, want)   # same serial order as the reference: exact
    n, mean, m2 = mom.cpu().numpy()
    assert n == N * T
    assert abs(mean - want.astype(np.float64).mean()) < 1e-9 * N * T
    assert rel_err(m2, ((want.astype(np.float64) - want.astype(np.float64).mean()) ** 2).sum()) < 1e-9
    # property: equals the flat scan of the env-major concatenation with row ends flagged terminal
    forced = term.copy()
    forced[:, -1] = True
    flat, _ = scan_flat(_lib, r.reshape(-1), forced.reshape(-1), 0.99, f64=False)
    assert rel_err(flat.cpu().numpy().reshape(N, T), want) < 1e-6


def test_scan_properties_random_shapes(eng):
    """Property test over random lengths, terminal densities and discounts (hypothesis): the flat
    scan equals the serial fp64 recurrence after the fp32 rounding (carries cross blocks through
    re-associated fp64 arithmetic: identical except for rare double-rounding ties), the column scan
    of a [T, N] rollout is bit-exact, and the all-terminal / no-terminal extremes hold."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    _, _lib = eng
    L = _lib.lib()

    @settings(max_examples=40, deadline=None, derandomize=True,
              suppress_health_check=list(HealthCheck))
    @given(T=st.integers(1, 9000), p=st.sampled_from([0.0, 0.002, 0.05, 0.5, 1.0]),
           gamma=st.sampled_from([0.9, 0.99, 0.999, 1.0]), seed=st.integers(0, 2 ** 31 - 1),
           f64=st.booleans())
    def flat(T, p, gamma, seed, f64):
        rng = np.random.default_rng(seed)
        r = rng.standard_normal(T).astype(np.float32)
        term = rng.random(T) < p
        want = O.returns_scan(r, term, gamma).astype(np.float32)
        out, mom = scan_flat(_lib, r, term, gamma, f64=f64)
        got = out.cpu().numpy()
        assert np.isfinite(got).all()
        assert rel_err(got, want) < 1e-6 and (got != want).sum() <= max(1, T // 2000)
        if p == 1.0:
            assert np.array_equal(got, r)                      # every step resets: return == reward
        assert mom.cpu().numpy()[0] == T

    @settings(max_examples=25, deadline=None, derandomize=True,
              suppress_health_check=list(HealthCheck))
    @given(N=st.integers(1, 700), T=st.integers(1, 140), p=st.sampled_from([0.0, 0.02, 1.0]),
           seed=st.integers(0, 2 ** 31 - 1))
    def rows(N, T, p, seed):
        rng = np.random.default_rng(seed)
        r = rng.standard_normal((N, T)).astype(np.float32)
        term = rng.random((N, T)) < p
        want = O.returns_scan_rows(r, term, 0.99).astype(np.float32)
        out = torch.empty((T, N), dtype=torch.float32, device="cuda")
        mom = torch.zeros(3, dtype=torch.float64, device="cuda")
        ws = torch.empty(L.ppo_workspace_bytes(4, 2, N * T), dtype=torch.uint8, device="cuda")
        r_tn, t_tn = dev(r.T.copy()), dev(term.T.copy().astype(np.uint8))    # keep both alive over the launch
        _lib.check(L.ppo_returns_scan_rows(r_tn.data_ptr(), t_tn.data_ptr(), T, N, 0.99, out.data_ptr(),
                                           mom.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().T, want)
        n, mean, _ = mom.cpu().numpy()
        assert n == N * T and abs(mean - want.astype(np.float64).mean()) < 1e-9 * N * T

    flat()
    rows()


# ------------------------------------------------------------------------------ evaluate / act
def flat_params(P):
    return dev(O.flatten_params(P))


def var32(std):
    return float(np.float32(float(std) * float(std)))


def run_evaluate(_lib, P, S, A, cont, std, states, actions):
    n = states.shape[0]
    flat = flat_params(P)
    x = dev(states, torch.float32)
    a = dev(actions, torch.float32 if cont else torch.int32)
    lp, v, ent = (torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3))
    _lib.check(_lib.lib().ppo_evaluate(flat.data_ptr(), S, A, int(cont), var32(std) if cont else 0.0,
                                       x.data_ptr(), a.data_ptr(), n, lp.data_ptr(), v.data_ptr(),
                                       ent.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return lp.cpu().numpy(), v.cpu().numpy(), ent.cpu().numpy()


@pytest.mark.parametrize("name", UPDATE_CASES)
def test_evaluate_golden(eng, name):
    _, _lib = eng
    d = load_case(name)
    cont = bool(d["continuous"])
    lp, v, ent = run_evaluate(_lib, case_params(d), int(d["S"]), int(d["A"]), cont,
                              float(d["action_std"]), d["states"], d["actions"])
    assert rel_err(lp, d["logprobs_e1"]) < tol_rel(name)
    assert rel_err(v, d["state_values_e1"]) < TOL_REL
    assert rel_err(ent, d["entropy_e1"]) < tol_rel(name)


def test_shipped_checkpoints(eng):
    _, _lib = eng
    d = np.load(os.path.join(GOLDEN, "checkpoints.npz"))
    for env in d["envs"]:
        P = {k[len(env) + 7:]: d[k] for k in d.files if k.startswith(f"{env}/param/")}
        cont = bool(d[f"{env}/continuous"])
        lp, v, ent = run_evaluate(_lib, P, int(d[f"{env}/S"]), int(d[f"{env}/A"]), cont, 0.6,
                                  d[f"{env}/states"], d[f"{env}/actions"])
        assert rel_err(v, d[f"{env}/critic_out"]) < TOL_REL, env
        assert rel_err(lp, d[f"{env}/logprobs"]) < TOL_REL, env
        assert rel_err(ent, d[f"{env}/entropy"]) < TOL_REL, env


@pytest.mark.parametrize("S,A,cont,N", [(4, 2, False, 4096), (8, 4, False, 1000), (26, 6, True, 4096),
                                        (3, 1, True, 77), (40, 8, True, 513), (63, 7, False, 300)])
def test_act_batched(eng, S, A, cont, N):
    ppo, _lib = eng
    torch.manual_seed(S * 100 + A)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 1, 0.2, cont, 0.6 if cont else None)
    x = torch.randn(N, S, device="cuda")
    act = agent.select_action(x)
    torch.cuda.synchronize()
    b = agent.buffer
    assert len(b.states) == N and len(b.actions) == N and len(b.logprobs) == N
    P = {k: v.cpu().numpy() for k, v in agent.policy_old.state_dict().items()}
    a_np = act.cpu().numpy()
    lp_want, v_want = O.act_logprob_value(P, x.cpu().numpy(), a_np, cont, 0.6)
    lp = b.rows("logprobs", 0, N).cpu().numpy()
    v = b.rows("state_values", 0, N).cpu().numpy()
    assert rel_err(lp, lp_want) < TOL_REL
    assert rel_err(v, v_want) < TOL_REL
    if cont:
        _, _, ao = O.mlp_forward(P, "actor", x.cpu().numpy())
        z = (a_np - np.tanh(ao)) / 0.6
        if N * A > 2000:
            assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05
    else:
        assert a_np.min() >= 0 and a_np.max() < A
        if N >= 1000:
            _, _, ao = O.mlp_forward(P, "actor", x.cpu().numpy())
            p = np.exp(ao - ao.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            freq = np.bincount(a_np, minlength=A) / N
            assert np.abs(freq - p.mean(0)).max() < 0.05


@pytest.mark.parametrize("A", [5, 6, 8])
def test_act_noise_is_independent_across_envs_and_action_dims(eng, A):
    """Every (step, env) owns its own Philox counter range: the Gaussian draw of env m in action
    dims 4.. must not be the draw of env m+1 in dims 0.. (one generator block = 4 normals), and
    adjacent envs / consecutive steps are uncorrelated."""
    ppo, _ = eng
    S, N = 7, 16384
    torch.manual_seed(A)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 1, 0.2, True, 0.6)
    x = torch.randn(N, S, device="cuda")
    P = {k: v.cpu().numpy() for k, v in agent.policy_old.state_dict().items()}
    _, _, ao = O.mlp_forward(P, "actor", x.cpu().numpy())
    mu = np.tanh(ao)
    z = [(agent.select_action(x).cpu().numpy() - mu) / 0.6 for _ in range(2)]    # two steps, same states
    corr = lambda a, b: float(np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1])
    w = A - 4
    assert np.abs(z[0][:-1, 4:] - z[0][1:, :w]).min() > 1e-6      # the old overlap made these equal
    assert abs(corr(z[0][:-1, 4:], z[0][1:, :w])) < 0.03
    assert abs(corr(z[0][:-1], z[0][1:])) < 0.03                    # adjacent envs
    assert abs(corr(z[0], z[1])) < 0.03                             # consecutive steps
    for a in range(A):
        for b in range(a + 1, A):
            assert abs(corr(z[0][:, a], z[0][:, b])) < 0.04         # action dims of one env
    assert abs(z[0].mean()) < 0.02 and abs(z[0].std() - 1.0) < 0.02


def test_synth_env_streams_do_not_overlap(eng):
    """Reward/terminal block and observation blocks of adjacent synthetic envs are distinct draws."""
    _, _lib = eng
    N, S = 16384, 10
    st = torch.empty(N, S, device="cuda")
    r = torch.empty(N, device="cuda")
    t = torch.empty(N, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().ppo_synth_env_step(st.data_ptr(), r.data_ptr(), t.data_ptr(), N, S, 1, 0.1, 5, 128,
                                             77, _lib.stream_ptr()))
    torch.cuda.synchronize()
    st, r = st.cpu().numpy(), r.cpu().numpy()
    corr = lambda a, b: float(np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1])
    assert np.abs(r[1:] - st[:-1, 0]).min() > 1e-7                  # was: reward[e+1] == state[e][0]
    assert abs(corr(r[1:], st[:-1, 0])) < 0.03
    assert abs(corr(st[:-1, 4:8], st[1:, 0:4])) < 0.03
    assert abs(corr(st[:-1], st[1:])) < 0.03
    assert abs(st.mean()) < 0.02 and abs(st.std() - 1.0) < 0.02


def test_select_action_reference_surface(eng):
    ppo, _ = eng
    agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 2, 0.2, False, None)
    a = agent.select_action(np.zeros(4))
    assert isinstance(a, int) and a in (0, 1)
    agent.buffer.rewards.append(1.0)
    agent.buffer.is_terminals.append(False)
    assert agent.buffer.states[0].shape == (4,) and agent.buffer.actions[0].dtype == torch.int64
    assert agent.buffer.state_values[0].shape == (1,) and agent.buffer.logprobs[0].shape == ()
    agent.select_action(np.ones(4))
    agent.buffer.rewards.append(0.5)
    agent.buffer.is_terminals.append(True)
    agent.update()
    assert len(agent.buffer.states) == 0 and len(agent.buffer.rewards) == 0
    cagent = ppo.PPO(3, 2, 3e-4, 1e-3, 0.99, 2, 0.2, True, 0.6)
    a = cagent.select_action(np.zeros(3))
    assert isinstance(a, np.ndarray) and a.shape == (2,) and a.dtype == np.float32
    assert cagent.buffer.actions[0].shape == (1, 2) and cagent.buffer.logprobs[0].shape == (1,)
    cagent.buffer.clear()
    assert len(cagent.buffer.actions) == 0


def _flat_scan_cases():
    rng = np.random.default_rng(77)
    cases = []
    for n, p, f64 in [(2, 0.5, False), (63, 0.1, True), (257, 0.0, False), (900, 1 / 300, True), (1600, 1 / 200, True),
                      (2048, 0.01, False), (2049, 0.01, True), (6000, 1 / 1500, True), (8191, 0.002, False),
                      (8192, 0.002, True)]:
        r = rng.standard_normal(n)
        cases.append((r if f64 else r.astype(np.float32), (rng.random(n) < p).astype(np.uint8), f64))
    return cases


def _run_flat_scans(_lib):
    L = _lib.lib()
    outs = []
    for r, t, f64 in _flat_scan_cases():
        n = r.size
        rd, td_ = torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()
        ret = torch.empty(n, dtype=torch.float32, device="cuda")
        mom = torch.empty(3, dtype=torch.float64, device="cuda")
        ws = torch.empty(L.ppo_workspace_bytes(4, 2, n), dtype=torch.uint8, device="cuda")
        _lib.check(L.ppo_returns_scan_flat(rd.data_ptr(), int(f64), td_.data_ptr(), n, 0.99, ret.data_ptr(),
                                           mom.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((ret.cpu().numpy(), mom.cpu().numpy()))
    return outs


def test_single_launch_flat_scan_equals_three_launch_form(eng, tmp_path):
    """Sequences of up to 8192 steps take a single-workgroup kernel (one launch instead of four).  Its
    returns are the oracle's bit for bit, and returns AND moments equal the three-launch form's bit
    for bit (a child process with PPO_SCAN_SMALL=0 runs the same cases through that form)."""
    import subprocess
    import sys
    _, _lib = eng
    got = _run_flat_scans(_lib)
    for (r, t, f64), (ret, mom) in zip(_flat_scan_cases(), got):
        want = O.returns_scan(r, t.astype(bool), 0.99)
        assert np.array_equal(ret, want.astype(np.float32)), r.size
        assert mom[0] == r.size and abs(mom[1] - ret.astype(np.float64).mean()) < 1e-12 * max(1.0, abs(mom[1]))
    out = tmp_path / "three_launch.npz"
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from ppo_amd import _lib\nimport test_gpu_parity as T\n"
            "o = T._run_flat_scans(_lib)\n"
            "np.savez(%r, **{f'r{i}': a for i, (a, _) in enumerate(o)}, **{f'm{i}': b for i, (_, b) in enumerate(o)})\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), str(out)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPO_SCAN_SMALL="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    z = np.load(out)
    for i, (ret, mom) in enumerate(got):
        assert np.array_equal(ret, z[f"r{i}"]), i
        assert np.array_equal(mom, z[f"m{i}"]), i


def _torch_reference_model(P, S, A, cont, std):
    """The reference's ActorCritic in plain PyTorch on the CPU (PPO.py:49-73, 102-121), with the given
    weights: the fp32 reference the autograd path is compared with."""
    import torch.nn as nn
    from torch.distributions import Categorical, MultivariateNormal
    head = nn.Tanh() if cont else nn.Softmax(dim=-1)
    actor = nn.Sequential(nn.Linear(S, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, A), head)
    critic = nn.Sequential(nn.Linear(S, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, 1))
    model = nn.ModuleDict({"actor": actor, "critic": critic})
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in P.items()})

    def evaluate(state, action):
        if cont:
            mean = actor(state)
            var = torch.full((A,), std * std).expand_as(mean)
            dist = MultivariateNormal(mean, torch.diag_embed(var))
            if A == 1:
                action = action.reshape(-1, A)
        else:
            dist = Categorical(actor(state))
        return dist.log_prob(action), critic(state), dist.entropy()

    return model, evaluate


@pytest.mark.parametrize("S,A,cont,T", [(4, 2, False, 1000), (8, 4, False, 333), (24, 4, True, 2049), (26, 6, True, 700),
                                        (40, 3, False, 500)])
def test_evaluate_is_differentiable_like_the_reference(eng, S, A, cont, T):
    """policy.evaluate() carries autograd history w.r.t. the 12 parameters, as the reference's does
    (PPO.py:102-121 + PPO.py:244): for a loss built from all three outputs with random per-sample
    weights, .grad of every parameter equals plain PyTorch's on the same model (1e-4 of the tensor's
    largest entry), a second backward accumulates, and no_grad() gives plain tensors."""
    ppo, _ = eng
    rng = np.random.default_rng(S * 100 + A)
    std = 0.5
    P = {k: (rng.standard_normal(s) * (0.3 if k.endswith("weight") else 0.1)).astype(np.float32)
         for k, s in O.param_shapes(S, A).items()}
    x = rng.standard_normal((T, S)).astype(np.float32)
    acts = rng.standard_normal((T, A)).astype(np.float32) if cont else rng.integers(0, A, T)
    w = [rng.standard_normal(T).astype(np.float32) for _ in range(3)]
    model, ref_eval = _torch_reference_model(P, S, A, cont, std)
    lp, v, ent = ref_eval(torch.from_numpy(x), torch.from_numpy(acts))
    loss_ref = (torch.from_numpy(w[0]) * lp + torch.from_numpy(w[1]) * v.squeeze(-1) ** 2 + torch.from_numpy(w[2]) * ent).sum()
    loss_ref.backward()
    want = {k: p.grad.numpy() for k, p in model.named_parameters()}

    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 1, 0.2, cont, std if cont else None)
    agent.policy.load_state_dict({k: torch.from_numpy(val) for k, val in P.items()})
    pol = agent.policy
    assert all(p.requires_grad for p in pol.parameters())
    dev = lambda a: torch.from_numpy(a).cuda()
    lp_g, v_g, ent_g = pol.evaluate(dev(x), dev(acts))
    assert lp_g.requires_grad and v_g.shape == (T, 1)
    assert rel_err(lp_g.detach().cpu().numpy(), lp.detach().numpy()) < TOL_REL
    loss = (dev(w[0]) * lp_g + dev(w[1]) * v_g.squeeze(-1) ** 2 + dev(w[2]) * ent_g).sum()
    assert rel_err(np.array([loss.item()]), np.array([loss_ref.item()])) < 1e-4
    loss.backward()
    for k, p in pol.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        assert max_abs(p.grad.cpu().numpy(), want[k]) < 1e-4 * max(float(np.abs(want[k]).max()), 1e-30), k
    first = {k: p.grad.clone() for k, p in pol.named_parameters()}
    lp_g, v_g, ent_g = pol.evaluate(dev(x), dev(acts))
    (dev(w[0]) * lp_g).sum().backward()                      # only one output used: the others' seeds are zeros
    lp2, _, _ = ref_eval(torch.from_numpy(x), torch.from_numpy(acts))
    model.zero_grad()
    (torch.from_numpy(w[0]) * lp2).sum().backward()
    for k, p in pol.named_parameters():
        acc = p.grad - first[k]
        ref_g = dict(model.named_parameters())[k].grad
        ref_g = np.zeros(p.shape, np.float32) if ref_g is None else ref_g.numpy()
        scale = max(float(np.abs(want[k]).max()), 1e-30)
        assert max_abs(acc.cpu().numpy(), ref_g) < 2e-4 * scale, k
    with torch.no_grad():
        assert not pol.evaluate(dev(x), dev(acts))[0].requires_grad
    # a write to the parameters between evaluate() and backward() is refused, as autograd refuses it
    lp_g, _, _ = pol.evaluate(dev(x), dev(acts))
    with torch.no_grad():
        pol.actor[0].bias.add_(1.0)
    with pytest.raises(RuntimeError):
        lp_g.sum().backward()


def test_reference_update_loop_written_with_autograd_matches_fused_update(eng):
    """The K-epoch loop of PPO.py:224-245 written out by a caller - policy.evaluate, the clipped loss,
    loss.mean().backward(), torch.optim.Adam over the two parameter groups - on this engine's
    ActorCritic gives the parameters the fused update() gives (both compute PPO.py:227-245)."""
    ppo, _ = eng
    d = load_case("c2_lunarlander")
    K = 4
    fused = agent_from_case(ppo, d, K)
    fused.update()
    agent = agent_from_case(ppo, d, K)
    pol = agent.policy
    opt = torch.optim.Adam([{"params": pol.actor.parameters(), "lr": float(d["lr_actor"])},
                            {"params": pol.critic.parameters(), "lr": float(d["lr_critic"])}])
    dev = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to("cuda", dt)
    states, actions = dev(d["states"]), dev(d["actions"], torch.int64)
    old_lp, rewards = dev(d["old_logprobs"]), dev(d["returns_norm"])
    adv = dev(d["advantages"])
    mse = torch.nn.MSELoss()
    for _ in range(K):
        logprobs, state_values, dist_entropy = pol.evaluate(states, actions)
        state_values = torch.squeeze(state_values)
        ratios = torch.exp(logprobs - old_lp.detach())
        surr1 = ratios * adv
        surr2 = torch.clamp(ratios, 1 - 0.2, 1 + 0.2) * adv
        loss = -torch.min(surr1, surr2) + 0.5 * mse(state_values, rewards) - 0.01 * dist_entropy
        opt.zero_grad()
        loss.mean().backward()
        opt.step()
    torch.cuda.synchronize()
    assert max_abs(pol.flat.cpu().numpy(), fused.policy.flat.cpu().numpy()) < 2e-5
    assert max_abs(pol.flat.cpu().numpy(), np.concatenate([d[f"K{K}/param/{k}"].reshape(-1) for k in O.PARAM_KEYS])) < TOL_PARAM


# ------------------------------------------------------------------------------ update
def agent_from_case(ppo, d, K):
    cont = bool(d["continuous"])
    agent = ppo.PPO(int(d["S"]), int(d["A"]), float(d["lr_actor"]), float(d["lr_critic"]),
                    float(d["gamma"]), K, float(d["eps_clip"]), cont,
                    float(d["action_std"]) if cont else None)
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    agent.policy.load_state_dict(sd)
    agent.policy_old.load_state_dict(sd)
    agent.buffer.load_rollout(d["states"], d["actions"], d["old_logprobs"], d["old_values"],
                              d["rewards"], d["terminals"])
    return agent


@pytest.mark.parametrize("name", UPDATE_CASES)
def test_update_first_epoch_golden(eng, name):
    ppo, _ = eng
    d = load_case(name)
    K = int(d["K_list"][0])
    agent = agent_from_case(ppo, d, K)
    agent.update(debug=True)
    torch.cuda.synchronize()
    g = agent.debug_outputs
    T = int(d["T"])
    n, mean, m2 = g["moments"].cpu().numpy()
    assert n == T
    assert rel_err(mean, d["ret_mean"], floor=float(np.abs(d["returns32"]).mean())) < TOL_REL
    assert rel_err(np.sqrt(m2 / (T - 1)), d["ret_std"]) < TOL_REL
    assert rel_err(g["returns_norm"].cpu().numpy(), d["returns_norm"]) < TOL_REL
    assert rel_err(g["advantages"].cpu().numpy(), d["advantages"]) < TOL_REL
    assert rel_err(g["logprobs"].cpu().numpy(), d["logprobs_e1"]) < tol_rel(name)
    assert rel_err(g["state_values"].cpu().numpy(), d["state_values_e1"]) < TOL_REL
    assert rel_err(g["entropy"].cpu().numpy(), d["entropy_e1"]) < tol_rel(name)
    # per-sample loss = -surrogate + 0.5*mse - 0.01*entropy (PPO.py:240)
    mse = float(((d["state_values_e1"].astype(np.float64) - d["returns_norm"]) ** 2).mean())
    loss = -g["surrogate"].cpu().numpy() + np.float32(0.5 * mse) - np.float32(0.01) * g["entropy"].cpu().numpy()
    assert rel_err(loss, d["loss_per_sample_e1"]) < tol_rel(name)
    assert rel_err(agent.last_loss.cpu().numpy()[0], d["loss_mean_e1"]) < TOL_REL
    flat_grad = g["grad"].cpu().numpy()
    grads = O.unflatten_params(flat_grad, int(d["S"]), int(d["A"]))
    for k in O.PARAM_KEYS:
        want = d["grad_e1/" + k]
        assert rel_err(grads[k], want, floor=float(np.abs(want).max())) < 1e-4, k
    got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
    worst = max(max_abs(got[k], d[f"K{K}/param/{k}"]) for k in O.PARAM_KEYS)
    assert worst < TOL_PARAM, worst
    old = {k: v.cpu().numpy() for k, v in agent.policy_old.state_dict().items()}
    assert all(np.array_equal(old[k], got[k]) for k in O.PARAM_KEYS)       # PPO.py:248
    assert agent.optimizer.step_count == K


@pytest.mark.parametrize("name", UPDATE_CASES)
def test_update_k_epochs_golden(eng, name):
    ppo, _ = eng
    d = load_case(name)
    for K in d["K_list"][1:]:
        K = int(K)
        agent = agent_from_case(ppo, d, K)
        agent.update()
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
        S, A = int(d["S"]), int(d["A"])
        if K <= 4:
            worst = max(max_abs(got[k], d[f"K{K}/param/{k}"]) for k in O.PARAM_KEYS)
            check_adam_state(agent, lambda kind, k: d[f"K{K}/{kind}/{k}"], S, A)
        else:
            # the reference's own end point this run lands on, and the reference optimizer's moments THERE
            worst, idx = nearest_branch(agent.policy.flat.cpu().numpy(), name)
            bm, bv = (O.unflatten_params(x, S, A) for x in load_branch_state(name, idx))
            check_adam_state(agent, lambda kind, k: (bm if kind == "exp_avg" else bv)[k], S, A, tol=3e-4)
        assert worst < TOL_PARAM, (K, worst)
        assert rel_err(agent.last_loss.cpu().numpy(), d[f"K{K}/loss_mean_all"]) < 1e-4, K
        assert agent.optimizer.step_count == int(d[f"K{K}/step"])


@pytest.mark.parametrize("name", ["c1_cartpole", "c2_lunarlander", "c3_bipedal"])
def test_update_trajectory_within_reference_spread(eng, name):
    """Parameters after 1, 2, 4, 8, ... epochs against the golden run's trajectory (one update()
    per snapshot epoch count, each from the same initial state)."""
    ppo, _ = eng
    d = load_case(name)
    epochs, snaps, curve = load_trajectory(name)
    for K in epochs[:-1]:
        agent = agent_from_case(ppo, d, K)
        agent.update()
        dev = max_abs(agent.policy.flat.cpu().numpy(), snaps[K])
        assert dev < TOL_PARAM + float(curve[K - 1]), (K, dev)


def check_adam_state(agent, want, S, A, tol=1.5e-4):
    """exp_avg / exp_avg_sq of the engine's optimizer against the reference optimizer's state,
    per tensor, relative to the tensor's largest entry (the gradients they average agree to
    <= 1e-4 of the tensor's max).  Measured: <= 4.8e-5 after K <= 4 epochs, <= 9.9e-5 against the
    nearest recorded end point's state after 30 / 80 epochs, <= 6.5e-6 after each of two
    consecutive updates (tools/parity_report.py)."""
    m = O.unflatten_params(agent.optimizer.exp_avg.cpu().numpy(), S, A)
    v = O.unflatten_params(agent.optimizer.exp_avg_sq.cpu().numpy(), S, A)
    for k in O.PARAM_KEYS:
        wm, wv = want("exp_avg", k), want("exp_avg_sq", k)
        assert max_abs(m[k], wm) < tol * max(float(np.abs(wm).max()), 1e-30), ("exp_avg", k)
        assert max_abs(v[k], wv) < tol * max(float(np.abs(wv).max()), 1e-30), ("exp_avg_sq", k)


@pytest.mark.parametrize("name", TWO_UPDATE_CASES)
def test_two_consecutive_updates_golden(eng, name):
    """update, refill, update on one agent: Adam's step count and moments persist across update()
    calls (PPO.py:139 is the optimizer's only construction); the continuous case decays the std in
    between.  Parameters, exp_avg, exp_avg_sq and step against the reference after each update."""
    ppo, _ = eng
    d = load_case(name)
    cont, K, S, A = bool(d["continuous"]), int(d["K"]), int(d["S"]), int(d["A"])
    agent = ppo.PPO(S, A, float(d["lr_actor"]), float(d["lr_critic"]), float(d["gamma"]), K,
                    float(d["eps_clip"]), cont, float(d["action_std"]) if cont else None)
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    agent.policy.load_state_dict(sd)
    agent.policy_old.load_state_dict(sd)
    for u in (1, 2):
        if cont and float(d[f"u{u}/action_std"]) != agent.action_std:
            agent.decay_action_std(0.05, 0.1)
            assert agent.action_std == float(d[f"u{u}/action_std"])
        agent.buffer.load_rollout(d[f"u{u}/states"], d[f"u{u}/actions"], d[f"u{u}/old_logprobs"],
                                  d[f"u{u}/old_values"], d[f"u{u}/rewards"], d[f"u{u}/terminals"])
        agent.update()
        torch.cuda.synchronize()
        assert agent.optimizer.step_count == int(d[f"u{u}/after/step"]) == u * K
        got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
        worst = max(max_abs(got[k], d[f"u{u}/after/param/{k}"]) for k in O.PARAM_KEYS)
        assert worst < TOL_PARAM, (u, worst)
        check_adam_state(agent, lambda kind, k: d[f"u{u}/after/{kind}/{k}"], S, A, tol=5e-5)
        assert rel_err(agent.last_loss.cpu().numpy(), d[f"u{u}/loss_mean_all"]) < 1e-4


def test_resume_checkpoint_continues_the_same_run(eng, tmp_path):
    """update, save_resume, update  ==  (fresh agent) load_resume, update: parameters, Adam state and
    losses bit for bit - the second update depends on the first one's Adam moments and step count,
    which the reference's own checkpoint (policy_old only) does not carry."""
    ppo, _ = eng
    d = load_case("two_update_continuous")
    cont, K, S, A = bool(d["continuous"]), int(d["K"]), int(d["S"]), int(d["A"])
    make = lambda: ppo.PPO(S, A, float(d["lr_actor"]), float(d["lr_critic"]), float(d["gamma"]), K,
                           float(d["eps_clip"]), cont, float(d["action_std"]))
    load = lambda ag, u: ag.buffer.load_rollout(d[f"u{u}/states"], d[f"u{u}/actions"], d[f"u{u}/old_logprobs"],
                                                d[f"u{u}/old_values"], d[f"u{u}/rewards"], d[f"u{u}/terminals"])
    a = make()
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    a.policy.load_state_dict(sd)
    a.policy_old.load_state_dict(sd)
    load(a, 1)
    a.update()
    if float(d["u2/action_std"]) != a.action_std:
        a.decay_action_std(0.05, 0.1)
    a.save_resume(str(tmp_path / "r.pt"))
    load(a, 2)
    a.update()
    b = make()
    b.load_resume(str(tmp_path / "r.pt"))
    assert b.action_std == float(d["u2/action_std"]) and b.optimizer.step_count == K
    load(b, 2)
    b.update()
    torch.cuda.synchronize()
    assert torch.equal(a.policy.flat, b.policy.flat) and torch.equal(a.policy_old.flat, b.policy_old.flat)
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg) and torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)
    assert torch.equal(a.last_loss, b.last_loss) and b.optimizer.step_count == 2 * K
    got = {k: v.cpu().numpy() for k, v in b.policy.state_dict().items()}
    assert max(max_abs(got[k], d[f"u2/after/param/{k}"]) for k in O.PARAM_KEYS) < TOL_PARAM


def test_update_deterministic(eng):
    ppo, _ = eng
    d = load_case("c2_lunarlander")
    outs = []
    for _ in range(2):
        agent = agent_from_case(ppo, d, 4)
        agent.update()
        outs.append(agent.policy.flat.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["c1_cartpole", "c3_bipedal"])
def test_fused_reduce_adam_equals_two_launch(eng, name):
    """ppo_update_epochs (one call for all epochs) == per-epoch ppo_reduce_adam_step (one launch)
    == ppo_grad_reduce + ppo_adam_step, bit for bit; kernel_events selects the per-epoch fused
    path, debug=True the two-launch path."""
    ppo, _ = eng
    d = load_case(name)
    outs = []
    for mode in ("one_call", "per_epoch", "two_launch"):
        agent = agent_from_case(ppo, d, 4)
        if mode == "per_epoch":
            agent.kernel_events = []
        agent.update(debug=(mode == "two_launch"))
        outs.append((agent.policy.flat.clone(), agent.last_loss.clone(),
                     agent.optimizer.exp_avg.clone(), agent.optimizer.exp_avg_sq.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert agent.optimizer.step_count == 4


@pytest.mark.parametrize("cont,S,A,N,T", [(False, 4, 2, 333, 7), (True, 6, 3, 200, 5),
                                          (True, 24, 6, 100, 3), (False, 63, 5, 50, 3),
                                          (False, 17, 8, 16, 2), (True, 1, 1, 1, 4)])
def test_synthetic_rollout_equals_per_step_calls(eng, cont, S, A, N, T):
    """The one-launch rollout kernel (16-env tiles on the 16x16x4 MFMA, env in a wave of its own)
    writes exactly the buffers of the per-step API (32-sample tiles on the 32x32x2 MFMA), bit for
    bit, and the update that follows lands on the same parameters."""
    ppo, _lib = eng
    cols = ("states", "actions", "logprobs", "state_values", "rewards", "is_terminals")
    got = []
    for fused in (True, False):
        torch.manual_seed(11)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
        agent.buffer.reserve(N * T)
        if fused:
            agent.synthetic_rollout(N, T, reward_kind=int(cont), p_terminal=0.1, step0=3, env_seed=99)
        else:
            b = agent.buffer
            for t in range(T):
                slot = agent.next_state_slot(N)
                r, term = b.claim_env_slots(N)
                _lib.check(_lib.lib().ppo_synth_env_step(
                    slot.data_ptr(), r.data_ptr(), term.data_ptr(), N, S, int(cont), 0.1, 3 + t, T, 99,
                    _lib.stream_ptr()))
                agent.act_inplace(N)
        b = agent.buffer
        got.append([b.rows(c, 0, N * T).clone() for c in cols])
        agent.update()
        got[-1].append(agent.policy.flat.clone())
    for c, a, bb in zip(cols + ("params",), *got):
        assert torch.equal(a, bb), c


@pytest.mark.parametrize("mode", ["session", "session_ctypes", "session_slow_env", "ticket"])
@pytest.mark.parametrize("cont", [False, True])
def test_single_state_host_path_equals_batched_rows(eng, cont, mode, monkeypatch):
    """select_action(np.ndarray[S]) stores exactly what the batched call stores for the same states
    (the one-unit-per-lane VALU kernel reproduces the MFMA tile's fmaf chain), and returns the
    reference's Python types.  session: a long-lived kernel that waits for each state in a mailbox;
    session_slow_env: the env takes longer than the kernel waits, so every waiting kernel gives up
    and the step is served by a fresh one; ticket: one launch per call (ppo_act_host)."""
    ppo, _ = eng
    monkeypatch.setenv("PPO_ACT_SESSION", "0" if mode == "ticket" else "1")
    monkeypatch.setenv("PPO_FASTCALL", "0" if mode == "session_ctypes" else "1")   # the per-step binding: extension / ctypes
    monkeypatch.setenv("PPO_ACT_WAIT_US", "200" if mode == "session_slow_env" else "1000")
    S, A = (24, 4) if cont else (8, 4)
    xs = np.random.default_rng(8).standard_normal((6, S)).astype(np.float32)
    cols = ("states", "actions", "logprobs", "state_values")
    got = []
    for host in (True, False):
        torch.manual_seed(21)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.4 if cont else None)
        outs = []
        for i, x in enumerate(xs):
            if host:
                if mode == "session_slow_env" and i:
                    time.sleep(0.003)
                a = agent.select_action(x.astype(np.float64))       # gym hands over float64
                assert isinstance(a, np.ndarray) and a.dtype == np.float32 if cont else isinstance(a, int)
                outs.append(np.asarray(a))
            else:
                outs.append(agent.select_action(torch.from_numpy(x[None]).cuda()).cpu().numpy()[0])
        assert (agent._act_session is not None) == (host and mode != "ticket")
        got.append(([agent.buffer.rows(c, 0, 6).clone() for c in cols], np.array(outs)))
        if host:
            # an update in between drains the session (new weights), then the protocol goes on
            for _ in range(6):
                agent.buffer.rewards.append(1.0)
                agent.buffer.is_terminals.append(False)
            agent.update()
            a2 = agent.select_action(xs[0])
            ref = agent.policy_old.act(torch.from_numpy(xs[0]).cuda(), _seed=agent._rng_seed,
                                       _offset=agent._rng_offset)
            assert np.array_equal(np.asarray(a2).reshape(-1), ref[0].cpu().numpy().reshape(-1))
            assert torch.equal(agent.buffer.rows("logprobs", 0, 1), ref[1].reshape(1))
            assert torch.equal(agent.buffer.rows("state_values", 0, 1), ref[2].reshape(1))
    for c, a, b in zip(cols, got[0][0], got[1][0]):
        assert torch.equal(a, b), c
    assert np.array_equal(got[0][1], got[1][1])


def test_single_state_session_survives_buffer_growth(eng):
    """1100 single-state steps cross the buffer's first capacity (1024 rows): the session is drained
    before the columns move, the rows written before and after the move are all there, and equal
    the batched call's."""
    ppo, _ = eng
    S, A, n = 4, 2, 1100
    xs = np.random.default_rng(3).standard_normal((n, S)).astype(np.float32)
    got = []
    for host in (True, False):
        torch.manual_seed(5)
        agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, False, None)
        if host:
            acts = [agent.select_action(x) for x in xs]
        else:
            acts = []
            for i in range(0, n, 1):   # one row per call, as the host path numbers its draws
                acts.append(int(agent.select_action(torch.from_numpy(xs[i:i + 1]).cuda())[0]))
        got.append((np.array(acts), [agent.buffer.rows(c, 0, n).clone()
                                     for c in ("states", "actions", "logprobs", "state_values")]))
    assert np.array_equal(got[0][0], got[1][0])
    for a, b in zip(got[0][1], got[1][1]):
        assert torch.equal(a, b)


def test_single_state_session_request_numbers_wrap(eng):
    """70 000 single-state steps through one session: the 16-bit request number carried by every
    mailbox dword wraps once, two pauses make the waiting kernel give up and be replaced, the buffer
    grows several times.  Every stored row equals evaluate() of the stored state and action bit for
    bit, the states are the inputs, the returned actions are the stored ones."""
    ppo, _ = eng
    S, A, n = 6, 3, 70000
    torch.manual_seed(9)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, False, None)
    xs = np.random.default_rng(12).standard_normal((n, S)).astype(np.float32)
    acts = np.empty(n, dtype=np.int64)
    for i in range(n):
        if i in (30000, 65600):
            time.sleep(0.003)
        acts[i] = agent.select_action(xs[i])
    agent._act_drain()
    b = agent.buffer
    st, ac = b.rows("states", 0, n), b.rows("actions", 0, n)
    with torch.no_grad():
        lp, val, _ = agent.policy_old.evaluate(st, ac)
    assert torch.equal(st.cpu(), torch.from_numpy(xs))
    assert np.array_equal(acts, ac.cpu().numpy().astype(np.int64))
    assert torch.equal(lp, b.rows("logprobs", 0, n))
    assert torch.equal(val.reshape(-1), b.rows("state_values", 0, n))
    agent.close()


@pytest.mark.parametrize("cont", [False, True])
def test_single_state_session_sees_in_place_weight_writes(eng, cont):
    """The kernel that serves step k+1 was launched earlier and loaded its weights then.  An
    in-place write to policy_old's parameters between two steps - through the flat vector or through
    one of the nn.Linear views, outside update() / load() / load_state_dict() - must still be seen
    by the next step: its stored log-prob and value equal evaluate() under the NEW weights."""
    ppo, _ = eng
    S, A = (24, 4) if cont else (4, 2)
    torch.manual_seed(3)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
    xs = np.random.default_rng(4).standard_normal((6, S)).astype(np.float32)
    agent.select_action(xs[0])
    agent.select_action(xs[1])          # steady state: a kernel with the current weights waits for step 3
    assert agent._act_session is not None
    pol = agent.policy_old
    writes = [lambda: pol.flat.mul_(0.5),
              lambda: pol.actor[0].weight.add_(0.25),
              lambda: pol.critic[4].bias.fill_(3.0),
              lambda: pol.flat.copy_(agent.policy.flat * 1.5)]
    for i, write in enumerate(writes):
        with torch.no_grad():
            write()
        agent.select_action(xs[2 + i])
        row = 2 + i
        lp, v, _ = pol.evaluate(agent.buffer.rows("states", row, row + 1),
                                agent.buffer.rows("actions", row, row + 1))
        assert torch.equal(agent.buffer.rows("logprobs", row, row + 1), lp.reshape(1)), i
        assert torch.equal(agent.buffer.rows("state_values", row, row + 1), v.reshape(1)), i
    agent.close()
    assert agent._act_session is None
    agent.close()                        # idempotent
    with pytest.raises(ValueError):
        agent.select_action(np.zeros(1))         # a length-1 state is not broadcast into S inputs


@pytest.mark.parametrize("cont,S,A", [(False, 63, 8), (True, 40, 3), (False, 30, 2), (True, 3, 1)])
def test_session_step_by_pointer_equals_batched_act(eng, cont, S, A):
    """ppo_act_session_step through the C-ABI (rows given by pointer, ordinary host memory for state and
    action): rows walked in order, a jump within the same columns (served by the waiting kernel), other
    columns (a new binding), a pause longer than the kernel waits.  Observations of
    30, 40 and 63 values make the request two and three dwords per lane.  Every row equals ppo_act's
    row for the same state and draw, bit for bit, and so does the action handed back."""
    ppo, _lib = eng
    L = _lib.lib()
    torch.manual_seed(17)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
    pol = agent.policy_old
    n = 40
    xs = np.random.default_rng(6).standard_normal((n, S)).astype(np.float32)

    def columns():
        return (torch.zeros((n, S), device="cuda"),
                torch.zeros((n, A), device="cuda") if cont else torch.zeros((n,), dtype=torch.int32, device="cuda"),
                torch.zeros((n,), device="cuda"), torch.zeros((n,), device="cuda"))

    cols_a, cols_b = columns(), columns()
    torch.cuda.synchronize()
    sess = ctypes.c_void_p()
    _lib.check(L.ppo_act_session_create(ctypes.byref(sess), S, A, int(cont), 300))
    state = np.zeros(S, dtype=np.float32)
    out = np.zeros(8, dtype=np.float32)
    seed, stream = 4242, _lib.stream_ptr()
    aw = (A if cont else 1) * 4
    try:
        def step(cols, row, off, var):
            state[:] = xs[row]
            _lib.check(L.ppo_act_session_step(
                sess, pol.flat.data_ptr(), 1, var, state.ctypes.data, seed, off,
                cols[0].data_ptr() + row * S * 4, cols[1].data_ptr() + row * aw, cols[2].data_ptr() + row * 4,
                cols[3].data_ptr() + row * 4, out.ctypes.data, stream))
            return out[:A].copy() if cont else int(out[:1].view(np.int32)[0])

        plan = [(cols_a, r, 100 + r, pol._var_host) for r in (0, 1, 2, 3, 17, 18, 5)]
        plan += [(cols_b, r, 200 + r, pol._var_host) for r in (7, 8, 9)]
        plan += [(cols_a, 30, 300, pol._var_host)]
        returned = []
        for i, (cols, row, off, var) in enumerate(plan):
            if i == 5:
                time.sleep(0.002)     # the waiting kernel gives up (300 us)
            returned.append(step(cols, row, off, var))
        _lib.check(L.ppo_act_session_drain(sess))
        for (cols, row, off, var), ret in zip(plan, returned):
            a, lp, v = pol.act(torch.from_numpy(xs[row]).cuda(), _seed=seed, _offset=off)
            assert torch.equal(cols[0][row].cpu(), torch.from_numpy(xs[row])), row
            assert torch.equal(cols[2][row:row + 1], lp.reshape(1)), row
            assert torch.equal(cols[3][row:row + 1], v.reshape(1)), row
            if cont:
                assert torch.equal(cols[1][row], a.reshape(-1)), row
                assert np.array_equal(ret, a.reshape(-1).cpu().numpy()), row
            else:
                assert int(cols[1][row]) == int(a) == ret, row
        # rows no step addressed stay untouched
        assert float(cols_b[2][:7].abs().sum()) == 0.0 and float(cols_a[2][31:].abs().sum()) == 0.0
    finally:
        _lib.check(L.ppo_act_session_destroy(sess))


@pytest.mark.parametrize("cont,S,A", [(False, 4, 2), (True, 24, 6), (False, 63, 8)])
def test_act_tile_shapes_agree(eng, cont, S, A):
    """Small batches run ppo_act / ppo_evaluate on 16-sample tiles (16x16x4 MFMA), large ones on
    32-sample tiles (32x32x2): the first rows of a large batch equal a small batch of those rows, bit
    for bit (sampling is keyed by the row index)."""
    ppo, _ = eng
    torch.manual_seed(13)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, 2, 0.2, cont, 0.5 if cont else None)
    big, small = 9000, 150
    x = torch.from_numpy(np.random.default_rng(2).standard_normal((big, S)).astype(np.float32)).cuda()
    pol = agent.policy_old
    a_big = pol.act(x, _seed=77, _offset=5)
    a_small = pol.act(x[:small].contiguous(), _seed=77, _offset=5)
    for b_, s_ in zip(a_big, a_small):
        assert torch.equal(b_[:small], s_)
    e_big = pol.evaluate(x, a_big[0])
    e_small = pol.evaluate(x[:small].contiguous(), a_big[0][:small].contiguous())
    for b_, s_ in zip(e_big, e_small):
        assert torch.equal(b_[:small], s_)


def test_kernel_timings_export(eng):
    """ppo_kernel_timings reports the launches of the K-epoch loop of the default update path, and
    nothing once switched off."""
    ppo, _lib = eng
    L = _lib.lib()
    torch.manual_seed(1)
    agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 5, 0.2, False, None)
    out = (ctypes.c_float * 3)()
    for on in (1, 0):
        _lib.check(L.ppo_kernel_timings_enable(on))
        agent.synthetic_rollout(64, 16)
        agent.update()
        _lib.check(L.ppo_kernel_timings(out, 3))
        if on:
            assert int(out[2]) == 5 and 0.0 < out[0] < 50.0 and 0.0 < out[1] < 50.0
        else:
            assert int(out[2]) == 0 and out[0] == 0.0
    assert L.ppo_kernel_timings(out, 2) != 0      # room for fewer than three values: bad argument


def test_batched_update_equals_flat_with_row_end_terminals(eng):
    """N lockstep envs x T steps through the tensor-valued buffer facade == the oracle's flat
    update of the env-major concatenation with every row end flagged terminal."""
    ppo, _ = eng
    S, A, N, T, K = 4, 2, 96, 33, 3
    torch.manual_seed(5)
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, K, 0.2, False, None)
    P0 = {k: v.cpu().numpy().copy() for k, v in agent.policy.state_dict().items()}
    rng = np.random.default_rng(3)
    xs, rs, ts = [], [], []
    for t in range(T):
        x = torch.from_numpy(rng.standard_normal((N, S)).astype(np.float32)).cuda()
        agent.select_action(x)
        r = rng.standard_normal(N).astype(np.float32)
        done = rng.random(N) < 0.05
        agent.buffer.rewards.append(torch.from_numpy(r).cuda())
        agent.buffer.is_terminals.append(torch.from_numpy(done).cuda())
        xs.append(x.cpu().numpy()); rs.append(r); ts.append(done)
    b = agent.buffer
    n = N * T
    # time-major [T, N] rows -> env-major flat order for the oracle
    tn = lambda a: a.reshape(T, N, *a.shape[1:]).swapaxes(0, 1).reshape(n, *a.shape[1:])
    states = tn(b.rows("states", 0, n).cpu().numpy())
    actions = tn(b.rows("actions", 0, n).cpu().numpy())
    logp = tn(b.rows("logprobs", 0, n).cpu().numpy())
    vals = tn(b.rows("state_values", 0, n).cpu().numpy())
    rew = tn(np.stack(rs).reshape(n))
    term = np.stack(ts)
    term[-1, :] = True
    term = tn(term.reshape(n))
    st = O.adam_init(P0)
    O.update(P0, st, states, actions, logp, vals, rew, term, 0.99, 0.2, K, 3e-4, 1e-3, False)
    agent.update()
    got = {k: v.cpu().numpy() for k, v in agent.policy.state_dict().items()}
    assert max(max_abs(got[k], P0[k]) for k in O.PARAM_KEYS) < TOL_PARAM


# ------------------------------------------------------------------------------ data parallel
def _dp_gpu_worker(rank, world, port, case, K, q):
    import contextlib
    import io
    import torch.distributed as td
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    with contextlib.redirect_stdout(io.StringIO()):
        from ppo_amd import ppo
    torch.manual_seed(0)                                  # as bench.py: identical seeding on every rank
    d = load_case(case)
    T = int(d["T"])
    cut = (T * 2) // 5                                   # uneven shards
    lo, hi = (0, cut) if rank == 0 else (cut, T)
    cont = bool(d["continuous"])
    agent = ppo.PPO(int(d["S"]), int(d["A"]), float(d["lr_actor"]), float(d["lr_critic"]),
                    float(d["gamma"]), K, float(d["eps_clip"]), cont,
                    float(d["action_std"]) if cont else None)
    sd = {k: torch.from_numpy(v) for k, v in case_params(d).items()}
    agent.policy.load_state_dict(sd)
    agent.policy_old.load_state_dict(sd)
    # identical states on both ranks: the sampled actions must still differ (rank folded into the key)
    probe = torch.from_numpy(np.random.default_rng(0).standard_normal((256, int(d["S"]))).astype(np.float32))
    probe_actions = agent.select_action(probe.cuda()).cpu().numpy()
    agent.buffer.clear()
    # a shard must end at an episode boundary of the flat sequence for its scan to be independent:
    # scan the whole sequence with the oracle and hand each rank rewards that reproduce its slice
    ret64 = O.returns_scan(d["rewards"], d["terminals"], float(d["gamma"]))
    term = d["terminals"].copy()
    rew = d["rewards"].copy()
    if hi < T:                                            # carry entering this shard from the right
        rew[hi - 1] = ret64[hi - 1]
        term[hi - 1] = True
    agent.buffer.load_rollout(d["states"][lo:hi], d["actions"][lo:hi], d["old_logprobs"][lo:hi],
                              d["old_values"][lo:hi], rew[lo:hi], term[lo:hi])
    agent.update()
    torch.cuda.synchronize()
    q.put((rank, agent.policy.flat.cpu().numpy(), agent.last_loss.cpu().numpy(), probe_actions))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("case", ["c2_lunarlander", "edge_std01"])
def test_two_rank_update_equals_single_buffer(eng, case):
    """Two processes (gloo rendezvous, both on this GPU) each own a shard of the rollout and run
    the engine's real update(): merged moments + all-reduced gradients must reproduce the
    reference's single-buffer result (golden) and leave identical parameters on both ranks."""
    import torch.multiprocessing as mp
    d = load_case(case)
    K = int(d["K_list"][1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, case, K, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=110) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])          # replicas stay bit-identical
    assert (res[0][3] != res[1][3]).mean() > 0.3          # but their exploration noise is not shared
    got = O.unflatten_params(res[0][1], int(d["S"]), int(d["A"]))
    worst = max(max_abs(got[k], d[f"K{K}/param/{k}"]) for k in O.PARAM_KEYS)
    assert worst < TOL_PARAM, worst
    assert rel_err(res[0][2], d[f"K{K}/loss_mean_all"]) < 1e-4


def test_bench_two_rank_line_is_self_checking(eng):
    """bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one rank per
    process; here two ranks share this GPU over gloo): rank 0 prints ONE JSON line whose `distributed`
    entry records the backend, the world size, the all-reduce time, and that parameters and loss are
    bit-identical on every rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29300 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--k-epochs", "3", "--n-envs", "256", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=dict(os.environ, PPO_BENCH_BACKEND="gloo", PPO_BENCH_WATCHDOG="100"),
                       capture_output=True, text=True, timeout=170, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 2 * 256 * 128 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    dist = out["distributed"]
    assert dist["backend"] == "gloo" and dist["world_size"] == 2
    assert dist["replica_check"]["replicas_identical"] is True
    assert dist["allreduce_us_per_epoch"] > 0 and dist["allreduce_bytes"] == 4 * (9155 + 2)
    assert out["roofline"]["launches_timed"] == 3 and "other_configs" not in out


# ------------------------------------------------------------------------------ checkpoints / schedule
def test_checkpoint_roundtrip(eng, tmp_path):
    ppo, _lib = eng
    d = np.load(os.path.join(GOLDEN, "checkpoints.npz"))
    agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 1, 0.2, False, None)
    agent.load(os.path.join(GOLDEN, "ckpt_CartPole-v1.pth"))      # written by the reference's save()
    env = "CartPole-v1"
    lp, v, ent = agent.policy_old.evaluate(d[f"{env}/states"], d[f"{env}/actions"])
    assert lp.requires_grad and v.requires_grad           # as the reference's evaluate(): differentiable
    assert rel_err(lp.detach().cpu().numpy(), d[f"{env}/logprobs"]) < TOL_REL
    assert rel_err(v.detach().cpu().numpy().reshape(-1), d[f"{env}/critic_out"]) < TOL_REL
    out = tmp_path / "x.pth"
    agent.save(str(out))
    sd = torch.load(str(out))
    ref = torch.load(os.path.join(GOLDEN, "ckpt_CartPole-v1.pth"))
    assert list(sd.keys()) == list(ref.keys())
    for k in sd:
        assert sd[k].dtype == ref[k].dtype and sd[k].shape == ref[k].shape and torch.equal(sd[k], ref[k])


def test_decay_schedule_on_device(eng, capsys):
    ppo, _ = eng
    with open(os.path.join(GOLDEN, "decay_sequence.json")) as f:
        g = json.load(f)
    agent = ppo.PPO(3, 2, 3e-4, 1e-3, 0.99, 1, 0.2, True, 0.6)
    for want, var in zip(g["std"][1:], g["var"][1:]):
        agent.decay_action_std(g["rate"], g["min"])
        assert agent.action_std == want
        assert np.array_equal(agent.policy_old.action_var.cpu().numpy(), np.asarray(var, dtype=np.float32))
    capsys.readouterr()
