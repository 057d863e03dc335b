"""CPU-only tests: the C-ABI library loads and exports what include/ppo_hip.h declares, the host
mirror of the reference surface behaves like the reference where no kernel is needed, and the
data-parallel helpers are correct over gloo with two processes.  No compute calls into the library
(there is no GPU here)."""
import ctypes
import io
import json
import os
import re
import contextlib

import numpy as np
import pytest
import torch

from helpers import GOLDEN, case_params, load_case, max_abs, rel_err
from oracle import ppo_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ppo():
    with contextlib.redirect_stdout(io.StringIO()):
        from ppo_amd import ppo
    return ppo


def test_library_exports_every_declared_symbol():
    from ppo_amd import _lib
    header = open(os.path.join(ROOT, "include", "ppo_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ppo_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in ppo_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    assert L.ppo_hip_abi_version() == _lib.ABI_VERSION
    n_actor = ctypes.c_int64()
    assert L.ppo_param_count(4, 2, ctypes.byref(n_actor)) == 9155 and n_actor.value == 4610
    assert L.ppo_param_count(26, 6, None) == 12231            # HalfCheetah shape
    assert L.ppo_workspace_bytes(4, 2, 524288) > 0
    assert b"workspace" in L.ppo_hip_strerror(-2)


def test_compute_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ppo = _ppo()
    agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 2, 0.2, False, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        agent.select_action(np.zeros(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        agent.update()


def test_state_dict_layout_and_checkpoint_format(tmp_path):
    ppo = _ppo()
    agent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 2, 0.2, False, None)
    ref = torch.load(os.path.join(GOLDEN, "ckpt_CartPole-v1.pth"))   # written by the reference's save()
    sd = agent.policy.state_dict()
    assert list(sd.keys()) == list(ref.keys()) == O.PARAM_KEYS
    for k in sd:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype
    agent.load(os.path.join(GOLDEN, "ckpt_CartPole-v1.pth"))
    # the 12 tensors are views of ONE flat vector in state_dict order
    flat = torch.cat([ref[k].reshape(-1) for k in O.PARAM_KEYS])
    assert torch.equal(agent.policy.flat.cpu(), flat) and torch.equal(agent.policy_old.flat.cpu(), flat)
    p = tmp_path / "out.pth"
    agent.save(str(p))
    back = torch.load(str(p))
    assert all(torch.equal(back[k], ref[k]) for k in ref)
    # a second shipped shape (continuous)
    cagent = ppo.PPO(15, 3, 3e-4, 1e-3, 0.99, 2, 0.2, True, 0.6)
    cagent.load(os.path.join(GOLDEN, "ckpt_RoboschoolHopper-v1.pth"))
    assert cagent.policy.flat.numel() == 64 * 15 + 64 + 4160 + 3 * 64 + 3 + 64 * 15 + 64 + 4160 + 65


def test_default_init_matches_reference_construction_order():
    """Same nn.Linear construction order as PPO.py:49-73 => same weights under the same seed
    (golden initial parameters were drawn by the reference after torch.manual_seed(rng_seed))."""
    ppo = _ppo()
    d = load_case("c2_lunarlander")
    torch.manual_seed(2)
    agent = ppo.PPO(8, 4, 3e-4, 1e-3, 0.99, 1, 0.2, False, None)
    P = case_params(d)
    for k, v in agent.policy.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), P[k]), k


def test_decay_action_std_sequence(capsys):
    ppo = _ppo()
    with open(os.path.join(GOLDEN, "decay_sequence.json")) as f:
        g = json.load(f)
    agent = ppo.PPO(3, 2, 3e-4, 1e-3, 0.99, 1, 0.2, True, 0.6)
    for want, var in zip(g["std"][1:], g["var"][1:]):
        agent.decay_action_std(g["rate"], g["min"])
        assert agent.action_std == want
        assert np.array_equal(agent.policy.action_var.cpu().numpy(), np.asarray(var, dtype=np.float32))
    out = capsys.readouterr().out
    assert "setting actor output action_std to min_action_std" in out
    agent2 = ppo.PPO(3, 2, 3e-4, 1e-3, 0.99, 1, 0.2, True, 0.6)
    for want in g["std2"][1:]:
        agent2.decay_action_std(g["rate2"], g["min2"])
        assert agent2.action_std == want
    dagent = ppo.PPO(4, 2, 3e-4, 1e-3, 0.99, 1, 0.2, False, None)
    dagent.decay_action_std(0.05, 0.1)          # discrete: warning banner, no exception (PPO.py:171-172)
    assert "WARNING : Calling PPO::decay_action_std() on discrete" in capsys.readouterr().out


def test_rollout_buffer_facade():
    ppo = _ppo()
    b = ppo.RolloutBuffer(3, 2, False, dev=torch.device("cpu"))
    assert len(b.states) == 0 and len(b.rewards) == 0
    b.rewards.append(1.5)
    b.is_terminals.append(True)
    assert list(b.rewards) == [1.5] and list(b.is_terminals) == [True] and len(b.rewards) == 1
    b.states.append(torch.tensor([1.0, 2.0, 3.0]))
    b.actions.append(torch.tensor(1))
    b.logprobs.append(torch.tensor(-0.7))
    b.state_values.append(torch.tensor([0.25]))
    assert b.states[0].tolist() == [1.0, 2.0, 3.0] and b.actions[0].dtype == torch.int64
    assert b.state_values[-1].shape == (1,)
    with pytest.raises(IndexError):
        b.states[1]
    # batched step: tensors of N values go to device columns
    b.rewards.append(torch.tensor([1.0, 2.0, 3.0, 4.0]))
    b.is_terminals.append(np.array([False, True, False, False]))
    assert len(b.rewards) == 5 and b.n_envs == 4
    b.clear()
    assert all(len(c) == 0 for c in (b.states, b.actions, b.logprobs, b.rewards, b.state_values,
                                     b.is_terminals))
    b.load_rollout(np.zeros((5, 3)), np.arange(5) % 2, np.zeros(5), np.zeros(5), [1.0] * 5, [False] * 5)
    assert len(b.states) == 5 and len(b.rewards) == 5 and b.actions[3].item() == 1
    lo = b.claim_step(7)
    assert lo == 5 and len(b.states) == 12


def test_moments_merge_matches_direct():
    from ppo_amd import dist
    rng = np.random.default_rng(0)
    parts = [rng.standard_normal(n) * 3 + 50 for n in (100, 1, 57, 1000)]
    rows = torch.tensor([[len(p), p.mean(), ((p - p.mean()) ** 2).sum()] for p in parts],
                        dtype=torch.float64)
    n, mean, m2 = dist.merge_moments_list(rows).tolist()
    allv = np.concatenate(parts)
    assert n == allv.size and abs(mean - allv.mean()) < 1e-12
    assert abs(m2 - ((allv - allv.mean()) ** 2).sum()) < 1e-8


def _dp_worker(rank, world, port, q):
    import torch.distributed as td
    from ppo_amd import dist
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    d = load_case("c2_lunarlander")
    T = int(d["T"])
    lo, hi = (0, 400) if rank == 0 else (400, T)          # uneven shards
    ret64 = O.returns_scan(d["rewards"], d["terminals"], float(d["gamma"]))
    r32 = ret64.astype(np.float32)[lo:hi].astype(np.float64)
    mom = torch.tensor([hi - lo, r32.mean(), ((r32 - r32.mean()) ** 2).sum()], dtype=torch.float64)
    merged = dist.merge_moments(mom)
    n, mean, m2 = merged.tolist()
    std = np.float32(np.sqrt(m2 / (n - 1)))
    ret_n = ((ret64.astype(np.float32)[lo:hi] - np.float32(mean)) / (std + np.float32(1e-7))).astype(np.float32)
    adv = ret_n - d["old_values"][lo:hi]
    P = case_params(d)
    out, grads = O.ppo_loss_and_grads(P, d["states"][lo:hi], d["actions"][lo:hi], d["old_logprobs"][lo:hi],
                                      adv, ret_n, float(d["eps_clip"]), False, n_global=int(n))
    flat = torch.from_numpy(O.flatten_params(grads))
    dist.all_reduce_sum(flat)
    q.put((rank, n, float(mean), float(std), flat.numpy()))
    td.barrier()
    td.destroy_process_group()


def test_two_shard_update_equals_single_buffer_gloo():
    """World size 2 over gloo: merged moments and all-reduced shard gradients (each scaled by
    1/n_global) equal the reference's single-buffer values (golden)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = load_case("c2_lunarlander")
    for rank, n, mean, std, flat in res:
        assert n == int(d["T"])
        assert rel_err(mean, d["ret_mean"], floor=float(np.abs(d["returns32"]).mean())) < 1e-5
        assert rel_err(std, d["ret_std"]) < 1e-5
        g = O.unflatten_params(flat, 8, 4)
        for k in O.PARAM_KEYS:
            want = d["grad_e1/" + k]
            assert rel_err(g[k], want, floor=float(np.abs(want).max())) < 1e-4, k
    assert np.array_equal(res[0][4], res[1][4])       # identical on both ranks


def test_c_abi_argument_errors_without_touching_the_gpu():
    """Argument validation happens before any HIP call: bad shapes, null pointers and short
    workspaces come back as negative PPO_E_* codes with a text."""
    import ctypes
    from ppo_amd import _lib
    L = _lib.lib()
    E_BADARG, E_WORKSPACE, E_UNSUPPORTED = -1, -2, -3
    na = ctypes.c_int64(0)
    assert L.ppo_param_count(4, 2, ctypes.byref(na)) == 9155 and na.value == 4610
    assert L.ppo_param_count(26, 6, None) == 12231
    assert L.ppo_workspace_bytes(4, 2, 1600) > 0
    # state_dim / action_dim beyond the supported range
    assert L.ppo_act(1, 64, 2, 0, 0.0, 1, 8, 0, 0, 1, 1, 1, None) == E_UNSUPPORTED
    assert L.ppo_act(1, 4, 9, 0, 0.0, 1, 8, 0, 0, 1, 1, 1, None) == E_UNSUPPORTED
    # null pointers / empty batch
    assert L.ppo_act(None, 4, 2, 0, 0.0, 1, 8, 0, 0, 1, 1, 1, None) == E_BADARG
    assert L.ppo_act(1, 4, 2, 0, 0.0, 1, 0, 0, 0, 1, 1, 1, None) == E_BADARG
    assert L.ppo_returns_scan_flat(None, 1, 1, 8, 0.99, 1, 1, 1, 1 << 20, None) == E_BADARG
    # workspace missing or too small
    assert L.ppo_epoch_fwdbwd(1, 4, 2, 0, 0.0, 1, 1, 1, 1, 1, 8, 8, 0.2, None, None, None, None,
                              None, 0, None) == E_WORKSPACE
    assert L.ppo_epoch_fwdbwd(1, 4, 2, 0, 0.0, 1, 1, 1, 1, 1, 8, 8, 0.2, None, None, None, None,
                              1, 16, None) == E_WORKSPACE
    assert L.ppo_epoch_fwdbwd(1, 64, 2, 0, 0.0, 1, 1, 1, 1, 1, 8, 8, 0.2, None, None, None, None,
                              1, 1 << 30, None) == E_UNSUPPORTED
    assert L.ppo_epoch_fwdbwd(1, 4, 2, 0, 0.0, 1, 1, 1, 1, 1, 8, 4, 0.2, None, None, None, None,
                              1, 1 << 30, None) == E_BADARG          # n_global < n_local
    assert L.ppo_grad_reduce(4, 2, 1, 16, 1, None) == E_WORKSPACE
    assert L.ppo_adam_step(1, 1, 1, 1, 0, 0, 1, 3e-4, 1e-3, 0.9, 0.999, 1e-8, None, None) == E_BADARG
    assert L.ppo_adam_step(1, 1, 1, 1, 10, 5, 0, 3e-4, 1e-3, 0.9, 0.999, 1e-8, None, None) == E_BADARG
    assert L.ppo_cartpole_step(None, 1, 1, None, None, 1, 1, 1, 1, 4, 500, 0, None) == E_BADARG
    for code in (E_BADARG, E_WORKSPACE, E_UNSUPPORTED):
        assert len(L.ppo_hip_strerror(code)) > 5
    with pytest.raises(RuntimeError):
        _lib.check(E_UNSUPPORTED)


def test_public_header_is_plain_c(tmp_path):
    """include/ppo_hip.h is the C-ABI contract: it must compile as C (no C++, no HIP, no torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "ppo_hip.h"\n'
                   'int probe(void) { return ppo_hip_abi_version() + (int)sizeof(&ppo_returns_scan_flat_noalloc)'
                   ' + PPO_SCAN_NOALLOC_MAX / PPO_SCAN_NOALLOC_MAX; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_resume_checkpoint_round_trip(tmp_path):
    """save_resume / load_resume carry what the reference's checkpoint leaves out (PPO.py:253-258 saves
    policy_old only): both policies, Adam's moments and step count, the learning rates, action_std and
    the sampling stream's position.  No compute involved: runs without a GPU."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from ppo_amd import ppo
        a = ppo.PPO(11, 3, 3e-4, 1e-3, 0.99, 5, 0.2, True, 0.6)
        b = ppo.PPO(11, 3, 1e-4, 2e-3, 0.99, 5, 0.2, True, 0.3)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        a.policy.flat.copy_(torch.randn(a.policy.flat.shape, generator=g))
        a.policy_old.flat.copy_(torch.randn(a.policy.flat.shape, generator=g))
        a.optimizer.exp_avg.copy_(torch.randn(a.policy.flat.shape, generator=g))
        a.optimizer.exp_avg_sq.copy_(torch.rand(a.policy.flat.shape, generator=g))
    a.optimizer.step_count = 123
    with contextlib.redirect_stdout(io.StringIO()):
        a.decay_action_std(0.05, 0.1)
    a._rng_seed, a._rng_offset = 987654321, 4242
    path = tmp_path / "resume.pt"
    a.save_resume(str(path))
    b.load_resume(str(path))
    assert torch.equal(b.policy.flat.cpu(), a.policy.flat.cpu()) and torch.equal(b.policy_old.flat.cpu(), a.policy_old.flat.cpu())
    assert torch.equal(b.optimizer.exp_avg.cpu(), a.optimizer.exp_avg.cpu())
    assert torch.equal(b.optimizer.exp_avg_sq.cpu(), a.optimizer.exp_avg_sq.cpu())
    assert b.optimizer.step_count == 123 and [g_["lr"] for g_ in b.optimizer.param_groups] == [3e-4, 1e-3]
    assert b.action_std == a.action_std == 0.55 and b.policy_old._var_host == a.policy_old._var_host
    assert (b._rng_seed, b._rng_offset) == (987654321, 4242)
    # the reference-format file is untouched by all this: 12 tensors of policy_old
    a.save(str(tmp_path / "ref.pth"))
    assert len(torch.load(str(tmp_path / "ref.pth"))) == 12
    with pytest.raises(ValueError):
        b.load_resume(str(tmp_path / "ref.pth"))


def test_fastcall_extension_marshals_like_the_ctypes_binding():
    """The ctypes-free per-step binding (ppo_amd/csrc/fastcall.c), driven on the CPU with a stand-in for
    ppo_act_session_step_bound: float64 / float32 vectors land in the state buffer as float32, row and
    offset arrive unchanged (64 bits), the discrete action comes back as an int (None for a continuous
    policy), a library error code is raised, and anything it cannot take (wrong length, list, strided,
    2-D, integer dtype) is declined with NotImplemented so that the caller's ctypes path runs instead."""
    import ctypes
    from ppo_amd import _fastcall
    calls = []
    proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64)
    stand_in = proto(lambda sess, row, off: (calls.append((sess, row, off)), -2 if row == 7 else 0)[1])
    fn = ctypes.cast(stand_in, ctypes.c_void_p).value
    state = np.zeros(4, dtype=np.float32)
    act = np.array([5, 0], dtype=np.int32)
    st = _fastcall.Stepper(fn, 1234, state.ctypes.data, act.ctypes.data, 4, False)
    assert st.step(np.array([1.0, 2.0, 3.0, 0.1]), 3, 99) == 5
    assert np.array_equal(state, np.array([1.0, 2.0, 3.0, 0.1], dtype=np.float32)) and calls[-1] == (1234, 3, 99)
    assert st.step(np.array([4, 3, 2, 1], dtype=np.float32), 2 ** 40, 2 ** 63 + 5) == 5
    assert np.array_equal(state, [4, 3, 2, 1]) and calls[-1] == (1234, 2 ** 40, 2 ** 63 + 5)
    n = len(calls)
    for bad in (np.zeros(3), [1, 2, 3, 4], np.zeros((4, 1)), np.zeros(8)[::2], np.zeros(4, dtype=np.int64)):
        assert st.step(bad, 0, 0) is NotImplemented
    assert len(calls) == n
    with pytest.raises(RuntimeError) as e:
        st.step(np.zeros(4), 7, 0)
    assert e.value.args == (-2,)
    cont = _fastcall.Stepper(fn, 1, state.ctypes.data, act.ctypes.data, 4, True)
    assert cont.step(np.zeros(4), 1, 1) is None
    with pytest.raises(ValueError):
        _fastcall.Stepper(0, 1, 2, 3, 4, False)
