#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference implementation.

Runs only in the build container (needs /root/reference, which never travels).
Nothing here restates the reference's arithmetic: every expected value is
captured from the reference's own execution of ``PPO.select_action`` /
``PPO.update`` / ``PPO.decay_action_std`` / ``PPO.save`` (PPO.py:124-258), by
  * a TorchFunctionMode that records the tensors flowing through the calls the
    reference makes (``torch.tensor`` of the return list, ``mean``/``std``,
    ``exp``, ``clamp``, ``min``, the per-sample loss handed to ``.mean()``),
  * a wrapper around ``policy.evaluate`` that records its outputs, and
  * a wrapper around ``optimizer.step`` that records ``.grad`` before the step
    and Adam's state after it.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [name ...]
        (no names: everything; names select single outputs, e.g. ``c3_bipedal spread two_update``)
Output: tests/golden/*.npz, tests/golden/decay_sequence.json, tests/golden/reference_spread.json,
        tests/golden/ckpt_*.pth (written by the reference's own ``save``).
"""
import copy
import io
import json
import os
import sys
import contextlib

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import PPO as ref  # noqa: E402  (the reference module, PPO.py)

torch.set_num_threads(1)  # fixed CPU reduction order for the captured values


class Recorder(TorchFunctionMode):
    """Records (name, args-as-numpy, result-as-numpy) of selected torch calls."""

    WATCH = {"tensor", "mean", "std", "exp", "clamp", "min", "sub", "__sub__",
             "rsub", "__rsub__", "mul", "__mul__"}

    def __init__(self):
        super().__init__()
        self.calls = []

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        name = getattr(func, "__name__", str(func))
        if name in self.WATCH:
            def conv(a):
                if isinstance(a, torch.Tensor):
                    return a.detach().cpu().numpy().copy()
                return a
            self.calls.append((name, [conv(a) for a in args], conv(out)))
        return out


def make_agent(S, A, K, continuous, std, lr_a=3e-4, lr_c=1e-3, gamma=0.99, eps=0.2, seed=0):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = ref.PPO(S, A, lr_a, lr_c, gamma, K, eps, continuous, std)
    return agent


def fill_buffer(agent, S, T, rng, p_term, reward_kind, logp_noise=0.0, term_mask=None):
    """Drive the reference's own select_action over seeded synthetic states."""
    states = rng.standard_normal((T, S)).astype(np.float32)
    for t in range(T):
        agent.select_action(states[t].astype(np.float64))
        if reward_kind == "one":
            r = 1.0
        else:  # fp32-representable gaussian rewards
            r = float(np.float32(rng.standard_normal()))
        agent.buffer.rewards.append(r)
        if term_mask is not None:
            agent.buffer.is_terminals.append(bool(term_mask[t]))
        else:
            agent.buffer.is_terminals.append(bool(rng.random() < p_term))
    if logp_noise > 0.0:
        # push many ratios outside the clip interval (exercises the zero-gradient branch)
        for t in range(T):
            agent.buffer.logprobs[t] = agent.buffer.logprobs[t] + float(
                np.float32(rng.uniform(-logp_noise, logp_noise)))


def snapshot_inputs(agent, S, A, continuous):
    d = {}
    for k, v in agent.policy.state_dict().items():
        d["param/" + k] = v.detach().cpu().numpy().copy()
    b = agent.buffer
    d["states"] = torch.stack(b.states, 0).reshape(len(b.states), S).numpy().copy()
    if continuous:
        d["actions"] = torch.stack(b.actions, 0).reshape(len(b.actions), A).numpy().copy()
    else:
        d["actions"] = torch.stack(b.actions, 0).reshape(-1).numpy().astype(np.int64)
    d["old_logprobs"] = torch.stack(b.logprobs, 0).reshape(-1).numpy().copy()
    d["old_values"] = torch.stack(b.state_values, 0).reshape(-1).numpy().copy()
    d["rewards"] = np.asarray(b.rewards, dtype=np.float64)
    d["terminals"] = np.asarray(b.is_terminals, dtype=np.bool_)
    return d


def clone_agent(agent, S, A, K, continuous, std, lr_a, lr_c, gamma, eps):
    """Fresh reference agent with identical weights, Adam state and buffer."""
    with contextlib.redirect_stdout(io.StringIO()):
        new = ref.PPO(S, A, lr_a, lr_c, gamma, K, eps, continuous, std)
    new.policy.load_state_dict(copy.deepcopy(agent.policy.state_dict()))
    new.policy_old.load_state_dict(copy.deepcopy(agent.policy_old.state_dict()))
    new.optimizer.load_state_dict(copy.deepcopy(agent.optimizer.state_dict()))
    if continuous:
        new.action_std = agent.action_std
        new.policy.action_var = agent.policy.action_var.clone()
        new.policy_old.action_var = agent.policy_old.action_var.clone()
    b = agent.buffer
    new.buffer.states = [x.clone() for x in b.states]
    new.buffer.actions = [x.clone() for x in b.actions]
    new.buffer.logprobs = [x.clone() for x in b.logprobs]
    new.buffer.state_values = [x.clone() for x in b.state_values]
    new.buffer.rewards = list(b.rewards)
    new.buffer.is_terminals = list(b.is_terminals)
    return new


def run_update_recorded(agent):
    """Run the reference's update() once; return everything observed during it."""
    rec = Recorder()
    evals, steps = [], []
    orig_eval = agent.policy.evaluate
    orig_step = agent.optimizer.step
    names = [k for k, _ in agent.policy.named_parameters()]

    def eval_wrap(state, action):
        out = orig_eval(state, action)
        evals.append(tuple(o.detach().cpu().numpy().copy() for o in out))
        return out

    def step_wrap(*a, **k):
        grads = {n: p.grad.detach().cpu().numpy().copy()
                 for n, p in agent.policy.named_parameters()}
        r = orig_step(*a, **k)
        st = agent.optimizer.state
        adam = {}
        for n, p in agent.policy.named_parameters():
            adam[n] = (st[p]["exp_avg"].numpy().copy(), st[p]["exp_avg_sq"].numpy().copy(),
                       float(st[p]["step"]))
        steps.append((grads, adam))
        return r

    agent.policy.evaluate = eval_wrap
    agent.optimizer.step = step_wrap
    with rec:
        agent.update()
    agent.policy.evaluate = orig_eval
    agent.optimizer.step = orig_step
    return rec.calls, evals, steps, names


def extract(calls, evals, steps, T):
    """Pick the recorded tensors by the order in which update() (PPO.py:200-251) makes the calls."""
    out = {}
    # PPO.py:211  torch.tensor(list_of_python_floats, dtype=float32)
    tcalls = [c for c in calls if c[0] == "tensor"]
    out["returns64"] = np.asarray(tcalls[0][1][0], dtype=np.float64)
    out["returns32"] = tcalls[0][2].astype(np.float32)
    # PPO.py:212  rewards.mean(), rewards.std()
    means = [c for c in calls if c[0] == "mean"]
    stds = [c for c in calls if c[0] == "std"]
    out["ret_mean"] = np.float32(means[0][2])
    out["ret_std"] = np.float32(stds[0][2])
    # PPO.py:240/244  loss.mean(): remaining mean() calls with a [T] argument, one per epoch
    loss_calls = [c for c in means[1:] if np.ndim(c[1][0]) == 1 and c[1][0].shape[0] == T]
    out["loss_per_sample_e1"] = loss_calls[0][1][0]
    out["loss_mean_e1"] = np.float32(loss_calls[0][2])
    out["loss_mean_all"] = np.asarray([c[2] for c in loss_calls], dtype=np.float32)
    # PPO.py:233 torch.exp(logprobs - old_logprobs)
    exps = [c for c in calls if c[0] == "exp" and np.ndim(c[2]) == 1 and c[2].shape[0] == T]
    out["ratios_e1"] = exps[0][2]
    # PPO.py:227 evaluate outputs, first epoch
    lp, sv, ent = evals[0]
    out["logprobs_e1"] = lp.reshape(-1)
    out["state_values_e1"] = sv.reshape(-1)
    out["entropy_e1"] = np.broadcast_to(ent, (T,)).astype(np.float32).copy()
    # PPO.py:236 ratios * advantages -> second operand of the first [T]x[T] mul after exp
    # PPO.py:240 MseLoss(state_values, rewards): normalised returns are recovered from the
    # sub call rewards - rewards.mean() divided ... simpler: advantages + old values (exact? no) ->
    # take them from the recorded sub/div chain instead:
    subs = [c for c in calls if c[0] in ("sub", "__sub__")]
    # first sub: rewards - mean (PPO.py:212); second: rewards_norm.detach() - old_values (PPO.py:221)
    adv_call = None
    for c in subs:
        a0, a1 = c[1][0], c[1][1]
        if isinstance(a0, np.ndarray) and isinstance(a1, np.ndarray) and a0.shape == (T,) and a1.shape == (T,):
            adv_call = c
            break
    out["returns_norm"] = adv_call[1][0].astype(np.float32)
    out["advantages"] = adv_call[2].astype(np.float32)
    for n, g in steps[0][0].items():
        out["grad_e1/" + n] = g
    return out


def case(name, S, A, T, K_list, continuous, std, rng_seed, p_term=1 / 200, reward_kind="gauss",
         logp_noise=0.0, term_mask=None, weight_scale=None, lr_a=3e-4, lr_c=1e-3, gamma=0.99,
         eps=0.2):
    rng = np.random.default_rng(rng_seed)
    base = make_agent(S, A, K_list[0], continuous, std, lr_a, lr_c, gamma, eps, seed=rng_seed)
    if weight_scale is not None:  # saturate the softmax / tanh head
        with torch.no_grad():
            base.policy.actor[4].weight.mul_(weight_scale)
            base.policy_old.load_state_dict(base.policy.state_dict())
    fill_buffer(base, S, T, rng, p_term, reward_kind, logp_noise, term_mask)
    d = snapshot_inputs(base, S, A, continuous)
    d.update(dict(S=S, A=A, T=T, continuous=continuous, action_std=np.float64(std or 0.0),
                  gamma=np.float64(gamma), eps_clip=np.float64(eps), lr_actor=np.float64(lr_a),
                  lr_critic=np.float64(lr_c), K_list=np.asarray(K_list),
                  torch_version=str(torch.__version__)))
    for K in K_list:
        ag = clone_agent(base, S, A, K, continuous, std, lr_a, lr_c, gamma, eps)
        calls, evals, steps, names = run_update_recorded(ag)
        if K == K_list[0]:
            d.update(extract(calls, evals, steps, T))
        means = [c for c in calls if c[0] == "mean"]
        loss_calls = [c for c in means[1:] if np.ndim(c[1][0]) == 1 and c[1][0].shape[0] == T]
        d[f"K{K}/loss_mean_all"] = np.asarray([c[2] for c in loss_calls], dtype=np.float32)
        for n, p in ag.policy.state_dict().items():
            d[f"K{K}/param/{n}"] = p.numpy().copy()
        for n, p in ag.policy_old.state_dict().items():
            assert torch.equal(p, ag.policy.state_dict()[n])  # PPO.py:248
        adam = steps[-1][1]
        for n, (m, v, st) in adam.items():
            d[f"K{K}/exp_avg/{n}"] = m
            d[f"K{K}/exp_avg_sq/{n}"] = v
            d[f"K{K}/step"] = np.float64(st)
        assert len(ag.buffer.states) == 0 and len(ag.buffer.rewards) == 0  # PPO.py:251
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: T={T} S={S} A={A} K={K_list} -> {os.path.getsize(path) / 1024:.0f} KiB")


def scan_cases():
    """Return-scan goldens over many lengths / terminal masks (PPO.py:202-212)."""
    d = {}
    rng = np.random.default_rng(7)
    specs = []
    for T in (2, 3, 63, 64, 65, 255, 256, 257, 1000, 1600, 4099):
        specs.append((T, "rand"))
    specs += [(257, "none"), (257, "all"), (300, "first"), (300, "last"), (300, "both")]
    for i, (T, kind) in enumerate(specs):
        if kind == "rand":
            mask = rng.random(T) < (1 / 20)
        elif kind == "none":
            mask = np.zeros(T, bool)
        elif kind == "all":
            mask = np.ones(T, bool)
        else:
            mask = rng.random(T) < (1 / 50)
            if kind in ("first", "both"):
                mask[0] = True
            if kind in ("last", "both"):
                mask[-1] = True
        gamma = [0.99, 0.999, 0.9, 1.0][i % 4]
        ag = make_agent(2, 2, 1, False, None, gamma=gamma, seed=i)
        fill_buffer(ag, 2, T, rng, 0.0, "gauss", term_mask=mask)
        rewards = np.asarray(ag.buffer.rewards, dtype=np.float64)
        calls, evals, steps, _ = run_update_recorded(ag)
        ex = extract(calls, evals, steps, T)
        d[f"{i}/gamma"] = np.float64(gamma)
        d[f"{i}/rewards"] = rewards
        d[f"{i}/terminals"] = mask
        for k in ("returns64", "returns32", "ret_mean", "ret_std", "returns_norm"):
            d[f"{i}/{k}"] = ex[k]
    d["n_cases"] = len(specs)
    d["torch_version"] = str(torch.__version__)
    path = os.path.join(HERE, "scan_cases.npz")
    np.savez_compressed(path, **d)
    print(f"scan_cases: {len(specs)} cases -> {os.path.getsize(path) / 1024:.0f} KiB")


def decay_sequence():
    """decay_action_std 0.6 -> 0.1 by 0.05 incl. the floor (PPO.py:159-173)."""
    ag = make_agent(3, 2, 1, True, 0.6)
    seq, var = [ag.action_std], [ag.policy.action_var.numpy().astype(np.float32).tolist()]
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(14):
            ag.decay_action_std(0.05, 0.1)
            seq.append(ag.action_std)
            var.append(ag.policy.action_var.numpy().astype(np.float32).tolist())
    ag2 = make_agent(3, 2, 1, True, 0.6)
    seq2 = [ag2.action_std]
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(8):
            ag2.decay_action_std(0.0733, 0.123)
            seq2.append(ag2.action_std)
    with open(os.path.join(HERE, "decay_sequence.json"), "w") as f:
        json.dump({"rate": 0.05, "min": 0.1, "std": seq, "var": var,
                   "rate2": 0.0733, "min2": 0.123, "std2": seq2}, f, indent=1)
    print("decay_sequence:", seq)


def checkpoints():
    """Shipped checkpoints: reference forward outputs + a file written by reference save()."""
    shapes = {"CartPole-v1": (4, 2, False), "LunarLander-v2": (8, 4, False),
              "BipedalWalker-v2": (24, 4, True), "RoboschoolHalfCheetah-v1": (26, 6, True),
              "RoboschoolHopper-v1": (15, 3, True), "RoboschoolWalker2d-v1": (22, 6, True)}
    d = {}
    rng = np.random.default_rng(11)
    for env, (S, A, cont) in shapes.items():
        ag = make_agent(S, A, 1, cont, 0.6 if cont else None)
        ag.load(f"{REF}/PPO_preTrained/{env}/PPO_{env}_0_0.pth")
        x = rng.standard_normal((37, S)).astype(np.float32)
        for k, v in ag.policy_old.state_dict().items():
            d[f"{env}/param/{k}"] = v.numpy().copy()
        with torch.no_grad():
            xt = torch.from_numpy(x)
            d[f"{env}/states"] = x
            d[f"{env}/actor_out"] = ag.policy_old.actor(xt).numpy()
            d[f"{env}/critic_out"] = ag.policy_old.critic(xt).numpy().reshape(-1)
            if cont:
                acts = rng.standard_normal((37, A)).astype(np.float32)
            else:
                acts = rng.integers(0, A, 37)
            lp, sv, ent = ag.policy_old.evaluate(xt, torch.from_numpy(acts))
            d[f"{env}/actions"] = acts
            d[f"{env}/logprobs"] = lp.numpy()
            d[f"{env}/entropy"] = np.broadcast_to(ent.numpy(), (37,)).copy()
        d[f"{env}/S"], d[f"{env}/A"], d[f"{env}/continuous"] = S, A, cont
        if env in ("CartPole-v1", "RoboschoolHopper-v1"):
            ag.save(os.path.join(HERE, f"ckpt_{env}.pth"))  # reference's own save(), PPO.py:253
    d["envs"] = np.asarray(list(shapes.keys()))
    np.savez_compressed(os.path.join(HERE, "checkpoints.npz"), **d)
    print("checkpoints: 6 envs")


def agent_from_arrays(d, K, order=None, returns_as_rewards=False):
    """A reference agent rebuilt from a committed golden's INPUTS (initial weights + buffer), with
    the buffer elements in the shapes the reference's own select_action stores (PPO.py:182-185).
    order: optional permutation of the samples.  returns_as_rewards: the buffer's rewards are the
    fp64 Monte-Carlo returns and every step is flagged terminal, so that PPO.py:202-208 reproduces
    exactly those returns whatever the sample order."""
    S, A, cont = int(d["S"]), int(d["A"]), bool(d["continuous"])
    std = float(d["action_std"]) if cont else None
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        ag = ref.PPO(S, A, float(d["lr_actor"]), float(d["lr_critic"]), float(d["gamma"]), K,
                     float(d["eps_clip"]), cont, std)
    sd = {k[len("param/"):]: torch.from_numpy(np.array(v)) for k, v in d.items() if k.startswith("param/")}
    ag.policy.load_state_dict(sd)
    ag.policy_old.load_state_dict(sd)
    T = int(d["T"])
    idx = np.arange(T) if order is None else np.asarray(order)
    rewards = d["returns64"] if returns_as_rewards else d["rewards"]
    for t in idx:
        ag.buffer.states.append(torch.from_numpy(d["states"][t].copy()))
        if cont:
            ag.buffer.actions.append(torch.from_numpy(d["actions"][t].copy()).reshape(1, A))
            ag.buffer.logprobs.append(torch.tensor([d["old_logprobs"][t]]))
        else:
            ag.buffer.actions.append(torch.tensor(int(d["actions"][t])))
            ag.buffer.logprobs.append(torch.tensor(d["old_logprobs"][t]))
        ag.buffer.state_values.append(torch.tensor([d["old_values"][t]]))
        ag.buffer.rewards.append(float(rewards[t]))
        ag.buffer.is_terminals.append(True if returns_as_rewards else bool(d["terminals"][t]))
    return ag


def reference_spread(names=("c1_cartpole", "c2_lunarlander", "c3_bipedal"), n_perm=40):
    """What the REFERENCE itself produces after K_cfg epochs on identical data when only the order
    of its fp32 summations changes: (a) its update() under 4 / 8 intra-op threads instead of 1,
    (b) its update() on the same samples in reversed and in n_perm random orders (rewards := the
    recorded returns, every step terminal, so the return of each sample is unchanged).
    Every run is compared epoch by epoch with the golden run's parameter trajectory.  The runs
    agree to a few 1e-5 for tens of epochs and then separate in discrete jumps onto a small number
    of distinct end points ("branches": Adam normalises gradient rows that are at fp32 noise level
    into lr-sized steps, so which side of zero / of an exact saturation a handful of terms falls on
    decides whole weight rows).  Written:
      reference_spread.json   per run: final distance from the golden run, first epoch > 1e-4 away;
                              per case: diameter of the set of end points, number of branches;
      <case>_branches.npz     the distinct end points (flat parameter vectors, 2e-5 apart or more):
                              a correct implementation in another summation order must land within
                              1e-4 of ONE of them; with the optimizer's exp_avg / exp_avg_sq there;
      <case>_traj.npz         the golden run's flat parameter vector after selected epochs, and per
                              epoch the largest distance of any run from the golden run."""
    out = {"torch_version": str(torch.__version__), "cases": {}}
    for name in names:
        z = np.load(os.path.join(HERE, name + ".npz"))
        d = {k: z[k] for k in z.files}
        K = int(d["K_list"][-1])
        T = int(d["T"])
        keys = [k[len("param/"):] for k in d if k.startswith("param/")]
        want = np.concatenate([d[f"K{K}/param/{k}"].reshape(-1) for k in keys])

        last_state = [None]

        def run(threads, **kw):
            torch.set_num_threads(threads)
            ag = agent_from_arrays(d, K, **kw)
            traj = []
            orig = ag.optimizer.step

            def step_wrap(*a, **k):
                r = orig(*a, **k)
                traj.append(torch.cat([p.detach().reshape(-1) for p in ag.policy.state_dict().values()])
                            .numpy().copy())
                return r

            ag.optimizer.step = step_wrap
            with contextlib.redirect_stdout(io.StringIO()):
                ag.update()
            torch.set_num_threads(1)
            # Adam's moments at the end point, flat in the optimizer's parameter order (= state_dict order)
            ps = [p for g in ag.optimizer.param_groups for p in g["params"]]
            last_state[0] = tuple(torch.cat([ag.optimizer.state[p][kind].detach().reshape(-1) for p in ps])
                                  .numpy().copy() for kind in ("exp_avg", "exp_avg_sq"))
            return np.stack(traj)

        def run_s(threads, **kw):            # (trajectory, Adam moments at its end)
            tr = run(threads, **kw)
            return tr, last_state[0]

        base, base_state = run_s(1)         # rebuilding the agent from the arrays reproduces the golden
        assert np.array_equal(base[-1], want), name
        runs = {"threads4": run_s(4), "threads8": run_s(8),
                "returns_as_rewards_same_order": run_s(1, returns_as_rewards=True),
                "reversed_order": run_s(1, order=np.arange(T)[::-1], returns_as_rewards=True)}
        rng = np.random.default_rng(1000 + T)
        for i in range(n_perm):
            runs[f"permutation{i}"] = run_s(1, order=rng.permutation(T), returns_as_rewards=True)
        branch_states = [base_state]
        rec = {"K": K, "T": T, "runs": {}}
        branches, counts = [base[-1]], [1]
        curve = np.zeros(K)
        for rn, (tr, st) in runs.items():
            dev = np.abs(tr - base).max(axis=1)                     # per epoch
            curve = np.maximum(curve, dev)
            over = np.nonzero(dev > 1e-4)[0]
            rec["runs"][rn] = {"final": float(dev[-1]),
                               "first_epoch_over_1e-4": int(over[0]) + 1 if over.size else None,
                               "max_over_first_4_epochs": float(dev[:4].max())}
            dist = [float(np.abs(tr[-1] - b).max()) for b in branches]
            if min(dist) > 2e-5:
                branches.append(tr[-1])
                branch_states.append(st)
                counts.append(1)
            else:
                counts[int(np.argmin(dist))] += 1
        assert rec["runs"]["returns_as_rewards_same_order"]["final"] == 0.0
        rec["spread"] = max(r["final"] for r in rec["runs"].values())
        fin = np.stack(branches)
        rec["diameter"] = float(np.abs(fin[:, None, :] - fin[None, :, :]).max())
        rec["n_branches"] = len(branches)
        rec["branch_counts"] = counts
        out["cases"][name] = rec
        # exp_avg / exp_avg_sq: the reference optimizer's moments at each end point (of the first run
        # that reached it), so that an engine that lands on branch i can be checked against ITS state
        np.savez_compressed(os.path.join(HERE, name + "_branches.npz"), finals=fin,
                            counts=np.asarray(counts),
                            exp_avg=np.stack([s_[0] for s_ in branch_states]),
                            exp_avg_sq=np.stack([s_[1] for s_ in branch_states]))
        snaps = sorted({k for k in (1, 2, 4, 8, 16, 24, 32, 48, 64, K) if k <= K})
        # spread_curve[e-1]: largest distance of any of the runs above from the golden run after epoch e
        np.savez_compressed(os.path.join(HERE, name + "_traj.npz"), epochs=np.asarray(snaps),
                            spread_curve=curve, **{f"K{k}": base[k - 1] for k in snaps})
        print("spread", name, {k: v for k, v in rec.items() if k != "runs"},
              sorted(round(r["final"] * 1e4, 2) for r in rec["runs"].values()))
    with open(os.path.join(HERE, "reference_spread.json"), "w") as f:
        json.dump(out, f, indent=1)


def two_update(name, S, A, T, K, continuous, std, rng_seed, decay=None):
    """Two consecutive update() calls on ONE agent (update, refill with the new policy_old, update):
    Adam's step count and moments persist across updates (PPO.py:139 is the only construction).
    Records both rollouts as inputs and parameters + Adam state after each update; with `decay`
    = (rate, min) the std schedule (PPO.py:159-173) runs between the two rollouts."""
    rng = np.random.default_rng(rng_seed)
    ag = make_agent(S, A, K, continuous, std, seed=rng_seed)
    d = dict(S=S, A=A, T=T, K=K, continuous=continuous, action_std=np.float64(std or 0.0),
             gamma=np.float64(0.99), eps_clip=np.float64(0.2), lr_actor=np.float64(3e-4),
             lr_critic=np.float64(1e-3), torch_version=str(torch.__version__))
    for k, v in ag.policy.state_dict().items():
        d["param/" + k] = v.detach().cpu().numpy().copy()
    for u in (1, 2):
        fill_buffer(ag, S, T, rng, 1 / 100, "gauss")
        for k, v in snapshot_inputs(ag, S, A, continuous).items():
            if not k.startswith("param/"):
                d[f"u{u}/{k}"] = v
        d[f"u{u}/action_std"] = np.float64(ag.action_std if continuous else 0.0)
        calls, evals, steps, names = run_update_recorded(ag)
        for n, p in ag.policy.state_dict().items():
            d[f"u{u}/after/param/{n}"] = p.numpy().copy()
        for n, (m, v, st) in steps[-1][1].items():
            d[f"u{u}/after/exp_avg/{n}"] = m
            d[f"u{u}/after/exp_avg_sq/{n}"] = v
            d[f"u{u}/after/step"] = np.float64(st)
        means = [c for c in calls if c[0] == "mean"]
        loss_calls = [c for c in means[1:] if np.ndim(c[1][0]) == 1 and c[1][0].shape[0] == T]
        d[f"u{u}/loss_mean_all"] = np.asarray([c[2] for c in loss_calls], dtype=np.float32)
        if u == 1 and decay is not None:
            with contextlib.redirect_stdout(io.StringIO()):
                ag.decay_action_std(*decay)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: two updates of T={T} K={K} -> {os.path.getsize(path) / 1024:.0f} KiB, "
          f"step={d['u2/after/step']}")


TARGETS = {
    # C1 CartPole-shaped (train.py:19-48 defaults), C2 LunarLander-shaped, C3 BipedalWalker-shaped
    # (update_timestep = 4 * max_ep_len = 6000, PPO_preTrained/README.md)
    "c1_cartpole": lambda: case("c1_cartpole", 4, 2, 1600, [1, 4, 80], False, None, 1, p_term=1 / 100,
                                reward_kind="one"),
    "c2_lunarlander": lambda: case("c2_lunarlander", 8, 4, 900, [1, 4, 30], False, None, 2),
    "c3_bipedal": lambda: case("c3_bipedal", 24, 4, 6000, [1, 4, 80], True, 0.6, 3, p_term=1 / 400),
    "c5_halfcheetah_small": lambda: case("c5_halfcheetah_small", 26, 6, 2048, [1, 4], True, 0.6, 4,
                                         p_term=0.0, term_mask=(np.arange(2048) % 128 == 127)),
    # edge cases
    "edge_a1_continuous": lambda: case("edge_a1_continuous", 3, 1, 257, [1, 4], True, 0.5, 5),
    "edge_offclip": lambda: case("edge_offclip", 4, 2, 515, [1, 4], False, None, 6, logp_noise=0.6),
    "edge_saturated_softmax": lambda: case("edge_saturated_softmax", 6, 3, 333, [1, 4], False, None, 8,
                                           weight_scale=60.0),
    "edge_std01": lambda: case("edge_std01", 5, 2, 130, [1, 2], True, 0.1, 9),
    "scan_cases": scan_cases,
    "decay_sequence": decay_sequence,
    "checkpoints": checkpoints,
    # Adam state persisting over two update() calls; the continuous one decays the std in between
    "two_update": lambda: (two_update("two_update_discrete", 4, 2, 700, 10, False, None, 21),
                           two_update("two_update_continuous", 11, 3, 500, 8, True, 0.6, 22,
                                      decay=(0.05, 0.1))),
    "spread": reference_spread,       # after the cases it reads
}

if __name__ == "__main__":
    for t in (sys.argv[1:] or list(TARGETS)):
        TARGETS[t]()
