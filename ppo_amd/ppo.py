"""Host-side mirror of the reference's PPO.py class surface, backed by the gfx950 HIP engine.

Same names, argument order and observable behaviour as the reference module
(/root/reference/PPO.py): module-level ``device``, ``RolloutBuffer``, ``ActorCritic``, ``PPO`` with
``select_action`` / ``update`` / ``set_action_std`` / ``decay_action_std`` / ``save`` / ``load``.
All arithmetic runs in libppo_hip.so through the C ABI of include/ppo_hip.h; PyTorch owns the
device memory and the streams, nothing else.  There is no CPU fallback: without a GPU the objects
can be constructed and (de)serialised, and every compute call raises.

Beyond the reference surface (which handles one env step per call) the same objects accept a
batch of N lockstep envs per call: ``select_action(states[N, S])`` and tensor-valued
``buffer.rewards.append(r[N])`` / ``buffer.is_terminals.append(done[N])``.
"""
import atexit
import ctypes
import os
import time
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import dist as _dist
try:
    from . import _fastcall   # optional: the ctypes-free per-step binding (python -m ppo_amd.build makes it)
except ImportError:
    _fastcall = None

################################## set device ##################################
print("============================================================================================")
# set device to cpu or cuda  (PPO.py:6-16)
device = torch.device('cpu')
if torch.cuda.is_available():
    # one process per GPU under torch.distributed.run: rank-local device; cuda:0 otherwise (PPO.py:11)
    device = torch.device('cuda:' + os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(device)
    torch.cuda.empty_cache()
    print("Device set to : " + str(torch.cuda.get_device_name(device)))
else:
    print("Device set to : cpu")
print("============================================================================================")

HIDDEN = 64


def _close_at_exit(agent_ref):
    agent = agent_ref()
    if agent is not None:
        try:
            agent.close()
        except Exception:
            pass


def _require_gpu():
    if device.type != 'cuda':
        raise RuntimeError("ppo_amd: no GPU visible; the HIP engine has no CPU fallback")


def _param_layout(state_dim, action_dim):
    """(key, shape) in state_dict order of the reference's ActorCritic (PPO.py:49-73)."""
    out = []
    for pre, o in (("actor", action_dim), ("critic", 1)):
        out += [(f"{pre}.0.weight", (HIDDEN, state_dim)), (f"{pre}.0.bias", (HIDDEN,)),
                (f"{pre}.2.weight", (HIDDEN, HIDDEN)), (f"{pre}.2.bias", (HIDDEN,)),
                (f"{pre}.4.weight", (o, HIDDEN)), (f"{pre}.4.bias", (o,))]
    return out


################################## PPO Policy ##################################
class _Column:
    """List-like view of one per-step column of the device-resident rollout buffer.

    Supports what callers of the reference do with ``buffer.<name>``: ``len``, indexing,
    iteration, ``append`` and ``del col[:]`` (RolloutBuffer.clear, PPO.py:29-35).
    """

    def __init__(self, buf, name):
        self._buf = buf
        self._name = name

    def __len__(self):
        return self._buf._count[self._name]

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("list index out of range")
        return self._buf._element(self._name, i)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def append(self, value):
        self._buf._append(self._name, value)

    def __delitem__(self, key):
        if not (isinstance(key, slice) and key.indices(len(self)) == (0, len(self), 1)):
            raise NotImplementedError("only `del column[:]` is supported on a device column")
        self._buf._count[self._name] = 0

    def __repr__(self):
        return f"<device column {self._name}: {len(self)} steps>"


class _HostColumn(list):
    """``buffer.rewards`` / ``buffer.is_terminals``: a plain Python list for scalar appends
    (train.py:180-181); a tensor/array of N values is one lockstep step of N envs and goes straight
    to the device buffer."""

    def __init__(self, buf, name):
        super().__init__()
        self._buf = buf
        self._name = name

    # train.py:180-181 appends one scalar per step: that case is one dictionary lookup on the exact type
    # (torch.Tensor's instance check is a Python-level call and cost more than the append itself).
    # Python scalars go in as they are, numpy scalars as the Python scalar of the same value.
    _FAST = {float: None, bool: None, int: None, np.float32: float, np.float64: float, np.bool_: bool}

    def append(self, value, _fast=_FAST, _append=list.append):
        conv = _fast.get(type(value), _fast)
        if conv is None:
            _append(self, value)
            return
        if conv is not _fast:
            _append(self, conv(value))
            return
        if isinstance(value, (torch.Tensor, np.ndarray)) and getattr(value, "ndim", 0) >= 1 \
                and value.shape[0] > 1:
            self._buf._append_step(self._name, value)
        else:
            if isinstance(value, (torch.Tensor, np.ndarray)):
                value = value.reshape(-1)[0].item()
            super().append(value)

    def __len__(self):
        return super().__len__() + self._buf._count[self._name]

    def __delitem__(self, key):
        super().__delitem__(key)
        if isinstance(key, slice):
            self._buf._count[self._name] = 0


class RolloutBuffer:
    """Device-resident structure-of-arrays rollout buffer with the reference's list facade.

    Storage (grown geometrically, fp32 unless noted): states [cap, S], actions int32 [cap] or
    [cap, A], logprobs [cap], state_values [cap], returns/advantages scratch; in batched mode also
    rewards [cap] and terminals uint8 [cap].  Rows are in append order; a batched step of N envs
    occupies N consecutive rows (time-major [steps, N]).
    """

    def __init__(self, state_dim=None, action_dim=None, continuous=False, dev=None):
        self._S, self._A, self._cont = state_dim, action_dim, continuous
        self._dev = dev if dev is not None else device
        self._cap = 0
        self._before_move = None   # called before the columns' storage is reallocated
        self._count = {"actions": 0, "states": 0, "logprobs": 0, "state_values": 0,
                       "rewards": 0, "is_terminals": 0}
        self._n_envs = 1
        self._store = {}
        self.actions = _Column(self, "actions")
        self.states = _Column(self, "states")
        self.logprobs = _Column(self, "logprobs")
        self.rewards = _HostColumn(self, "rewards")
        self.state_values = _Column(self, "state_values")
        self.is_terminals = _HostColumn(self, "is_terminals")

    # ---- storage
    def _shapes(self):
        S, A = self._S, self._A
        return {"states": ((S,), torch.float32),
                "actions": ((A,), torch.float32) if self._cont else ((), torch.int32),
                "logprobs": ((), torch.float32), "state_values": ((), torch.float32),
                "rewards": ((), torch.float32), "is_terminals": ((), torch.uint8),
                "returns": ((), torch.float32), "advantages": ((), torch.float32)}

    def reserve(self, rows):
        if rows <= self._cap:
            return
        hook = self._before_move() if self._before_move is not None else None   # (a weakref.WeakMethod)
        if hook is not None:
            hook()   # kernels that still write rows of the old storage finish first
        cap = max(rows, 2 * self._cap, 1024)
        for name, (shp, dt) in self._shapes().items():
            new = torch.empty((cap,) + shp, dtype=dt, device=self._dev)
            old = self._store.get(name)
            if old is not None and self._cap:
                new[:self._cap] = old
            self._store[name] = new
        self._cap = cap
        # base address and row size in bytes of every column: the single-step path computes row
        # addresses arithmetically instead of slicing tensors
        self._row_addr = {name: (t.data_ptr(), t.stride(0) * t.element_size() if t.dim() else 0)
                          for name, t in self._store.items()}

    def rows(self, name, lo, hi):
        return self._store[name][lo:hi]

    @property
    def n_envs(self):
        return self._n_envs

    # ---- list facade
    def _element(self, name, i):
        t = self._store[name][i]
        if name == "actions":
            return t.reshape(1, self._A) if self._cont else t.to(torch.int64)
        if name == "logprobs":
            return t.reshape(1) if self._cont else t
        if name == "state_values":
            return t.reshape(1)
        return t

    def _append(self, name, value):
        i = self._count[name]
        self.reserve(i + 1)
        dst = self._store[name]
        v = torch.as_tensor(value, device=self._dev)
        dst[i] = v.reshape(dst[i].shape).to(dst.dtype)
        self._count[name] = i + 1

    def _append_step(self, name, value):
        v = torch.as_tensor(value, device=self._dev).reshape(-1)
        n = v.shape[0]
        i = self._count[name]
        self.reserve(i + n)
        dst = self._store[name]
        dst[i:i + n] = v.to(dst.dtype) if name != "is_terminals" else (v != 0).to(torch.uint8)
        self._count[name] = i + n
        self._n_envs = n

    def claim_step(self, n):
        """Rows [lo, lo+n) for the next lockstep step: the four act() columns advance together."""
        lo = self._count["states"]
        self.reserve(lo + n)
        for k in ("states", "actions", "logprobs", "state_values"):
            self._count[k] = lo + n
        if n > 1:
            self._n_envs = n
        return lo

    def claim_env_slots(self, n):
        """In-place variant for device-side envs: views of the next step's reward/terminal rows
        (the env kernel writes them; counts advance here)."""
        lo = self._count["rewards"]
        self.reserve(lo + n)
        self._count["rewards"] = lo + n
        self._count["is_terminals"] = lo + n
        self._n_envs = n
        return self._store["rewards"][lo:lo + n], self._store["is_terminals"][lo:lo + n]

    def load_rollout(self, states, actions, logprobs, state_values, rewards, is_terminals):
        """Replay: replace the buffer content with one recorded flat rollout (arrays of T steps in
        the reference's append order).  rewards / is_terminals stay host lists of Python scalars,
        exactly what train.py:180-181 would have appended."""
        self.clear()
        T = len(rewards)
        self.reserve(T)
        put = lambda name, arr, dt: self._store[name][:T].copy_(
            torch.as_tensor(np.asarray(arr), dtype=dt).reshape(self._store[name][:T].shape))
        put("states", states, torch.float32)
        put("actions", actions, torch.float32 if self._cont else torch.int32)
        put("logprobs", logprobs, torch.float32)
        put("state_values", state_values, torch.float32)
        for k in ("states", "actions", "logprobs", "state_values"):
            self._count[k] = T
        list.extend(self.rewards, [float(r) for r in rewards])
        list.extend(self.is_terminals, [bool(t) for t in is_terminals])

    def clear(self):
        """PPO.py:29-35."""
        del self.actions[:]
        del self.states[:]
        del self.logprobs[:]
        del self.rewards[:]
        del self.state_values[:]
        del self.is_terminals[:]
        self._n_envs = 1


class ActorCritic(nn.Module):
    """Parameter holder with the reference's module tree (PPO.py:38-73): ``actor`` and ``critic``
    are nn.Sequential stacks whose weights are VIEWS into one flat fp32 device vector laid out in
    state_dict order - the vector the HIP kernels read and the fused Adam writes."""

    def __init__(self, state_dim, action_dim, has_continuous_action_space, action_std_init):
        super(ActorCritic, self).__init__()
        self.has_continuous_action_space = has_continuous_action_space
        self.state_dim = state_dim
        self.action_dim = action_dim
        self._var_host = 0.0
        if has_continuous_action_space:
            self.action_var = torch.full((action_dim,), action_std_init * action_std_init).to(device)
            self._var_host = float(np.float32(action_std_init * action_std_init))
        # same construction order as the reference so default init draws match under a seed
        head = nn.Tanh() if has_continuous_action_space else nn.Softmax(dim=-1)
        self.actor = nn.Sequential(nn.Linear(state_dim, 64), nn.Tanh(), nn.Linear(64, 64),
                                   nn.Tanh(), nn.Linear(64, action_dim), head)
        self.critic = nn.Sequential(nn.Linear(state_dim, 64), nn.Tanh(), nn.Linear(64, 64),
                                    nn.Tanh(), nn.Linear(64, 1))
        self._flatten()

    def _flatten(self):
        n_actor = ctypes.c_int64()
        total = _lib.lib().ppo_param_count(self.state_dim, self.action_dim, ctypes.byref(n_actor))
        self.n_actor = int(n_actor.value)
        flat = torch.empty(total, dtype=torch.float32, device=device)
        off = 0
        sd = dict(self.named_parameters())
        for key, shp in _param_layout(self.state_dim, self.action_dim):
            p = sd[key]
            n = p.numel()
            view = flat[off:off + n].view(shp)
            view.copy_(p.data)
            # a Parameter made FROM the view shares the flat vector's version counter (one assigned
            # through .data would not): an in-place write through any of the 12 tensors moves
            # flat._version, which the single-state session checks before every step
            mod_path, attr = key.rsplit(".", 1)
            setattr(self.get_submodule(mod_path), attr, nn.Parameter(view, requires_grad=True))
            off += n
        assert off == total
        self.flat = flat

    _before_write = None   # called before load_state_dict overwrites the parameters

    def load_state_dict(self, *args, **kwargs):
        hook = self._before_write() if self._before_write is not None else None   # (a weakref.WeakMethod)
        if hook is not None:
            hook()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, recurse=True):
        # The parameters are views into one flat fp32 vector on the engine's device; .to(device)
        # as the reference calls it (PPO.py:138,144) is a no-op, and a conversion that would change
        # dtype or device cannot be honoured (the kernels read that vector): refuse it loudly.
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.flat.device))
        if probe.dtype != torch.float32 or probe.device != self.flat.device:
            raise RuntimeError("ppo_amd.ActorCritic holds fp32 parameters on the engine's device; "
                               f"conversion to {probe.dtype} on {probe.device} is not supported")
        return self

    def set_action_std(self, new_action_std):
        if self.has_continuous_action_space:
            self.action_var = torch.full((self.action_dim,), new_action_std * new_action_std).to(device)
            self._var_host = float(np.float32(new_action_std * new_action_std))
        else:
            print("--------------------------------------------------------------------------------------------")
            print("WARNING : Calling ActorCritic::set_action_std() on discrete action space policy")
            print("--------------------------------------------------------------------------------------------")

    def forward(self):
        raise NotImplementedError

    def _var32(self):
        """fp32 value of std*std (what torch.full stores in action_var, PPO.py:46,77), kept on the
        host so that no act / evaluate / update call has to read the device tensor back."""
        return self._var_host

    def act(self, state, _seed=None, _offset=0, _out=None):
        """Batched ActorCritic.act (PPO.py:86-100): state [S] or [N, S] -> (action, logprob, value)."""
        _require_gpu()
        x = torch.as_tensor(state, dtype=torch.float32, device=device)
        single = x.dim() == 1
        x = x.reshape(-1, self.state_dim).contiguous()
        n = x.shape[0]
        if _out is None:
            act = (torch.empty((n, self.action_dim), dtype=torch.float32, device=device)
                   if self.has_continuous_action_space
                   else torch.empty((n,), dtype=torch.int32, device=device))
            logp = torch.empty((n,), dtype=torch.float32, device=device)
            val = torch.empty((n,), dtype=torch.float32, device=device)
        else:
            act, logp, val = _out
        if _seed is None:
            _seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        _lib.check(_lib.lib().ppo_act(self.flat.data_ptr(), self.state_dim, self.action_dim,
                                      int(self.has_continuous_action_space), self._var32(),
                                      x.data_ptr(), n, _seed, _offset, act.data_ptr(),
                                      logp.data_ptr(), val.data_ptr(), _lib.stream_ptr()))
        if _out is not None:
            return act, logp, val
        if self.has_continuous_action_space:
            return (act.reshape(1, -1), logp.reshape(1), val.reshape(1)) if single else (act, logp, val.reshape(-1, 1))
        if single:
            return act.to(torch.int64).reshape(()), logp.reshape(()), val.reshape(1)
        return act.to(torch.int64), logp, val.reshape(-1, 1)

    def evaluate(self, state, action):
        """ActorCritic.evaluate (PPO.py:102-121) -> (logprobs [T], values [T,1], entropy [T]).
        With autograd on and parameters that require grad (the default, as in the reference) the three
        outputs carry history w.r.t. the 12 parameters: a caller's own loss.backward() (PPO.py:244) fills
        their .grad through ppo_evaluate_backward.  Under torch.no_grad() they are plain tensors.
        PPO.update() does not come through here: its backward lives in the fused update kernel."""
        _require_gpu()
        x = torch.as_tensor(state, dtype=torch.float32, device=device).reshape(-1, self.state_dim).contiguous()
        n = x.shape[0]
        if self.has_continuous_action_space:
            a = torch.as_tensor(action, dtype=torch.float32, device=device).reshape(n, self.action_dim).contiguous()
        else:
            a = torch.as_tensor(action, device=device).reshape(n).to(torch.int32).contiguous()
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _EvaluateFn.apply(self, x.detach(), a.detach(), *params)
        return self._evaluate_forward(x, a)

    def _evaluate_forward(self, x, a):
        n = x.shape[0]
        logp = torch.empty((n,), dtype=torch.float32, device=device)
        val = torch.empty((n,), dtype=torch.float32, device=device)
        ent = torch.empty((n,), dtype=torch.float32, device=device)
        _lib.check(_lib.lib().ppo_evaluate(self.flat.data_ptr(), self.state_dim, self.action_dim,
                                           int(self.has_continuous_action_space), self._var32(),
                                           x.data_ptr(), a.data_ptr(), n, logp.data_ptr(),
                                           val.data_ptr(), ent.data_ptr(), _lib.stream_ptr()))
        return logp, val.reshape(-1, 1), ent

    def _evaluate_backward(self, x, a, g_logp, g_val, g_ent, var32):
        """dL/dparams (flat [P]) from dL/dlogprob, dL/dvalue, dL/dentropy (each [T] or None)."""
        L = _lib.lib()
        n = x.shape[0]
        P = self.flat.numel()
        need = L.ppo_workspace_bytes(self.state_dim, self.action_dim, n)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        grad = torch.empty(P + 2, dtype=torch.float32, device=device)
        prep = lambda g: None if g is None else g.to(torch.float32).reshape(-1).contiguous()
        gl, gv, ge = prep(g_logp), prep(g_val), prep(g_ent)
        _lib.check(L.ppo_evaluate_backward(self.flat.data_ptr(), self.state_dim, self.action_dim,
                                           int(self.has_continuous_action_space), var32, x.data_ptr(),
                                           a.data_ptr(), n, _lib.ptr(gl), _lib.ptr(gv), _lib.ptr(ge),
                                           grad.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return grad[:P]


class _EvaluateFn(torch.autograd.Function):
    """evaluate() as an autograd node: forward = ppo_evaluate, backward = ppo_evaluate_backward (the
    fused forward/backward kernel seeded with the incoming gradients).  Gradients flow to the 12
    parameters only (the reference never differentiates w.r.t. states or actions either)."""

    @staticmethod
    def forward(ctx, module, x, a, *params):
        ctx.module = module
        ctx.version = module.flat._version
        ctx.var32 = module._var32()
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(x, a)
        return module._evaluate_forward(x, a)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_logp, g_val, g_ent):
        module = ctx.module
        if module.flat._version != ctx.version:
            raise RuntimeError("ppo_amd: the parameters were modified in place between evaluate() and backward(); "
                               "the backward pass recomputes the forward from the parameters")
        x, a = ctx.saved_tensors
        flat_grad = module._evaluate_backward(x, a, g_logp, g_val, g_ent, ctx.var32)
        grads, off = [], 0
        for shp in ctx.shapes:
            k = int(np.prod(shp))
            grads.append(flat_grad[off:off + k].view(shp))
            off += k
        return (None, None, None, *grads)


class FusedAdam:
    """Adam state for the flat parameter vector with the two learning-rate groups of PPO.py:139-142.
    Keeps torch.optim's visible shape (``param_groups``, ``state_dict``, ``zero_grad``, ``step``)."""

    def __init__(self, policy, lr_actor, lr_critic, betas=(0.9, 0.999), eps=1e-8):
        self.policy = policy
        self.param_groups = [{'params': list(policy.actor.parameters()), 'lr': lr_actor,
                              'betas': betas, 'eps': eps},
                             {'params': list(policy.critic.parameters()), 'lr': lr_critic,
                              'betas': betas, 'eps': eps}]
        n = policy.flat.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n + 2, dtype=torch.float32, device=device)  # [P]=surrogate, [P+1]=value loss
        self.step_count = 0

    def zero_grad(self):
        pass  # the reduction kernel overwrites the gradient every epoch

    def step(self, loss_out=None):
        _require_gpu()
        self.step_count += 1
        g0, g1 = self.param_groups
        _lib.check(_lib.lib().ppo_adam_step(
            self.policy.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), self.policy.flat.numel(), self.policy.n_actor,
            self.step_count, g0['lr'], g1['lr'], g0['betas'][0], g0['betas'][1], g0['eps'],
            _lib.ptr(loss_out), _lib.stream_ptr()))

    def reduce_and_step(self, state_dim, action_dim, ws, loss_out=None):
        """grad <- fixed-order sum of the per-workgroup partials in `ws`, then step(): one launch."""
        _require_gpu()
        self.step_count += 1
        g0, g1 = self.param_groups
        _lib.check(_lib.lib().ppo_reduce_adam_step(
            state_dim, action_dim, ws.data_ptr(), ws.numel(), self.grad.data_ptr(),
            self.policy.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.step_count, g0['lr'], g1['lr'], g0['betas'][0], g0['betas'][1], g0['eps'],
            _lib.ptr(loss_out), _lib.stream_ptr()))

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg.cpu(), 'exp_avg_sq': self.exp_avg_sq.cpu(),
                'lr': [g['lr'] for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, lr in zip(self.param_groups, sd['lr']):
            g['lr'] = lr


class PPO:
    def __init__(self, state_dim, action_dim, lr_actor, lr_critic, gamma, K_epochs, eps_clip,
                 has_continuous_action_space, action_std_init=0.6):

        self.has_continuous_action_space = has_continuous_action_space

        if has_continuous_action_space:
            self.action_std = action_std_init

        self.gamma = gamma
        self.eps_clip = eps_clip
        self.K_epochs = K_epochs
        self.state_dim = state_dim
        self.action_dim = action_dim

        self.buffer = RolloutBuffer(state_dim, action_dim, has_continuous_action_space)

        self.policy = ActorCritic(state_dim, action_dim, has_continuous_action_space, action_std_init).to(device)
        self.optimizer = FusedAdam(self.policy, lr_actor, lr_critic)

        self.policy_old = ActorCritic(state_dim, action_dim, has_continuous_action_space, action_std_init).to(device)
        self.policy_old.load_state_dict(self.policy.state_dict())

        self.MseLoss = nn.MSELoss()

        # engine state
        self._rng_seed = None      # drawn from torch's generator on first use (honours torch.manual_seed)
        self._act_session = None   # single-env protocol: session with a long-lived act kernel, made on first use
        self._host_step = None     # ... and its per-step closure
        self._pin = None           # ... or, with PPO_ACT_SESSION=0, the page-locked staging of ppo_act_host
        # weak: the hooks must not keep this object alive (a cycle would leave the session to the cycle
        # collector, i.e. possibly to interpreter shutdown, after the HIP runtime is gone)
        self.buffer._before_move = weakref.WeakMethod(self._act_drain)
        self.policy_old._before_write = weakref.WeakMethod(self._act_drain)
        self._rng_offset = 0
        self._ws = None
        self._moments = torch.zeros(3, dtype=torch.float64, device=device)
        self.kernel_events = None  # set to a list to collect (start, end) CUDA events around every fwd/bwd launch
        self.phase_events = None   # set to a list to collect (name, start, end) events around scan / normalise
        self.comm_events = None    # set to a list to collect (start, end) events around every gradient all-reduce
        self.last_dp_enqueue_s = None  # data-parallel path: host seconds spent enqueueing the last K-epoch loop
        self.last_loss = None      # device tensor [K_epochs]: loss.mean() of every epoch of the last update
        self.debug_outputs = None  # optional dict of per-sample first-epoch outputs (set by update(debug=True))

    # ------------------------------------------------------------------ std schedule (PPO.py:149-173)
    def set_action_std(self, new_action_std):
        if self.has_continuous_action_space:
            self.action_std = new_action_std
            self.policy.set_action_std(new_action_std)
            self.policy_old.set_action_std(new_action_std)
        else:
            print("--------------------------------------------------------------------------------------------")
            print("WARNING : Calling PPO::set_action_std() on discrete action space policy")
            print("--------------------------------------------------------------------------------------------")

    def decay_action_std(self, action_std_decay_rate, min_action_std):
        print("--------------------------------------------------------------------------------------------")
        if self.has_continuous_action_space:
            self.action_std = self.action_std - action_std_decay_rate
            self.action_std = round(self.action_std, 4)
            if (self.action_std <= min_action_std):
                self.action_std = min_action_std
                print("setting actor output action_std to min_action_std : ", self.action_std)
            else:
                print("setting actor output action_std to : ", self.action_std)
            self.set_action_std(self.action_std)

        else:
            print("WARNING : Calling PPO::decay_action_std() on discrete action space policy")
        print("--------------------------------------------------------------------------------------------")

    # ------------------------------------------------------------------ rollout (PPO.py:175-198)
    def _next_rng(self):
        if self._rng_seed is None:
            # drawn from torch's generator (honours torch.manual_seed); data-parallel ranks seed torch
            # identically to get identical initial weights, so the rank is folded into the Philox key:
            # every rank's envs get their own noise streams
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            self._rng_seed = (seed ^ (_dist.rank() * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        self._rng_offset += 1
        return self._rng_seed, self._rng_offset

    def select_action(self, state):
        """One env: ``state`` of shape [S] -> int (discrete) or np.ndarray[A] float32 (continuous),
        exactly as the reference.  N lockstep envs: ``state`` of shape [N, S] (ndarray or tensor)
        -> device tensor of N actions (int64 [N] or float32 [N, A]); nothing is copied to the host."""
        if self._host_step is not None and type(state) is np.ndarray and state.ndim == 1:
            return self._host_step(state)      # the single-env protocol's steady state
        _require_gpu()
        if not isinstance(state, torch.Tensor) and np.ndim(state) == 1:
            return self._select_action_host(state)
        x = torch.as_tensor(state, dtype=torch.float32, device=device) \
            if not (isinstance(state, torch.Tensor) and state.dtype == torch.float32 and state.device == device) \
            else state
        batched = x.dim() == 2
        x = x.reshape(-1, self.state_dim)
        n = x.shape[0]
        b = self.buffer
        lo = b.claim_step(n)
        b.rows("states", lo, lo + n).copy_(x)
        self._act_rows(lo, n)
        act = b.rows("actions", lo, lo + n)
        if batched:
            return act if self.has_continuous_action_space else act.to(torch.int64)
        if self.has_continuous_action_space:
            return act.detach().cpu().numpy().flatten()
        return int(act.item())

    def _act_drain(self):
        """Withdraw the act kernel that waits for the next single state, and wait for the session's
        stream: required before policy_old's weights change or the buffer's storage moves."""
        if self._act_session is not None:
            _lib.check(self._act_drain_fn(self._act_session))

    def close(self):
        """Ends the single-state session (withdraws the kernel that waits for the next state, waits for
        the session's stream, frees its mailbox and stream).  Safe to call more than once; the next
        single-state select_action opens a new session.  Called at interpreter exit for every live
        agent, before the HIP runtime goes away."""
        sess, self._act_session = getattr(self, "_act_session", None), None
        self._host_step = None
        self._pin = None
        if sess is not None:
            _lib.check(_lib.lib().ppo_act_session_destroy(sess))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _select_action_host(self, state):
        """The reference's protocol, one host state per call (PPO.py:175-198).  Default: a session
        whose long-lived act kernel waits for each state in a page-locked mailbox
        (ppo_act_session_step) - no launch and no stream wait between "state known" and "action
        known".  PPO_ACT_SESSION=0: one launch per call with the state in its arguments, and a spin
        on a ticket word the kernel writes next to the action (ppo_act_host)."""
        if type(state) is not np.ndarray:
            state = np.asarray(state)
        if state.shape != (self.state_dim,):
            raise ValueError(f"select_action: state of shape {state.shape}, expected {(self.state_dim,)}")
        if self._host_step is not None:
            return self._host_step(state)
        if self._pin is None:
            L = _lib.lib()
            # [state (64 words)][action words (8)][ticket word]; page-locked for ppo_act_host's sake
            self._pin = torch.zeros(64 + 16, dtype=torch.float32).pin_memory()
            a = self._pin.numpy()
            self._pin_np = (a[:self.state_dim], a[64:64 + self.action_dim], a[64:65].view(np.int32))
            p = self._pin.data_ptr()
            self._pin_ptr = (p, p + 256, p + 256 + 32)
            self._ticket = 0
            self._act_host = L.ppo_act_host
            if os.environ.get("PPO_ACT_SESSION", "1") != "0":
                h = ctypes.c_void_p()
                _lib.check(L.ppo_act_session_create(ctypes.byref(h), self.state_dim, self.action_dim,
                                                    int(self.has_continuous_action_space),
                                                    int(os.environ.get("PPO_ACT_WAIT_US", "1000"))))
                self._act_session = h
                atexit.register(_close_at_exit, weakref.ref(self))
                self._act_step_fn = L.ppo_act_session_step
                self._act_drain_fn = L.ppo_act_session_drain
        if self._act_session is not None:
            if self._rng_seed is None:
                self._next_rng()
                self._rng_offset -= 1
            self._host_step = self._make_session_step()
            return self._host_step(state)
        self._pin_np[0][:] = state
        b = self.buffer
        lo = b.claim_step(1)
        seed, off = self._next_rng()
        pol = self.policy_old
        cont = self.has_continuous_action_space
        ra = b._row_addr
        (ps, ss), (pa, sa), (pl, sl), (pv, sv) = ra["states"], ra["actions"], ra["logprobs"], ra["state_values"]
        self._ticket = ticket = (self._ticket % 0x7fffffff) + 1
        rc = self._act_host(
            pol.flat.data_ptr(), self.state_dim, self.action_dim, int(cont), pol._var_host,
            self._pin_ptr[0], ps + lo * ss, seed, off, pa + lo * sa, pl + lo * sl, pv + lo * sv,
            self._pin_ptr[1], self._pin_ptr[2], ticket, torch.cuda.current_stream().cuda_stream)
        if rc:
            _lib.check(rc)
        if cont:
            return self._pin_np[1].copy()
        return int(self._pin_np[2][0])

    def _make_session_step(self):
        """The per-step work of the session path as one closure over everything that does not change
        between steps (this function is most of what a step costs on the host).  Whatever stays the
        same from step to step - weights and their version, variance, the page-locked buffers, the
        buffer columns, the stream - is bound in the library (ppo_act_session_bind) and bound again
        when it changes: the per-step call then carries three arguments instead of thirteen."""
        b, cnt, pol = self.buffer, self.buffer._count, self.policy_old
        state_np, act_np, act_int = self._pin_np
        state_ptr, act_ptr = self._pin_ptr[0], self._pin_ptr[1]
        L = _lib.lib()
        bind, step, sess, seed = L.ppo_act_session_bind, L.ppo_act_session_step_bound, self._act_session, self._rng_seed
        flat = pol.flat
        params = flat.data_ptr()               # the flat vector is allocated once
        want_shape = (self.state_dim,)
        cont = self.has_continuous_action_space
        raw_stream = torch._C._cuda_getCurrentRawStream
        dev_index = device.index or 0
        bound = [None, None, None, None]       # version, variance, the columns' address table, stream
        # the per-step call without ctypes where the extension is built (csrc/fastcall.c): it takes the
        # observation through the buffer protocol, converts it into the page-locked state buffer, calls
        # ppo_act_session_step_bound and hands back the discrete action
        fast = None
        if _fastcall is not None and os.environ.get("PPO_FASTCALL", "1") != "0":
            fast = _fastcall.Stepper(ctypes.cast(step, ctypes.c_void_p).value, sess.value, state_ptr, act_ptr,
                                     self.state_dim, cont).step

        def rebind(version, var, ra, stream):
            for col, width in (("states", self.state_dim * 4), ("actions", (self.action_dim if cont else 1) * 4),
                               ("logprobs", 4), ("state_values", 4)):
                assert ra[col][1] == width, "the session addresses the buffer's columns as dense rows"
            _lib.check(bind(sess, params, version, var, state_ptr, act_ptr, seed, ra["states"][0], ra["actions"][0],
                            ra["logprobs"][0], ra["state_values"][0], stream))
            bound[:] = [version, var, ra, stream]

        def host_step(state):
            lo = cnt["states"]
            if lo >= b._cap:
                b.reserve(lo + 1)
            # flat._version counts the in-place writes to the weights (through any of their views): a
            # write since the waiting kernel was launched makes the library replace that kernel
            version, var, ra = flat._version, pol._var_host, b._row_addr
            if version != bound[0] or var != bound[1] or ra is not bound[2]:
                rebind(version, var, ra, raw_stream(dev_index))
            off = self._rng_offset + 1
            if fast is not None:
                try:
                    act = fast(state, lo, off)
                except RuntimeError as e:
                    _lib.check(e.args[0])
                    raise
                if act is not NotImplemented:
                    cnt["states"] = cnt["actions"] = cnt["logprobs"] = cnt["state_values"] = lo + 1
                    self._rng_offset = off
                    return act_np.copy() if cont else act
            if state.shape != want_shape:
                raise ValueError(f"select_action: state of shape {state.shape}, expected {want_shape}")
            state_np[:] = state
            rc = step(sess, lo, off)
            if rc:
                _lib.check(rc)
            cnt["states"] = cnt["actions"] = cnt["logprobs"] = cnt["state_values"] = lo + 1
            self._rng_offset = off
            if cont:
                return act_np.copy()
            return int(act_int[0])

        return host_step

    def act_inplace(self, n_envs):
        """Device-side env variant: the env has already written this step's states into
        ``buffer.rows('states', lo, lo+n)`` obtained from ``next_state_slot``; acts on them in place
        and returns the device view of the N actions."""
        lo = self._pending_lo
        self._act_rows(lo, n_envs)
        return self.buffer.rows("actions", lo, lo + n_envs)

    def next_state_slot(self, n_envs):
        lo = self.buffer.claim_step(n_envs)
        self._pending_lo = lo
        return self.buffer.rows("states", lo, lo + n_envs)

    def synthetic_rollout(self, n_envs, n_steps, reward_kind=0, p_terminal=1.0 / 200.0, step0=0,
                          env_seed=1234):
        """Measurement harness: n_steps lockstep steps of n_envs synthetic envs (ppo_synth_env_step)
        each followed by the batched act, all launches enqueued by one C call.  Equivalent to calling
        next_state_slot / claim_env_slots / ppo_synth_env_step / act_inplace n_steps times."""
        _require_gpu()
        b = self.buffer
        n = n_envs * n_steps
        lo = b.claim_step(n)
        b._n_envs = n_envs
        b._count["rewards"] += n
        b._count["is_terminals"] += n
        seed, off = self._next_rng()
        self._rng_offset += n_steps - 1
        pol = self.policy_old
        _lib.check(_lib.lib().ppo_synth_rollout(
            pol.flat.data_ptr(), self.state_dim, self.action_dim, int(self.has_continuous_action_space),
            pol._var32(), n_envs, n_steps, reward_kind, p_terminal, step0, n_steps, env_seed, seed, off,
            b.rows("states", lo, lo + n).data_ptr(), b.rows("rewards", lo, lo + n).data_ptr(),
            b.rows("is_terminals", lo, lo + n).data_ptr(), b.rows("actions", lo, lo + n).data_ptr(),
            b.rows("logprobs", lo, lo + n).data_ptr(), b.rows("state_values", lo, lo + n).data_ptr(),
            _lib.stream_ptr()))

    def cartpole_rollout(self, env, n_steps):
        """n_steps lockstep steps of this policy on a ppo_amd.envs.CartPoleVecEnv, all launches
        enqueued by one C call (ppo_cartpole_rollout); equivalent to n_steps rounds of
        next_state_slot / act_inplace / claim_env_slots / env.step."""
        _require_gpu()
        if self.state_dim != 4 or self.action_dim != 2 or self.has_continuous_action_space:
            raise ValueError("cartpole_rollout needs a discrete policy with state_dim 4 and action_dim 2 "
                             f"(got {self.state_dim}, {self.action_dim}, continuous="
                             f"{self.has_continuous_action_space})")
        if not all(hasattr(env, a) for a in ("_state", "_steps", "_episodes", "_obs", "max_episode_steps")):
            raise ValueError("cartpole_rollout needs a ppo_amd.envs.CartPoleVecEnv")
        b = self.buffer
        N = env.n_envs
        n = N * n_steps
        lo = b.claim_step(n)
        b._n_envs = N
        b._count["rewards"] += n
        b._count["is_terminals"] += n
        seed, off = self._next_rng()
        self._rng_offset += n_steps - 1
        _lib.check(_lib.lib().ppo_cartpole_rollout(
            self.policy_old.flat.data_ptr(), env._state.data_ptr(), env._steps.data_ptr(),
            env._episodes.data_ptr(), env.finished_return_sum.data_ptr(), env.finished_count.data_ptr(),
            env._obs.data_ptr(), N, n_steps, env.max_episode_steps, env._seed, seed, off,
            b.rows("states", lo, lo + n).data_ptr(), b.rows("rewards", lo, lo + n).data_ptr(),
            b.rows("is_terminals", lo, lo + n).data_ptr(), b.rows("actions", lo, lo + n).data_ptr(),
            b.rows("logprobs", lo, lo + n).data_ptr(), b.rows("state_values", lo, lo + n).data_ptr(),
            _lib.stream_ptr()))

    def _act_rows(self, lo, n):
        b = self.buffer
        seed, off = self._next_rng()
        self.policy_old.act(b.rows("states", lo, lo + n), _seed=seed, _offset=off,
                            _out=(b.rows("actions", lo, lo + n), b.rows("logprobs", lo, lo + n),
                                  b.rows("state_values", lo, lo + n)))

    # ------------------------------------------------------------------ update (PPO.py:200-251)
    def _workspace(self, n):
        need = _lib.lib().ppo_workspace_bytes(self.state_dim, self.action_dim, n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def update(self, debug=False):
        _require_gpu()
        self._act_drain()
        L = _lib.lib()
        b = self.buffer
        st = _lib.stream_ptr()
        T = b._count["states"]
        host_rewards = list.__len__(b.rewards)
        dev_rewards = b._count["rewards"]
        if T == 0 or (host_rewards + dev_rewards) != T or \
                (list.__len__(b.is_terminals) + b._count["is_terminals"]) != T:
            raise RuntimeError(f"rollout buffer is inconsistent: {T} acted steps, "
                               f"{host_rewards + dev_rewards} rewards, "
                               f"{list.__len__(b.is_terminals) + b._count['is_terminals']} terminal flags")
        ws = self._workspace(T)
        returns = b.rows("returns", 0, T)
        adv = b.rows("advantages", 0, T)

        def stamp(name=None, since=None):
            if self.phase_events is None:
                return None
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if name is not None:
                self.phase_events.append((name, since, ev))
            return ev

        # Monte Carlo estimate of returns (PPO.py:202-211)
        ev_scan = stamp()
        if host_rewards:
            if dev_rewards:
                raise RuntimeError("mixing scalar and batched reward appends in one rollout is not supported")
            # Python floats are doubles: keep them as fp64 so the scan sees what the reference sees
            r64 = torch.tensor(list(b.rewards), dtype=torch.float64).to(device, non_blocking=True)
            term = torch.tensor([bool(x) for x in list(b.is_terminals)], dtype=torch.uint8).to(device, non_blocking=True)
            _lib.check(L.ppo_returns_scan_flat(r64.data_ptr(), 1, term.data_ptr(), T, float(self.gamma),
                                               returns.data_ptr(), self._moments.data_ptr(),
                                               ws.data_ptr(), ws.numel(), st))
        else:
            n_envs = b.n_envs
            if T % n_envs:
                raise RuntimeError(f"batched rollout of {T} samples is not a multiple of {n_envs} envs")
            _lib.check(L.ppo_returns_scan_rows(b.rows("rewards", 0, T).data_ptr(),
                                               b.rows("is_terminals", 0, T).data_ptr(), T // n_envs, n_envs,
                                               float(self.gamma), returns.data_ptr(),
                                               self._moments.data_ptr(), ws.data_ptr(), ws.numel(), st))
        stamp("scan", ev_scan)
        # Normalizing the rewards (PPO.py:212) over ALL data-parallel shards, advantages (PPO.py:221)
        moments = _dist.merge_moments(self._moments)
        T_global = T if not _dist.is_distributed() else int(round(float(moments[0].item())))
        ev_norm = stamp()
        _lib.check(L.ppo_normalize_advantage(returns.data_ptr(), b.rows("state_values", 0, T).data_ptr(), T,
                                             moments.data_ptr(), returns.data_ptr(), adv.data_ptr(), st))
        stamp("normalize", ev_norm)

        # Optimize policy for K epochs (PPO.py:224-245)
        pol, opt = self.policy, self.optimizer
        P = pol.flat.numel()
        losses = torch.empty(self.K_epochs, dtype=torch.float32, device=device)
        dbg = None
        if debug:
            dbg = {k: torch.empty(T, dtype=torch.float32, device=device)
                   for k in ("logprobs", "state_values", "entropy", "surrogate")}
        var32 = pol._var32()
        distributed = _dist.is_distributed()
        per_epoch = distributed or debug or self.kernel_events is not None
        if not per_epoch:
            # single process: the whole epoch loop is one library call
            g0, g1 = opt.param_groups
            _lib.check(L.ppo_update_epochs(
                pol.flat.data_ptr(), self.state_dim, self.action_dim, int(self.has_continuous_action_space),
                var32, b.rows("states", 0, T).data_ptr(), b.rows("actions", 0, T).data_ptr(),
                b.rows("logprobs", 0, T).data_ptr(), adv.data_ptr(), returns.data_ptr(), T,
                float(self.eps_clip), self.K_epochs, opt.grad.data_ptr(), opt.exp_avg.data_ptr(),
                opt.exp_avg_sq.data_ptr(), opt.step_count + 1, g0['lr'], g1['lr'], g0['betas'][0],
                g0['betas'][1], g0['eps'], losses.data_ptr(), ws.data_ptr(), ws.numel(), st))
            opt.step_count += self.K_epochs
        t_loop = time.perf_counter()
        for e in range(self.K_epochs if per_epoch else 0):
            d = dbg if (dbg is not None and e == 0) else None
            if self.kernel_events is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            _lib.check(L.ppo_epoch_fwdbwd(
                pol.flat.data_ptr(), self.state_dim, self.action_dim, int(self.has_continuous_action_space),
                var32, b.rows("states", 0, T).data_ptr(), b.rows("actions", 0, T).data_ptr(),
                b.rows("logprobs", 0, T).data_ptr(), adv.data_ptr(), returns.data_ptr(), T, T_global,
                float(self.eps_clip),
                _lib.ptr(d["logprobs"]) if d else None, _lib.ptr(d["state_values"]) if d else None,
                _lib.ptr(d["entropy"]) if d else None, _lib.ptr(d["surrogate"]) if d else None,
                ws.data_ptr(), ws.numel(), st))
            if self.kernel_events is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                self.kernel_events.append((ev0, ev1))
            if not distributed and not debug:
                opt.reduce_and_step(self.state_dim, self.action_dim, ws, loss_out=losses[e:e + 1])
                continue
            _lib.check(L.ppo_grad_reduce(self.state_dim, self.action_dim, ws.data_ptr(), ws.numel(),
                                         opt.grad.data_ptr(), st))
            if self.comm_events is not None:
                cev0 = torch.cuda.Event(enable_timing=True)
                cev0.record()
            _dist.all_reduce_sum(opt.grad)
            if self.comm_events is not None:
                cev1 = torch.cuda.Event(enable_timing=True)
                cev1.record()
                self.comm_events.append((cev0, cev1))
            if debug and e == 0:
                dbg["grad"] = opt.grad[:P].clone()
                dbg["returns_norm"] = returns.clone()
                dbg["advantages"] = adv.clone()
                dbg["moments"] = moments.clone()
            opt.zero_grad()
            opt.step(loss_out=losses[e:e + 1])
        self.last_dp_enqueue_s = (time.perf_counter() - t_loop) if distributed else None
        self.last_loss = losses
        self.debug_outputs = dbg

        # Copy new weights into old policy (PPO.py:248)
        self.policy_old.flat.copy_(self.policy.flat)

        # clear buffer (PPO.py:251)
        self.buffer.clear()

    # ------------------------------------------------------------------ checkpoints (PPO.py:253-258)
    def save(self, checkpoint_path):
        torch.save({k: v.detach().cpu().clone() for k, v in self.policy_old.state_dict().items()},
                   checkpoint_path)

    def load(self, checkpoint_path):
        self.policy_old.load_state_dict(torch.load(checkpoint_path, map_location=lambda storage, loc: storage))
        self.policy.load_state_dict(torch.load(checkpoint_path, map_location=lambda storage, loc: storage))

    # ------------------------------------------------------------------ resume (not in the reference)
    # The reference's checkpoint holds policy_old's weights only (PPO.py:253-258): no optimizer state, no
    # action_std, no sampling state, so a reloaded run is not the run that was saved.  save_resume /
    # load_resume write and read everything update() and select_action() depend on, in a file of their
    # own format (plain tensors and numbers; the reference-format file stays what save() writes).
    def save_resume(self, checkpoint_path):
        self._act_drain()
        torch.save({"format": "ppo_amd.resume.v1",
                    "policy": {k: v.detach().cpu().clone() for k, v in self.policy.state_dict().items()},
                    "policy_old": {k: v.detach().cpu().clone() for k, v in self.policy_old.state_dict().items()},
                    "optimizer": self.optimizer.state_dict(),
                    "action_std": self.action_std if self.has_continuous_action_space else None,
                    "rng_seed": self._rng_seed, "rng_offset": self._rng_offset}, checkpoint_path)

    def load_resume(self, checkpoint_path):
        ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if ck.get("format") != "ppo_amd.resume.v1":
            raise ValueError(f"{checkpoint_path} is not a resume checkpoint written by save_resume()")
        self.policy.load_state_dict(ck["policy"])
        self.policy_old.load_state_dict(ck["policy_old"])
        self.optimizer.load_state_dict(ck["optimizer"])
        if self.has_continuous_action_space and ck["action_std"] is not None:
            self.action_std = ck["action_std"]
            self.policy.set_action_std(self.action_std)
            self.policy_old.set_action_std(self.action_std)
        self._rng_seed, self._rng_offset = ck["rng_seed"], ck["rng_offset"]
        self._host_step = None     # a single-state session re-binds the seed on its next step
