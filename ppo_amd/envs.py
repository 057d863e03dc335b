"""Vectorised environments that live on the device, behind the call sequence the reference's
training loop uses on a gym env (env.reset() -> obs, env.step(action) -> (obs, reward, done, info),
env.seed(), env.observation_space.shape, env.action_space.n: train.py:136-143,170,177).

Every env object here steps N independent copies in lockstep: observations are [N, S] float32
device tensors, rewards [N] float32, dones [N] uint8; finished copies are reset inside step(), so
the observation returned is always the one the next select_action has to see.  With n_envs=1 and
`numpy_api=True` the same objects speak the reference's single-env protocol (np.ndarray state,
Python float reward, Python bool done).
"""
import numpy as np
import torch

from . import _lib


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape = shape
        self.n = n


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("ppo_amd.envs needs a GPU: the environments are HIP kernels, there is no CPU path")


class _VecEnv:
    def __init__(self, n_envs, state_dim, numpy_api):
        _require_gpu()
        self.n_envs = int(n_envs)
        self.numpy_api = bool(numpy_api)
        if self.numpy_api and self.n_envs != 1:
            raise ValueError("numpy_api=True is the reference's single-env protocol: n_envs must be 1")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.observation_space = _Space(shape=(state_dim,))
        self._seed = 0
        # one allocation: [obs (N*S floats) | reward (N floats) | done (N bytes)], so that the
        # single-env protocol fetches a whole step result with one device-to-host copy
        n, S = self.n_envs, state_dim
        self._packed = torch.zeros(n * S + n + (n + 3) // 4, dtype=torch.float32, device=self.device)
        self._obs = self._packed[:n * S].view(n, S)
        self._rew = self._packed[n * S:n * S + n]
        self._done = self._packed[n * S + n:].view(torch.uint8)[:n]
        if self.numpy_api:
            self._pin_out = torch.zeros_like(self._packed, device="cpu").pin_memory()
            self._pin_out_np = self._pin_out.numpy()
            self._pin_done_np = self._pin_out[S + 1:].view(torch.uint8).numpy()
            self._pin_act = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._pin_act_np = self._pin_act.numpy()
            self._act_dev = torch.zeros(1, dtype=torch.int32, device=self.device)

    def seed(self, s):
        self._seed = int(s)
        return [self._seed]

    def close(self):
        pass

    def _fetch(self):
        """Single-env protocol: the packed step result in page-locked host memory, one copy + wait."""
        self._pin_out.copy_(self._packed, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        S = self.observation_space.shape[0]
        return self._pin_out_np[:S].copy(), float(self._pin_out_np[S]), bool(self._pin_done_np[0])

    def _out(self):
        if self.numpy_api:
            o, r, d = self._fetch()
            return o, r, d, {}
        return self._obs, self._rew, self._done, {}


class CartPoleVecEnv(_VecEnv):
    """CartPole-v1 physics on the device (ppo_cartpole_reset / ppo_cartpole_step)."""
    max_episode_steps = 500

    def __init__(self, n_envs=1, numpy_api=False):
        super().__init__(n_envs, 4, numpy_api)
        self.action_space = _Space(n=2)
        n, d = self.n_envs, self.device
        self._state = torch.zeros(n, 4, dtype=torch.float64, device=d)
        self._steps = torch.zeros(n, dtype=torch.int32, device=d)
        self._episodes = torch.zeros(n, dtype=torch.int32, device=d)
        self.finished_return_sum = torch.zeros(n, dtype=torch.float32, device=d)
        self.finished_count = torch.zeros(n, dtype=torch.int32, device=d)
        if self.numpy_api:
            # [obs 0..3 | reward | terminal flag | ticket word | pad]: what ppo_cartpole_step_host writes
            self._pin_step = torch.zeros(8, dtype=torch.float32).pin_memory()
            self._pin_step_np = self._pin_step.numpy()
            self._ticket = 0
            self._step_host = _lib.lib().ppo_cartpole_step_host
            p = self._pin_step.data_ptr()
            self._step_host_args = (self._state.data_ptr(), self._steps.data_ptr(), self._episodes.data_ptr(),
                                    self.finished_return_sum.data_ptr(), self.finished_count.data_ptr(), p, p + 24)

    def reset(self):
        _lib.check(_lib.lib().ppo_cartpole_reset(
            self._state.data_ptr(), self._steps.data_ptr(), self._episodes.data_ptr(),
            self.finished_return_sum.data_ptr(), self.finished_count.data_ptr(), self._obs.data_ptr(),
            self.n_envs, self._seed, _lib.stream_ptr()))
        return self._fetch()[0] if self.numpy_api else self._obs

    def step(self, action, obs_out=None, reward_out=None, done_out=None):
        """action: int (numpy_api) or an [N] integer device tensor.  The *_out tensors let a caller
        have the results written straight into its rollout buffer rows."""
        if self.numpy_api:
            # the reference's per-step protocol: one launch with the action in its arguments; the kernel
            # writes observation, reward and flag into page-locked memory and then a ticket word
            self._ticket = ticket = (self._ticket % 0x7fffffff) + 1
            rc = self._step_host(*self._step_host_args[:5], int(action), *self._step_host_args[5:7], ticket,
                                 self.max_episode_steps, self._seed, torch.cuda.current_stream().cuda_stream)
            if rc:
                _lib.check(rc)
            rec = self._pin_step_np
            return rec[:4].copy(), float(rec[4]), bool(rec[5] != 0.0), {}
        if action.dtype != torch.int32:
            action = action.to(torch.int32)
        obs = self._obs if obs_out is None else obs_out
        rew = self._rew if reward_out is None else reward_out
        done = self._done if done_out is None else done_out
        _lib.check(_lib.lib().ppo_cartpole_step(
            self._state.data_ptr(), self._steps.data_ptr(), self._episodes.data_ptr(),
            self.finished_return_sum.data_ptr(), self.finished_count.data_ptr(), action.data_ptr(),
            obs.data_ptr(), rew.data_ptr(), done.data_ptr(), self.n_envs, self.max_episode_steps,
            self._seed, _lib.stream_ptr()))
        if obs_out is not None:
            return obs, rew, done, {}
        return self._out()

    def pop_episode_stats(self):
        """(sum of returns, number) of the episodes finished since the last call; one host sync."""
        s = float(self.finished_return_sum.sum().item())
        c = int(self.finished_count.sum().item())
        self.finished_return_sum.zero_()
        self.finished_count.zero_()
        return s, c


class SyntheticVecEnv(_VecEnv):
    """Shape-only stand-in (observations ~ N(0,1), reward 1 or ~N(0,1), Bernoulli terminals):
    ppo_synth_env_step.  Used where only throughput matters."""

    def __init__(self, n_envs, state_dim, action_dim, continuous, reward_kind=0, p_terminal=1.0 / 200.0,
                 horizon=1000, numpy_api=False):
        super().__init__(n_envs, state_dim, numpy_api)
        self.action_space = _Space(shape=(action_dim,)) if continuous else _Space(n=action_dim)
        self.reward_kind, self.p_terminal, self.horizon = reward_kind, p_terminal, horizon
        self._t = 0

    def _emit(self, with_reward):
        _lib.check(_lib.lib().ppo_synth_env_step(
            self._obs.data_ptr(), self._rew.data_ptr() if with_reward else None,
            self._done.data_ptr() if with_reward else None, self.n_envs, self.observation_space.shape[0],
            self.reward_kind, self.p_terminal, self._t, self.horizon, self._seed, _lib.stream_ptr()))
        self._t += 1

    def reset(self):
        self._emit(False)
        return self._fetch()[0] if self.numpy_api else self._obs

    def step(self, action):
        self._emit(True)
        return self._out()


def make(env_name, n_envs=1, numpy_api=None, **kw):
    """gym.make-style constructor (train.py:136).  numpy_api defaults to True for a single env."""
    if numpy_api is None:
        numpy_api = n_envs == 1
    if env_name in ("CartPole-v1", "CartPole-v0"):
        env = CartPoleVecEnv(n_envs, numpy_api)
        if env_name == "CartPole-v0":
            env.max_episode_steps = 200
        return env
    if env_name.startswith("Synthetic"):
        return SyntheticVecEnv(n_envs, numpy_api=numpy_api, **kw)
    raise ValueError(f"unknown environment {env_name!r}: this engine ships CartPole-v1 on the device and "
                     f"a synthetic shape-only env; any object with the gym step/reset API works with PPO")
