"""CPU restatement of the device-side vectorised CartPole-v1 (TEST INFRASTRUCTURE ONLY: imported by
tests/ and nothing else; the product path never touches it).

The reference creates its environment with gym.make("CartPole-v1") (train.py:19,136) and drives it
through env.reset() / env.step(action) (train.py:170,177).  gym is a third-party dependency that is
neither vendored in the reference tree nor installed here, and the reference does not pin its
version, so the physics below restates the published CartPole-v1 definition (Barto, Sutton &
Anderson; Euler integration with tau = 0.02, force 10 N, termination at |x| > 2.4 or
|theta| > 12 degrees, time limit 500, reward 1 per step, reset ~ U(-0.05, 0.05)^4).
PARITY UNPINNED against gym itself: no fixture of gym's outputs exists in the reference.  What the
tests pin is kernel == this restatement (fp64 state, bit-exact up to the last ulp of sin/cos).
"""
import math

import numpy as np

GRAVITY, MASSPOLE, TOTAL_MASS, LENGTH, POLEMASS_LENGTH = 9.8, 0.1, 1.1, 0.5, 0.05
FORCE_MAG, TAU = 10.0, 0.02
THETA_THR = 12.0 * 2.0 * math.pi / 360.0
X_THR = 2.4

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(seed, ctr_hi, ctr_lo):
    """One Philox4x32-10 block; counter = (ctr_lo, ctr_hi) as two 64-bit words, key = seed.
    Vectorised over ctr_lo (uint64 array).  Returns [n, 4] uint32."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    n = ctr_lo.shape[0]
    ctr_hi = np.broadcast_to(np.asarray(ctr_hi, dtype=np.uint64), (n,))
    x0 = ctr_lo & _MASK
    x1 = ctr_lo >> np.uint64(32)
    x2 = ctr_hi & _MASK
    x3 = ctr_hi >> np.uint64(32)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * x0
        p1 = _M1 * x2
        y0 = (p1 >> np.uint64(32)) ^ x1 ^ np.uint64(k0)
        y1 = p1 & _MASK
        y2 = (p0 >> np.uint64(32)) ^ x3 ^ np.uint64(k1)
        y3 = p0 & _MASK
        x0, x1, x2, x3 = y0, y1, y2, y3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([x0, x1, x2, x3], axis=1).astype(np.uint32)


def reset_states(seed, episodes, env_ids):
    """Initial state of episode number `episodes[i]` of env `env_ids[i]`: U(-0.05, 0.05)^4."""
    out = np.empty((len(env_ids), 4), dtype=np.float64)
    key = int(seed) ^ 0xC2B2AE3D27D4EB4F
    for i, (ep, e) in enumerate(zip(episodes, env_ids)):
        u = philox4x32_10(key, np.uint64(ep), np.asarray([e], dtype=np.uint64))[0]
        out[i] = (u.astype(np.float64) * (1.0 / 4294967296.0) - 0.5) * 0.1
    return out


def step(state, actions):
    """One Euler step for every env.  state [n, 4] fp64 (x, x_dot, theta, theta_dot), actions [n]
    in {0, 1}.  Returns (new_state, fell)."""
    x, xd, th, thd = (state[:, k].copy() for k in range(4))
    force = np.where(np.asarray(actions) == 1, FORCE_MAG, -FORCE_MAG)
    ct, st = np.cos(th), np.sin(th)
    temp = (force + POLEMASS_LENGTH * (thd * thd) * st) / TOTAL_MASS
    thacc = (GRAVITY * st - ct * temp) / (LENGTH * (4.0 / 3.0 - MASSPOLE * (ct * ct) / TOTAL_MASS))
    xacc = temp - POLEMASS_LENGTH * thacc * ct / TOTAL_MASS
    x = x + TAU * xd
    xd = xd + TAU * xacc
    th = th + TAU * thd
    thd = thd + TAU * thacc
    new = np.stack([x, xd, th, thd], axis=1)
    fell = (x < -X_THR) | (x > X_THR) | (th < -THETA_THR) | (th > THETA_THR)
    return new, fell


class VecCartPole:
    """Host mirror of the device env including the reset-in-the-same-call convention."""

    def __init__(self, n, seed, max_steps=500):
        self.n, self.seed, self.max_steps = n, seed, max_steps
        self.episodes = np.zeros(n, dtype=np.int64)
        self.steps = np.zeros(n, dtype=np.int64)
        self.state = np.zeros((n, 4))
        self.fin_sum = np.zeros(n, dtype=np.float32)
        self.fin_cnt = np.zeros(n, dtype=np.int64)

    def reset(self):
        ids = np.arange(self.n)
        self.state = reset_states(self.seed, self.episodes, ids)
        self.episodes += 1
        self.steps[:] = 0
        self.fin_sum[:] = 0
        self.fin_cnt[:] = 0
        return self.state.astype(np.float32)

    def step(self, actions):
        new, fell = step(self.state, actions)
        self.steps += 1
        done = fell | (self.steps >= self.max_steps)
        idx = np.nonzero(done)[0]
        self.fin_sum[idx] += self.steps[idx].astype(np.float32)
        self.fin_cnt[idx] += 1
        if len(idx):
            new[idx] = reset_states(self.seed, self.episodes[idx], idx)
            self.episodes[idx] += 1
            self.steps[idx] = 0
        self.state = new
        return new.astype(np.float32), np.ones(self.n, dtype=np.float32), done
