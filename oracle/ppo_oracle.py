"""CPU restatement (numpy, fp32) of the reference's PPO hot path.

TEST INFRASTRUCTURE ONLY.  This module is the checker for the HIP engine: it may
be imported by tests/, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline``
leg of ``bench.py`` and by nothing else.  The product path (``ppo_amd``) never
imports it and has no CPU fallback.

Parity status: PINNED.  Every function below is checked against golden vectors
captured from the reference's own execution (tests/golden/gen_golden.py ->
tests/golden/*.npz, torch 2.10.0+rocm7.0 CPU kernels) by tests/test_oracle.py.
The arithmetic the reference delegates to PyTorch (not under /root/reference,
not version-pinned by the reference) is restated from the torch sources cited
per function.

Everything is hand-derived: forward, the Categorical / diagonal-Gaussian
log-prob and entropy, the clipped surrogate, the backward pass and Adam.  No
autograd is used.
"""
import math

import numpy as np

F32 = np.float32
ACTOR_KEYS = ["actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias",
              "actor.4.weight", "actor.4.bias"]
CRITIC_KEYS = ["critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias",
               "critic.4.weight", "critic.4.bias"]
PARAM_KEYS = ACTOR_KEYS + CRITIC_KEYS  # state_dict order of ActorCritic, PPO.py:49-73
EPS_PROB = F32(np.finfo(np.float32).eps)  # torch/distributions/utils.py:121 clamp_probs eps
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- returns
def returns_scan(rewards, terminals, gamma):
    """Monte-Carlo discounted returns with terminal resets, PPO.py:202-208.

    Python-float (fp64) accumulation in reverse order, seed 0 (PPO.py:203); the
    reset happens BEFORE the step that is flagged terminal adds its reward
    (PPO.py:205-207).  Returns the fp64 sequence; PPO.py:211 rounds it to fp32.
    """
    rewards = np.asarray(rewards, dtype=np.float64)
    out = np.empty(rewards.shape[0], dtype=np.float64)
    d = 0.0
    g = float(gamma)
    for t in range(rewards.shape[0] - 1, -1, -1):
        if terminals[t]:
            d = 0.0
        d = float(rewards[t]) + g * d
        out[t] = d
    return out


def returns_scan_rows(rewards_nt, terminals_nt, gamma):
    """Vectorised-env variant: row n is one env's T steps; every row is scanned on its own
    (a row end truncates the return exactly like the buffer end does in PPO.py:203).
    Equivalent to the flat scan of the row-major concatenation with the last step of every
    row flagged terminal."""
    r = np.asarray(rewards_nt, dtype=np.float64)
    out = np.empty_like(r)
    d = np.zeros(r.shape[0], dtype=np.float64)
    g = float(gamma)
    for t in range(r.shape[1] - 1, -1, -1):
        d = np.where(terminals_nt[:, t], 0.0, d)
        d = r[:, t] + g * d
        out[:, t] = d
    return out


def normalize_returns(returns64):
    """PPO.py:211-212: round to fp32, subtract the mean, divide by (unbiased std + 1e-7).

    The moments are accumulated in fp64 here and rounded to fp32 scalars (torch's fp32
    reductions agree with that to ~1e-7 relative); the elementwise part is fp32.
    """
    r32 = np.asarray(returns64, dtype=np.float64).astype(F32)
    n = r32.shape[0]
    mean64 = r32.astype(np.float64).sum() / n
    var64 = ((r32.astype(np.float64) - mean64) ** 2).sum() / max(n - 1, 1)
    mean = F32(mean64)
    std = F32(math.sqrt(var64))
    norm = (r32 - mean) / (std + F32(1e-7))
    return r32, mean, std, norm.astype(F32)


# --------------------------------------------------------------------------- networks
def mlp_forward(params, prefix, x):
    """Linear(S,64)-Tanh-Linear(64,64)-Tanh-Linear(64,out), PPO.py:49-73. weight is [out,in]."""
    W1, b1 = params[prefix + ".0.weight"], params[prefix + ".0.bias"]
    W2, b2 = params[prefix + ".2.weight"], params[prefix + ".2.bias"]
    W3, b3 = params[prefix + ".4.weight"], params[prefix + ".4.bias"]
    h1 = np.tanh(x @ W1.T + b1).astype(F32)
    h2 = np.tanh(h1 @ W2.T + b2).astype(F32)
    o = (h2 @ W3.T + b3).astype(F32)
    return h1, h2, o


def mlp_backward(params, prefix, x, h1, h2, do):
    """Gradients of sum(do * o) w.r.t. the six tensors of one MLP (no input gradient)."""
    W2, W3 = params[prefix + ".2.weight"], params[prefix + ".4.weight"]
    g = {}
    g[prefix + ".4.weight"] = (do.T @ h2).astype(F32)
    g[prefix + ".4.bias"] = do.sum(0).astype(F32)
    dz2 = ((do @ W3) * (F32(1) - h2 * h2)).astype(F32)
    g[prefix + ".2.weight"] = (dz2.T @ h1).astype(F32)
    g[prefix + ".2.bias"] = dz2.sum(0).astype(F32)
    dz1 = ((dz2 @ W2) * (F32(1) - h1 * h1)).astype(F32)
    g[prefix + ".0.weight"] = (dz1.T @ x).astype(F32)
    g[prefix + ".0.bias"] = dz1.sum(0).astype(F32)
    return g


def categorical_head(o, actions):
    """Softmax (PPO.py:64) -> Categorical(probs) log_prob / entropy (PPO.py:116-118).

    torch/distributions/categorical.py:71 renormalises probs; :111 + utils.py:121-137 give
    logits = log(clamp(probs, eps, 1-eps)); :150-155 log_prob gathers logits; :157-161
    entropy = -sum(logits * probs).  Also returns what the backward needs.
    """
    m = o.max(-1, keepdims=True)
    e = np.exp(o - m).astype(F32)
    p = (e / e.sum(-1, keepdims=True)).astype(F32)
    q = (p / p.sum(-1, keepdims=True)).astype(F32)
    qc = np.clip(q, EPS_PROB, F32(1) - EPS_PROB)
    L = np.log(qc).astype(F32)
    pass_grad = ((q >= EPS_PROB) & (q <= F32(1) - EPS_PROB)).astype(F32)  # clamp: closed interval
    idx = np.arange(o.shape[0])
    logp = L[idx, actions]
    ent = (-(L * q).sum(-1)).astype(F32)
    return logp, ent, (q, L, pass_grad)


def categorical_backward(cache, actions, dlogp, dent):
    """d/d(pre-softmax logits) of sum(dlogp*logp + dent*entropy)."""
    q, L, c = cache
    n, A = q.shape
    onehot = np.zeros((n, A), dtype=F32)
    onehot[np.arange(n), actions] = 1
    ca = c[np.arange(n), actions][:, None]
    dz_logp = ca * (onehot - q)
    w = L + c
    dz_ent = -(q * (w - (w * q).sum(-1, keepdims=True)))
    return (dlogp[:, None] * dz_logp + dent[:, None] * dz_ent).astype(F32)


def action_sigma(action_std):
    """PPO.py:46/77: var = fp32(std*std) (product in Python double); the Cholesky factor of
    diag(var) is its fp32 sqrt (torch/distributions/multivariate_normal.py:193)."""
    var = F32(float(action_std) * float(action_std))
    return var, F32(np.sqrt(var))


def gaussian_head(o, actions, action_std):
    """Tanh head (PPO.py:55) -> MultivariateNormal(mean, diag(var)) (PPO.py:105-118).

    multivariate_normal.py:255-263: logp = -0.5*(A*log(2pi) + M) - sum(log sigma),
    M = sum(((a-mu)/sigma)^2);  :265-273 entropy = 0.5*A*(1+log(2pi)) + sum(log sigma).
    """
    A = o.shape[1]
    var, sigma = action_sigma(action_std)
    mu = np.tanh(o).astype(F32)
    diff = (actions.reshape(-1, A).astype(F32) - mu).astype(F32)
    zed = (diff / sigma).astype(F32)
    M = (zed * zed).sum(-1).astype(F32)
    half_log_det = F32(A) * np.log(sigma).astype(F32)
    logp = (F32(-0.5) * (F32(A * LOG_2PI) + M) - half_log_det).astype(F32)
    ent = np.full(o.shape[0], F32(0.5 * A * (1.0 + LOG_2PI)) + half_log_det, dtype=F32)
    return logp, ent, (mu, zed, sigma)


def gaussian_backward(cache, dlogp):
    """d/d(pre-tanh output) of sum(dlogp*logp); the entropy has no gradient (fixed variance)."""
    mu, zed, sigma = cache
    dmu = dlogp[:, None] * (zed / sigma)
    return (dmu * (F32(1) - mu * mu)).astype(F32)


def evaluate(params, states, actions, continuous, action_std=None):
    """ActorCritic.evaluate, PPO.py:102-121 -> (logprobs, state_values, entropy, cache)."""
    x = np.asarray(states, dtype=F32)
    ah1, ah2, ao = mlp_forward(params, "actor", x)
    ch1, ch2, co = mlp_forward(params, "critic", x)
    if continuous:
        logp, ent, hc = gaussian_head(ao, np.asarray(actions), action_std)
    else:
        logp, ent, hc = categorical_head(ao, np.asarray(actions).astype(np.int64))
    return logp, co.reshape(-1), ent, (x, ah1, ah2, ch1, ch2, hc)


def ppo_loss_and_grads(params, states, actions, old_logprobs, advantages, returns_norm,
                       eps_clip, continuous, action_std=None, n_global=None):
    """One epoch body up to backward(), PPO.py:227-244.

    loss_t = -min(r*A, clamp(r,1-e,1+e)*A) + 0.5*mean((V-R)^2) - 0.01*H_t; the gradient is of
    loss.mean().  ``n_global`` lets a shard scale its gradient by the global sample count.
    """
    T = states.shape[0]
    n = T if n_global is None else n_global
    logp, V, ent, cache = evaluate(params, states, actions, continuous, action_std)
    x, ah1, ah2, ch1, ch2, hc = cache
    ratio = np.exp(logp - old_logprobs).astype(F32)
    lo, hi = F32(1.0 - eps_clip), F32(1.0 + eps_clip)
    surr1 = ratio * advantages
    surr2 = np.clip(ratio, lo, hi) * advantages
    mse = F32(((V - returns_norm).astype(np.float64) ** 2).sum() / T)
    loss = (-np.minimum(surr1, surr2) + F32(0.5) * mse - F32(0.01) * ent).astype(F32)
    # backward seeds: min() ties split 1/2+1/2 and clamp passes gradient on the closed interval,
    # so inside the clip range the full gradient flows; outside only the surr1 branch if it is
    # the smaller one.
    gate = np.where(advantages >= 0, ratio <= hi, ratio >= lo).astype(F32)
    gate = np.where((ratio >= lo) & (ratio <= hi), F32(1), gate).astype(F32)
    inv_n = F32(1.0 / n)
    dlogp = (-(advantages * ratio * gate) * inv_n).astype(F32)
    dV = ((V - returns_norm) * inv_n).astype(F32)
    dent = np.full(T, F32(-0.01) * inv_n, dtype=F32)
    if continuous:
        dao = gaussian_backward(hc, dlogp)
    else:
        dao = categorical_backward(hc, np.asarray(actions).astype(np.int64), dlogp, dent)
    grads = mlp_backward(params, "actor", x, ah1, ah2, dao)
    grads.update(mlp_backward(params, "critic", x, ch1, ch2, dV[:, None]))
    out = dict(logprobs=logp, state_values=V, entropy=ent, ratios=ratio, loss=loss,
               loss_mean=F32(loss.astype(np.float64).mean()))
    return out, grads


# --------------------------------------------------------------------------- Adam
def adam_init(params):
    return {"step": 0, "m": {k: np.zeros_like(v) for k, v in params.items()},
            "v": {k: np.zeros_like(v) for k, v in params.items()}}


def adam_step(params, grads, state, lr_actor, lr_critic, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch/optim/adam.py:457-547 (single-tensor path, no weight decay, no amsgrad),
    two parameter groups (PPO.py:139-142).  In place."""
    state["step"] += 1
    step = state["step"]
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    bc2_sqrt = F32(bc2 ** 0.5)
    for k in PARAM_KEYS:
        lr = lr_actor if k.startswith("actor") else lr_critic
        g, m, v = grads[k], state["m"][k], state["v"][k]
        m += F32(1.0 - beta1) * (g - m)                       # lerp_, weight < 0.5 form
        v *= F32(beta2)
        v += (F32(1.0 - beta2) * g) * g                       # addcmul_
        denom = np.sqrt(v) / bc2_sqrt + F32(eps)
        params[k] += (F32(-(lr / bc1)) * m) / denom           # addcdiv_
    return params


# --------------------------------------------------------------------------- update
def update(params, adam_state, states, actions, old_logprobs, old_values, rewards, terminals,
           gamma, eps_clip, K_epochs, lr_actor, lr_critic, continuous, action_std=None,
           record=False):
    """PPO.update, PPO.py:200-251, on explicit arrays.  Mutates params/adam_state."""
    ret64 = returns_scan(rewards, terminals, gamma)
    ret32, mean, std, ret_n = normalize_returns(ret64)
    adv = (ret_n - np.asarray(old_values, dtype=F32)).astype(F32)
    rec = dict(returns64=ret64, returns32=ret32, ret_mean=mean, ret_std=std, returns_norm=ret_n,
               advantages=adv, loss_mean_all=[])
    for e in range(K_epochs):
        out, grads = ppo_loss_and_grads(params, np.asarray(states, dtype=F32), actions,
                                        np.asarray(old_logprobs, dtype=F32), adv, ret_n, eps_clip,
                                        continuous, action_std)
        if e == 0 and record:
            rec.update({k + "_e1": v for k, v in out.items()})
            rec["grads_e1"] = {k: v.copy() for k, v in grads.items()}
        rec["loss_mean_all"].append(out["loss_mean"])
        adam_step(params, grads, adam_state, lr_actor, lr_critic)
    rec["loss_mean_all"] = np.asarray(rec["loss_mean_all"], dtype=F32)
    return rec


# --------------------------------------------------------------------------- act
def act_logprob_value(params, states, actions, continuous, action_std=None):
    """What ActorCritic.act (PPO.py:86-100) returns for GIVEN (state, action) pairs:
    log-prob of the action under policy_old and the critic value."""
    logp, V, _, _ = evaluate(params, states, actions, continuous, action_std)
    return logp, V


def decay_action_std(action_std, rate, min_std):
    """PPO.decay_action_std, PPO.py:162-165."""
    s = round(action_std - rate, 4)
    if s <= min_std:
        s = min_std
    return s


# --------------------------------------------------------------------------- param packing
def param_shapes(S, A):
    shp = {}
    for pre, out in (("actor", A), ("critic", 1)):
        shp[pre + ".0.weight"] = (64, S)
        shp[pre + ".0.bias"] = (64,)
        shp[pre + ".2.weight"] = (64, 64)
        shp[pre + ".2.bias"] = (64,)
        shp[pre + ".4.weight"] = (out, 64)
        shp[pre + ".4.bias"] = (out,)
    return shp


def flatten_params(params):
    return np.concatenate([np.asarray(params[k], dtype=F32).reshape(-1) for k in PARAM_KEYS])


def unflatten_params(flat, S, A):
    out, off = {}, 0
    for k, shp in param_shapes(S, A).items():
        n = int(np.prod(shp))
        out[k] = np.asarray(flat[off:off + n], dtype=F32).reshape(shp).copy()
        off += n
    return out
