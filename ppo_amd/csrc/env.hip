// Device-side vectorised CartPole-v1: the environment on the other side of select_action in the
// reference's training loop (env.step / env.reset, train.py:170-183, created by
// gym.make("CartPole-v1"), train.py:19,136).  gym itself is not part of the reference tree; the
// dynamics below are the published CartPole-v1 definition (Barto, Sutton & Anderson cart-pole,
// Euler integration, tau = 0.02 s, force +-10 N, episode ends when |x| > 2.4, |theta| > 12 degrees
// or after 500 steps; reward 1 per step; reset draws the four state variables from U(-0.05, 0.05)).
// One thread per env; state kept in fp64 as gym does, observations emitted in fp32.  An env whose
// episode ends is reset in the same call, so the observation written is always the one the next
// select_action has to see (what train.py gets from env.reset() at the top of its episode loop).
#include "ppo_common.h"

struct CartPoleArgs {
    double* state;        // [n, 4]  x, x_dot, theta, theta_dot
    int* steps;           // [n]     steps taken in the running episode
    unsigned* episodes;   // [n]     episodes started so far (reset RNG counter)
    float* fin_sum;       // [n]     sum of returns of the episodes this env has finished
    int* fin_cnt;         // [n]     number of episodes this env has finished
    const int* actions;   // [n]     0: push left, 1: push right
    float* obs;           // [n, 4]
    float* rewards;       // [n]
    uint8_t* terminals;   // [n]
    int64_t n;
    int max_steps;
    uint64_t seed;
};

__device__ __forceinline__ void cartpole_reset_one(const CartPoleArgs& a, int64_t e) {
    const unsigned ep = a.episodes[e];
    Philox rng(a.seed ^ 0xC2B2AE3D27D4EB4Full, (uint64_t)ep, (uint64_t)e);
    uint32_t u[4];
    rng.next(u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // 32 random bits -> [0, 1) exactly representable in fp64 -> U(-0.05, 0.05)
        const double v = ((double)u[k] * (1.0 / 4294967296.0) - 0.5) * 0.1;
        a.state[e * 4 + k] = v;
        a.obs[e * 4 + k] = (float)v;
    }
    a.steps[e] = 0;
    a.episodes[e] = ep + 1;
}

__global__ __launch_bounds__(256) void cartpole_reset_kernel(const CartPoleArgs a) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n) return;
    cartpole_reset_one(a, e);
    if (a.fin_sum) a.fin_sum[e] = 0.f;
    if (a.fin_cnt) a.fin_cnt[e] = 0;
}

__global__ __launch_bounds__(256) void cartpole_step_kernel(const CartPoleArgs a) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n) return;
    const double gravity = 9.8, masspole = 0.1, total_mass = 1.1, length = 0.5,
                 polemass_length = 0.05, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    double x = a.state[e * 4 + 0], xd = a.state[e * 4 + 1], th = a.state[e * 4 + 2],
           thd = a.state[e * 4 + 3];
    const double force = a.actions[e] == 1 ? force_mag : -force_mag;
    const double ct = cos(th), st = sin(th);
    const double temp = (force + polemass_length * (thd * thd) * st) / total_mass;
    const double thacc = (gravity * st - ct * temp) /
                         (length * (4.0 / 3.0 - masspole * (ct * ct) / total_mass));
    const double xacc = temp - polemass_length * thacc * ct / total_mass;
    x = x + tau * xd;
    xd = xd + tau * xacc;
    th = th + tau * thd;
    thd = thd + tau * thacc;
    const int steps = a.steps[e] + 1;
    const bool fell = x < -x_thr || x > x_thr || th < -theta_thr || th > theta_thr;
    const bool done = fell || steps >= a.max_steps;
    a.rewards[e] = 1.0f;
    a.terminals[e] = done ? 1 : 0;
    if (done) {
        if (a.fin_sum) a.fin_sum[e] += (float)steps;
        if (a.fin_cnt) a.fin_cnt[e] += 1;
        cartpole_reset_one(a, e);
    } else {
        a.state[e * 4 + 0] = x;
        a.state[e * 4 + 1] = xd;
        a.state[e * 4 + 2] = th;
        a.state[e * 4 + 3] = thd;
        a.obs[e * 4 + 0] = (float)x;
        a.obs[e * 4 + 1] = (float)xd;
        a.obs[e * 4 + 2] = (float)th;
        a.obs[e * 4 + 3] = (float)thd;
        a.steps[e] = steps;
    }
}

extern "C" int ppo_cartpole_reset(double* state, int* steps, unsigned* episodes, float* fin_sum,
                                  int* fin_cnt, float* obs_out, int64_t n_envs, uint64_t seed,
                                  void* stream) {
    if (!state || !steps || !episodes || !obs_out || n_envs <= 0) return PPO_E_BADARG;
    CartPoleArgs a{};
    a.state = state; a.steps = steps; a.episodes = episodes; a.fin_sum = fin_sum; a.fin_cnt = fin_cnt;
    a.obs = obs_out; a.n = n_envs; a.seed = seed;
    cartpole_reset_kernel<<<(int)((n_envs + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}

extern "C" int ppo_cartpole_step(double* state, int* steps, unsigned* episodes, float* fin_sum,
                                 int* fin_cnt, const int* actions, float* obs_out,
                                 float* rewards_out, uint8_t* terminals_out, int64_t n_envs,
                                 int max_episode_steps, uint64_t seed, void* stream) {
    if (!state || !steps || !episodes || !actions || !obs_out || !rewards_out || !terminals_out ||
        n_envs <= 0 || max_episode_steps < 1)
        return PPO_E_BADARG;
    CartPoleArgs a{};
    a.state = state; a.steps = steps; a.episodes = episodes; a.fin_sum = fin_sum; a.fin_cnt = fin_cnt;
    a.actions = actions; a.obs = obs_out; a.rewards = rewards_out; a.terminals = terminals_out;
    a.n = n_envs; a.max_steps = max_episode_steps; a.seed = seed;
    cartpole_step_kernel<<<(int)((n_envs + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}

// A whole lockstep rollout on the device, enqueued by one call: for t in [0, n_steps) the policy
// acts on row t of the time-major buffers (ppo_act, rng_offset = act_offset0 + t) and the envs step
// with those actions, writing reward/terminal into row t and the next observation into row t + 1
// (into obs_carry after the last step).  Row 0 starts from obs_carry, i.e. from where the previous
// rollout (or ppo_cartpole_reset) left the envs.  Same launches as the per-step calls.
extern "C" int ppo_cartpole_rollout(const float* params, double* state, int* steps, unsigned* episodes,
                                    float* fin_sum, int* fin_cnt, float* obs_carry, int64_t n_envs,
                                    int64_t n_steps, int max_episode_steps, uint64_t env_seed,
                                    uint64_t act_seed, uint64_t act_offset0, float* states,
                                    float* rewards, uint8_t* terminals, int* actions, float* logprobs,
                                    float* values, void* stream) {
    if (!params || !obs_carry || !states || !rewards || !terminals || !actions || !logprobs || !values ||
        n_envs <= 0 || n_steps <= 0)
        return PPO_E_BADARG;
    hipError_t e = hipMemcpyAsync(states, obs_carry, (size_t)n_envs * 4 * sizeof(float),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    for (int64_t t = 0; t < n_steps; ++t) {
        const size_t r0 = (size_t)t * n_envs;
        int rc = ppo_act(params, 4, 2, 0, 0.f, states + r0 * 4, n_envs, act_seed, act_offset0 + t,
                         actions + r0, logprobs + r0, values + r0, stream);
        if (rc) return rc;
        float* next_obs = t + 1 < n_steps ? states + (r0 + n_envs) * 4 : obs_carry;
        rc = ppo_cartpole_step(state, steps, episodes, fin_sum, fin_cnt, actions + r0, next_obs,
                               rewards + r0, terminals + r0, n_envs, max_episode_steps, env_seed, stream);
        if (rc) return rc;
    }
    return 0;
}
