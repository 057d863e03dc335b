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
    double v[4];
    cartpole_reset_draw(a.seed, ep, e, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a.state[e * 4 + k] = v[k];
        a.obs[e * 4 + k] = (float)v[k];
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
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n) return;
    double s[4] = {a.state[e * 4 + 0], a.state[e * 4 + 1], a.state[e * 4 + 2], a.state[e * 4 + 3]};
    const bool fell = cartpole_advance(s, a.actions[e]);
    const int steps = a.steps[e] + 1;
    const bool done = fell || steps >= a.max_steps;
    a.rewards[e] = 1.0f;
    a.terminals[e] = done ? 1 : 0;
    if (done) {
        if (a.fin_sum) a.fin_sum[e] += (float)steps;
        if (a.fin_cnt) a.fin_cnt[e] += 1;
        cartpole_reset_one(a, e);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.state[e * 4 + k] = s[k];
            a.obs[e * 4 + k] = (float)s[k];
        }
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


// env.step(action) of ONE env as the reference's loop calls it (train.py:177): the action travels in
// the launch's arguments, observation / reward / terminal flag are written straight into page-locked
// host memory, then a ticket word on which the host spins: one launch, no copy commands, no stream
// wait.  The same step as cartpole_step_kernel's thread 0, bit for bit.
__global__ __launch_bounds__(64) void cartpole_step_host_kernel(const CartPoleArgs a, int action,
                                                                unsigned* host_ticket, unsigned ticket) {
    if (threadIdx.x != 0) return;
    double s[4] = {a.state[0], a.state[1], a.state[2], a.state[3]};
    const bool fell = cartpole_advance(s, action);
    const int steps = a.steps[0] + 1;
    const bool done = fell || steps >= a.max_steps;
    a.rewards[0] = 1.0f;
    a.obs[5] = done ? 1.0f : 0.0f;        // host record: [obs 0..3 | reward | terminal flag]
    if (done) {
        if (a.fin_sum) a.fin_sum[0] += (float)steps;
        if (a.fin_cnt) a.fin_cnt[0] += 1;
        cartpole_reset_one(a, 0);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.state[k] = s[k];
            a.obs[k] = (float)s[k];
        }
        a.steps[0] = steps;
    }
    __threadfence_system();
    __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int ppo_cartpole_step_host(double* state, int* steps, unsigned* episodes, float* fin_sum,
                                      int* fin_cnt, int action, float* host_out, unsigned* host_ticket,
                                      unsigned ticket, int max_episode_steps, uint64_t seed, void* stream) {
    if (!state || !steps || !episodes || !host_out || !host_ticket || max_episode_steps < 1 ||
        (action != 0 && action != 1))
        return PPO_E_BADARG;
    if (__atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket) return PPO_E_BADARG;
    CartPoleArgs a{};
    a.state = state; a.steps = steps; a.episodes = episodes; a.fin_sum = fin_sum; a.fin_cnt = fin_cnt;
    a.obs = host_out; a.rewards = host_out + 4; a.n = 1; a.max_steps = max_episode_steps; a.seed = seed;
    hipStream_t st = (hipStream_t)stream;
    cartpole_step_host_kernel<<<1, 64, 0, st>>>(a, action, host_ticket, ticket);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // spin on the ticket; every so often ask the stream, so that a kernel that ended without writing it
    // (a fault) surfaces as its error instead of a hang
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket) return 0;
        if ((spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return __atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket ? 0 : (int)hipErrorUnknown;
            if (q != hipErrorNotReady) return (int)q;
        }
    }
}
