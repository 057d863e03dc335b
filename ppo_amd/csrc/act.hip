// Lockstep batched act() / evaluate() forward, and the synthetic vectorised env step.
// One wave = 16 or 32 states; both MLPs run on the fp32 MFMA tiles of ppo_common.h.
#include "ppo_common.h"
#include <stdlib.h>

template <bool CONT, int AMAX, bool SAMPLE, int TM>
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ params, int S, int A,
                                                  float action_var,
                                                  const float* __restrict__ states, int64_t n,
                                                  uint64_t seed, uint64_t rng_offset,
                                                  void* __restrict__ actions_io,
                                                  float* __restrict__ logprobs,
                                                  float* __restrict__ values,
                                                  float* __restrict__ entropy,
                                                  float* __restrict__ states_copy,
                                                  void* __restrict__ actions_dup) {
    // states_copy (optional): the observations are also written there; actions_dup (optional):
    // second destination of the sampled actions.
    // Even workgroups run the actor on four TM-sample tiles, odd ones the critic on the same
    // tiles: each stages only its own network into LDS and the two halves of a step run side by side.
    // TM = 16 (16x16x4 MFMA) halves the step's dependent chain and is used for small batches, TM = 32
    // amortises the image fill over twice the samples; the results are the same bits.
    extern __shared__ __align__(16) float smem[];
    const NetLayout PA = net_layout(0, S, A);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane / TM, c = lane % TM;
    const int64_t m = ((int64_t)(blockIdx.x >> 1) * 4 + wave) * TM + c;
    const bool valid = m < n;
    const RowFromGlobal x{states, S, m, valid};
    TileNet<TM> net;
    if (blockIdx.x & 1) {
        const NetLayout PC = net_layout(PA.n, S, 1);
        net.stage(smem, params, PC, S, 1, 1);
        __syncthreads();
        float v[1];
        net.template forward<1>(S, 1, x, lane, v);
        if (valid && h == 0 && values) values[m] = v[0];
        if (states_copy && valid && h == 0)
            for (int k = 0; k < S; ++k) states_copy[m * S + k] = states[m * S + k];
        return;
    }
    net.stage(smem, params, PA, S, A, AMAX);
    __syncthreads();

    float o[AMAX];
    net.template forward<AMAX>(S, A, x, lane, o);

    float logp, ent, act_c[AMAX];
    int action;
    policy_head_act<CONT, AMAX, SAMPLE>(o, A, action_var, seed, rng_offset, m, valid, h, actions_io, logp, ent,
                                        action, act_c);
    if (SAMPLE && actions_dup && valid && h == 0) {
        if constexpr (CONT) {
#pragma unroll
            for (int k = 0; k < AMAX; ++k) if (k < A) reinterpret_cast<float*>(actions_dup)[m * A + k] = act_c[k];
        } else {
            reinterpret_cast<int*>(actions_dup)[m] = action;
        }
    }

    if (valid && h == 0) {
        if (logprobs) logprobs[m] = logp;
        if (entropy) entropy[m] = ent;
    }
}

// 16-sample tiles while the launch is latency-bound (the chip is far from full), 32-sample tiles
// beyond; PPO_ACT_TILE=16|32 overrides.
static int act_tile(int64_t n) {
    static const int forced = [] {
        const char* e = getenv("PPO_ACT_TILE");
        return e ? atoi(e) : 0;
    }();
    if (forced == 16 || forced == 32) return forced;
    return n <= 8192 ? 16 : 32;
}

template <bool SAMPLE>
static int launch_act(const float* params, int S, int A, int continuous, float action_var,
                      const float* states, int64_t n, uint64_t seed, uint64_t rng_offset,
                      void* actions, float* logprobs, float* values, float* entropy,
                      hipStream_t st, float* states_copy = nullptr, void* actions_dup = nullptr) {
    if (!params || !states || !actions || n <= 0) return PPO_E_BADARG;
    if (S < 1 || S > PPO_MAX_STATE_DIM || A < 1 || A > PPO_MAX_ACTION_DIM) return PPO_E_UNSUPPORTED;
    const int amax = A <= 4 ? 4 : 8;
    const int tm = act_tile(n);
    const size_t lds = (size_t)(tm == 16 ? TileNet<16>::floats(S, amax) : TileNet<32>::floats(S, amax)) * 4;  // the actor's image is the larger
    const int64_t tiles = (n + tm - 1) / tm;
    const int grid = 2 * (int)((tiles + 3) / 4);
#define ACT_LAUNCH(C, AM, TM)                                                                   \
    {                                                                                           \
        static size_t configured = 0;                                                           \
        if (lds > configured) {                                                                 \
            const hipError_t e = hipFuncSetAttribute((const void*)act_kernel<C, AM, SAMPLE, TM>, \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)lds);                                 \
            if (e != hipSuccess) return (int)e;                                                 \
            configured = lds;                                                                   \
        }                                                                                       \
        act_kernel<C, AM, SAMPLE, TM><<<grid, 256, lds, st>>>(params, S, A, action_var, states, n, \
                                                              seed, rng_offset, actions, logprobs, \
                                                              values, entropy, states_copy,     \
                                                              actions_dup);                     \
    }
#define ACT_LAUNCH_TM(C, AM) { if (tm == 16) ACT_LAUNCH(C, AM, 16) else ACT_LAUNCH(C, AM, 32) }
    if (continuous) {
        if (amax == 4) ACT_LAUNCH_TM(true, 4) else ACT_LAUNCH_TM(true, 8)
    } else {
        if (amax == 4) ACT_LAUNCH_TM(false, 4) else ACT_LAUNCH_TM(false, 8)
    }
#undef ACT_LAUNCH_TM
#undef ACT_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int ppo_act(const float* params, int state_dim, int action_dim, int continuous,
                       float action_var, const float* states, int64_t n, uint64_t seed,
                       uint64_t rng_offset, void* actions, float* logprobs, float* values,
                       void* stream) {
    return launch_act<true>(params, state_dim, action_dim, continuous, action_var, states, n, seed,
                            rng_offset, actions, logprobs, values, nullptr, (hipStream_t)stream);
}

extern "C" int ppo_evaluate(const float* params, int state_dim, int action_dim, int continuous,
                            float action_var, const float* states, const void* actions, int64_t n,
                            float* logprobs, float* values, float* entropy, void* stream) {
    return launch_act<false>(params, state_dim, action_dim, continuous, action_var, states, n, 0, 0,
                             const_cast<void*>(actions), logprobs, values, entropy,
                             (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- synthetic env
__global__ __launch_bounds__(256) void synth_env_kernel(float* __restrict__ states,
                                                        float* __restrict__ rewards,
                                                        uint8_t* __restrict__ terminals,
                                                        int64_t n_envs, int S, int reward_kind,
                                                        float p_terminal, int64_t step,
                                                        int64_t horizon, uint64_t seed) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    float reward;
    bool term;
    synth_env_reward_terminal(seed, step, e, reward_kind, p_terminal, horizon, reward, term);
    if (rewards) rewards[e] = reward;
    if (terminals) terminals[e] = term ? 1 : 0;
    if (states) {
        for (int k = 0; k < S; k += 4) {
            float v[4];
            synth_env_state_block(seed, step, e, k >> 2, v);
            for (int j = 0; j < 4 && k + j < S; ++j) states[e * S + k + j] = v[j];
        }
    }
}

extern "C" int ppo_synth_env_step(float* states_out, float* rewards_out, uint8_t* terminals_out,
                                  int64_t n_envs, int state_dim, int reward_kind, float p_terminal,
                                  int64_t step, int64_t horizon, uint64_t seed, void* stream) {
    if (n_envs <= 0 || state_dim < 1 || horizon < 1) return PPO_E_BADARG;
    const int grid = (int)((n_envs + 255) / 256);
    synth_env_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(states_out, rewards_out, terminals_out,
                                                           n_envs, state_dim, reward_kind,
                                                           p_terminal, step, horizon, seed);
    return (int)hipGetLastError();
}
