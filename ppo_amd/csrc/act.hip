// Lockstep batched act() / evaluate() forward, and the synthetic vectorised env step.
// One wave = 32 states; both MLPs run on the fp32 MFMA tile of ppo_common.h.
#include "ppo_common.h"

template <bool CONT, int AMAX, bool SAMPLE>
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
    // states_copy (optional): the observations are also written there (the rollout buffer's rows when
    // `states` is a page-locked host buffer); actions_dup (optional): second destination of the
    // sampled actions (a page-locked host buffer the caller reads after the stream has drained).
    // Even workgroups run the actor on four 32-sample tiles, odd ones the critic on the same
    // tiles: each stages only its own network into LDS and the two halves of a step run side by side.
    extern __shared__ __align__(16) float smem[];
    const NetLayout PA = net_layout(0, S, A);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int64_t m = ((int64_t)(blockIdx.x >> 1) * 4 + wave) * TILE_M + c;
    const bool valid = m < n;
    f32x16 H1[2], H2[2];
    if (blockIdx.x & 1) {
        const NetLayout PC = net_layout(PA.n, S, 1);
        const NetLds LC = carve_net_lds(smem, S, 1, false);
        fill_net_lds(LC, params, PC, S, 1, 1, false);
        __syncthreads();
        mlp_fwd_tile(LC, states, S, m, valid, c, h, H1, H2);
        float v[1];
        out_layer_act<1>(LC, 1, H2, h, v);
        if (valid && h == 0 && values) values[m] = v[0];
        if (states_copy && valid && h == 0)
            for (int k = 0; k < S; ++k) states_copy[m * S + k] = states[m * S + k];
        return;
    }
    const NetLds LA = carve_net_lds(smem, S, AMAX, false);
    fill_net_lds(LA, params, PA, S, A, AMAX, false);
    __syncthreads();

    mlp_fwd_tile(LA, states, S, m, valid, c, h, H1, H2);
    float o[AMAX];
    out_layer_act<AMAX>(LA, A, H2, h, o);

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

template <bool SAMPLE>
static int launch_act(const float* params, int S, int A, int continuous, float action_var,
                      const float* states, int64_t n, uint64_t seed, uint64_t rng_offset,
                      void* actions, float* logprobs, float* values, float* entropy,
                      hipStream_t st, float* states_copy = nullptr, void* actions_dup = nullptr) {
    if (!params || !states || !actions || n <= 0) return PPO_E_BADARG;
    if (S < 1 || S > PPO_MAX_STATE_DIM || A < 1 || A > PPO_MAX_ACTION_DIM) return PPO_E_UNSUPPORTED;
    const int amax = A <= 4 ? 4 : 8;
    const size_t lds = (size_t)net_lds_floats(S, amax, false) * 4;  // the actor's image is the larger
    const int64_t tiles = (n + TILE_M - 1) / TILE_M;
    const int grid = 2 * (int)((tiles + 3) / 4);
#define ACT_LAUNCH(C, AM)                                                                       \
    {                                                                                           \
        static size_t configured = 0;                                                           \
        if (lds > configured) {                                                                 \
            const hipError_t e = hipFuncSetAttribute((const void*)act_kernel<C, AM, SAMPLE>,    \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)lds);                                 \
            if (e != hipSuccess) return (int)e;                                                 \
            configured = lds;                                                                   \
        }                                                                                       \
        act_kernel<C, AM, SAMPLE><<<grid, 256, lds, st>>>(params, S, A, action_var, states, n,  \
                                                          seed, rng_offset, actions, logprobs,  \
                                                          values, entropy, states_copy,         \
                                                          actions_dup);                         \
    }
    if (continuous) {
        if (amax == 4) ACT_LAUNCH(true, 4) else ACT_LAUNCH(true, 8)
    } else {
        if (amax == 4) ACT_LAUNCH(false, 4) else ACT_LAUNCH(false, 8)
    }
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
