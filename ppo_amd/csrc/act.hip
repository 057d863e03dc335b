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
                                                  float* __restrict__ entropy) {
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
        out_layer<1>(LC, 1, H2, h, v);
        if (valid && h == 0 && values) values[m] = v[0];
        return;
    }
    const NetLds LA = carve_net_lds(smem, S, AMAX, false);
    fill_net_lds(LA, params, PA, S, A, AMAX, false);
    __syncthreads();

    mlp_fwd_tile(LA, states, S, m, valid, c, h, H1, H2);
    float o[AMAX];
    out_layer<AMAX>(LA, A, H2, h, o);

    float logp, ent;
    if constexpr (CONT) {
        float act[AMAX];
        float* ap = reinterpret_cast<float*>(actions_io);
        if constexpr (SAMPLE) {
            // MultivariateNormal.rsample: loc + scale_tril @ eps (multivariate_normal.py:250-253)
            Philox rng(seed, rng_offset, (uint64_t)m);
            const float sigma = sqrtf(action_var);
#pragma unroll
            for (int a = 0; a < AMAX; a += 4) {
                uint32_t u[4];
                rng.next(u);
                float e[4];
                box_muller(u[0], u[1], e[0], e[1]);
                box_muller(u[2], u[3], e[2], e[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (a + j < AMAX) act[a + j] = tanh_head(o[a + j]) + sigma * e[j];
            }
        } else {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) act[a] = (valid && a < A) ? ap[m * A + a] : 0.f;
        }
        GaussOut<AMAX> g;
        gaussian_head<AMAX>(o, A, action_var, act, g);
        logp = g.logp;
        ent = g.ent;
        if (SAMPLE && valid && h == 0) {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) if (a < A) ap[m * A + a] = act[a];
        }
    } else {
        CatOut<AMAX> cat;
        categorical_head<AMAX>(o, A, cat);
        int* ap = reinterpret_cast<int*>(actions_io);
        int action;
        if constexpr (SAMPLE) {
            Philox rng(seed, rng_offset, (uint64_t)m);
            uint32_t u[4];
            rng.next(u);
            const float uu = 1.0f - u01(u[0]);  // [0, 1)
            float cum = 0.f;
            action = A - 1;
            bool done = false;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    cum += cat.q[a];
                    if (!done && uu < cum) { action = a; done = true; }
                }
            }
            if (valid && h == 0) ap[m] = action;
        } else {
            action = valid ? ap[m] : 0;
        }
        logp = cat.L[0];
#pragma unroll
        for (int a = 1; a < AMAX; ++a) logp = (a == action) ? cat.L[a] : logp;
        ent = cat.ent;
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
                      hipStream_t st) {
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
                                                          values, entropy);                     \
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

// Single-env latency path (select_action as train.py:176 calls it, one state per call): host state
// in, host action out, one call.  The state is copied into its buffer row, the batched kernel acts
// on that one row, the action is copied back and the stream is waited for; with page-locked host
// buffers both copies are asynchronous and the only wait is the final one.
extern "C" int ppo_act_host(const float* params, int state_dim, int action_dim, int continuous,
                            float action_var, const float* host_state, float* state_row,
                            uint64_t seed, uint64_t rng_offset, void* action_row, float* logprob_row,
                            float* value_row, void* host_action_out, void* stream) {
    if (!host_state || !state_row || !host_action_out) return PPO_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(state_row, host_state, (size_t)state_dim * sizeof(float),
                                  hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    const int rc = launch_act<true>(params, state_dim, action_dim, continuous, action_var, state_row, 1,
                                    seed, rng_offset, action_row, logprob_row, value_row, nullptr, st);
    if (rc) return rc;
    const size_t abytes = continuous ? (size_t)action_dim * sizeof(float) : sizeof(int);
    e = hipMemcpyAsync(host_action_out, action_row, abytes, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(st);
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
    Philox rng(seed ^ 0x9E3779B97F4A7C15ull, (uint64_t)step, (uint64_t)e);
    uint32_t u[4];
    rng.next(u);
    float n0, n1;
    box_muller(u[0], u[1], n0, n1);
    if (rewards) rewards[e] = reward_kind == 0 ? 1.0f : n0;
    if (terminals) {
        bool t;
        if (p_terminal < 0.f) t = (step % horizon) == horizon - 1;
        else t = (1.0f - u01(u[2])) < p_terminal;
        terminals[e] = t ? 1 : 0;
    }
    if (states) {
        for (int k = 0; k < S; k += 4) {
            rng.next(u);
            float v[4];
            box_muller(u[0], u[1], v[0], v[1]);
            box_muller(u[2], u[3], v[2], v[3]);
            for (int j = 0; j < 4 && k + j < S; ++j) states[e * S + k + j] = v[j];
        }
    }
}

extern "C" int ppo_synth_env_step(float* states_out, float* rewards_out, uint8_t* terminals_out,
                                  int64_t n_envs, int state_dim, int reward_kind, float p_terminal,
                                  int64_t step, int64_t horizon, uint64_t seed, void* stream);

extern "C" int ppo_synth_rollout(const float* params, int state_dim, int action_dim, int continuous,
                                 float action_var, int64_t n_envs, int64_t n_steps, int reward_kind,
                                 float p_terminal, int64_t step0, int64_t horizon, uint64_t env_seed,
                                 uint64_t act_seed, uint64_t act_offset0, float* states, float* rewards,
                                 uint8_t* terminals, void* actions, float* logprobs, float* values,
                                 void* stream) {
    if (n_steps <= 0 || n_envs <= 0) return PPO_E_BADARG;
    const size_t arow = continuous ? (size_t)action_dim * sizeof(float) : sizeof(int);
    for (int64_t t = 0; t < n_steps; ++t) {
        const size_t r0 = (size_t)t * n_envs;
        int rc = ppo_synth_env_step(states + r0 * state_dim, rewards + r0, terminals + r0, n_envs,
                                    state_dim, reward_kind, p_terminal, step0 + t, horizon, env_seed,
                                    stream);
        if (rc) return rc;
        rc = launch_act<true>(params, state_dim, action_dim, continuous, action_var,
                              states + r0 * state_dim, n_envs, act_seed, act_offset0 + t,
                              (char*)actions + r0 * arow, logprobs + r0, values + r0, nullptr,
                              (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
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
