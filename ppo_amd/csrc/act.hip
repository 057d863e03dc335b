// Lockstep batched act() / evaluate() forward, and the synthetic vectorised env step.
// One wave = 32 states; both MLPs run on the fp32 MFMA tile of ppo_common.h.
#include "ppo_common.h"
#include <string.h>

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

// ---------------------------------------------------------------------------- one state, low latency
// select_action as train.py:176 calls it: ONE state per call.  A tile kernel would stage 17 KB of
// weights into LDS to use one column of a 32-column MFMA; here a wave per network holds one hidden
// unit per lane and reads its weight rows straight from global memory.  The fp32 MFMA adds its
// k-slices as a chain of fmaf (one rounding each, lower lane group first), so the same chain on the
// VALU, in the tile's unit order, gives the tile kernel's result bit for bit:
//   layer 1: inputs ascending, then the bias;  layer 2: bias first, then the units in the order
//   0,4,1,5,2,6,3,7 | 8,12,...;  out layer: the four partial sums of out_layer_act.
// The activation a lane needs from another lane comes by v_readlane (layer 2) or through 64 floats
// of LDS (out layer).  Wave 0 = actor (+ head, sampling, the host copy of the action and the
// completion ticket), wave 1 = critic.
struct OneState {   // the observation, passed by value: it reaches the device with the launch itself
    float v[PPO_MAX_STATE_DIM + 1];
};

template <bool CONT, int AMAX>
__global__ __launch_bounds__(128) void act_one_kernel(const float* __restrict__ params, int S, int A,
                                                      float action_var, const OneState state,
                                                      float* __restrict__ state_row, uint64_t seed,
                                                      uint64_t rng_offset, void* __restrict__ action_row,
                                                      float* __restrict__ logprob_row,
                                                      float* __restrict__ value_row,
                                                      void* host_action_out, unsigned* host_ticket,
                                                      unsigned ticket) {
    // every multiply and add below is rounded as written (a prescale product must be rounded before
    // it enters the chain, as the tile kernels' staged images are)
#pragma clang fp contract(off)
    __shared__ float h2s[2][HID];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), u = threadIdx.x & 63;
    const int aout = wave ? 1 : A;
    const NetLayout PA = net_layout(0, S, A);
    const NetLayout P = wave ? net_layout(PA.n, S, 1) : PA;

    // weight rows of this lane's unit: issued up front, consumed below
    f32x4 w2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w2[i] = *reinterpret_cast<const f32x4*>(params + P.w2 + u * HID + 4 * i);
    const float b1 = params[P.b1 + u], b2 = params[P.b2 + u];
    // out layer: lane 4 a + g sums class g of output a (16 units, ascending)
    const int oa = u >> 2, og = u & 3;
    const bool out_lane = oa < aout;
    float w3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w3[i] = out_lane ? params[P.w3 + oa * HID + u16(og, i >> 2, i & 3)] : 0.f;

    float z = 0.f;
    if (wave == 0 && u < S) state_row[u] = state.v[u];
    for (int k0 = 0; k0 < S; k0 += 8) {   // eight loads in flight; a zero weight leaves z as it is
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = k0 + j < S ? params[P.w1 + u * S + k0 + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) z = fmaf(w[j] * TANH_PRESCALE, k0 + j < S ? state.v[k0 + j] : 0.f, z);
    }
    z = b1 * TANH_PRESCALE + z;   // the tile's ones column: fmaf(b1s, 1, z), b1s rounded first
    const int h1 = __float_as_int(tanh_prescaled(z));

    float y = b2 * TANH_PRESCALE;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int k = 8 * (i >> 3) + ((i & 7) >> 1) + 4 * (i & 1);
        y = fmaf(w2[k >> 2][k & 3] * TANH_PRESCALE, __int_as_float(__builtin_amdgcn_readlane(h1, k)), y);
    }
    h2s[wave][u] = tanh_prescaled(y);
    __syncthreads();

    float p = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) p = fmaf(w3[i], h2s[wave][u16(og, i >> 2, i & 3)], p);
    const int pb = __float_as_int(p);
    float o[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        float v = 0.f;
        if (a < aout) {
            const float p0 = __int_as_float(__builtin_amdgcn_readlane(pb, 4 * a + 0));
            const float p1 = __int_as_float(__builtin_amdgcn_readlane(pb, 4 * a + 1));
            const float p2 = __int_as_float(__builtin_amdgcn_readlane(pb, 4 * a + 2));
            const float p3 = __int_as_float(__builtin_amdgcn_readlane(pb, 4 * a + 3));
            v = ((p0 + p1) + (p2 + p3)) + params[P.b3 + a];
        }
        o[a] = v;
    }
    if (wave) {
        if (u == 0) value_row[0] = o[0];
        return;
    }
    float logp, ent, act_c[AMAX];
    int action;
    policy_head_act<CONT, AMAX, true>(o, A, action_var, seed, rng_offset, 0, true, u, action_row, logp, ent,
                                      action, act_c);
    if (u == 0) {
        logprob_row[0] = logp;
        if constexpr (CONT) {
#pragma unroll
            for (int k = 0; k < AMAX; ++k) if (k < A) reinterpret_cast<float*>(host_action_out)[k] = act_c[k];
        } else {
            reinterpret_cast<int*>(host_action_out)[0] = action;
        }
        if (host_ticket) {
            __threadfence_system();
            __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// host_action_out and host_ticket are page-locked host buffers, which the device addresses
// directly: the observation travels as a kernel argument (the kernel files it in its buffer row),
// and the kernel writes the sampled action both to its buffer row and to the host buffer, then the
// ticket.  The host waits for the ticket word instead of for the stream (no interrupt, no
// wake-up): one launch per step, no copy commands.  With host_ticket == NULL the stream is waited for.
extern "C" int ppo_act_host(const float* params, int state_dim, int action_dim, int continuous,
                            float action_var, const float* host_state, float* state_row,
                            uint64_t seed, uint64_t rng_offset, void* action_row, float* logprob_row,
                            float* value_row, void* host_action_out, unsigned* host_ticket,
                            unsigned ticket, void* stream) {
    if (!params || !host_state || !state_row || !host_action_out || !action_row || !logprob_row || !value_row)
        return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 || action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (host_ticket && __atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket) return PPO_E_BADARG;
    OneState state;
    memcpy(state.v, host_state, sizeof(float) * state_dim);
#define ACT_ONE(C, AM)                                                                              \
    act_one_kernel<C, AM><<<1, 128, 0, st>>>(params, state_dim, action_dim, action_var, state,      \
                                             state_row, seed, rng_offset, action_row, logprob_row,  \
                                             value_row, host_action_out, host_ticket, ticket)
    if (continuous) {
        if (action_dim <= 4) ACT_ONE(true, 4); else ACT_ONE(true, 8);
    } else {
        if (action_dim <= 4) ACT_ONE(false, 4); else ACT_ONE(false, 8);
    }
#undef ACT_ONE
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return (int)le;
    if (!host_ticket) return (int)hipStreamSynchronize(st);
    // Spin on the ticket; every so often ask the stream, so a kernel that ended without writing the
    // ticket (a fault) surfaces as its error instead of a hang.
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket) return 0;
        if ((spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess)
                return __atomic_load_n(host_ticket, __ATOMIC_ACQUIRE) == ticket ? 0 : (int)hipErrorUnknown;
            if (q != hipErrorNotReady) return (int)q;
        }
    }
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
