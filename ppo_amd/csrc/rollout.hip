// Whole lockstep rollouts in ONE launch: the loop of train.py:173-181 (select_action, env.step,
// reward / terminal appends) for n_envs envs and n_steps steps.
//
// A workgroup owns a tile of TM envs for all steps: wave 0 runs the actor (forward, head, sampling,
// log-prob), wave 1 the critic, wave 2 (synthetic env only) the env; both MLP images are staged into
// LDS once.  TM = 16 (the 16x16x4 MFMA
// tile: half the matrix and tanh work in each step's dependent chain, twice the waves) by default,
// TM = 32 on request; the two produce the same bits.  The envs live inside the
// kernel: the synthetic env's draws depend on (seed, step, env) only, so the env wave produces the
// next step's observations while the other two work on this one; the CartPole env integrates in the actor
// wave's lanes, state in registers, right after the action is drawn.  Observations pass between the
// waves through a double-buffered LDS tile (one workgroup barrier per step) and are written to the
// time-major rollout buffer as they are produced.  Every arithmetic piece (MLP tile, heads, Philox
// streams, env dynamics) is the device function the per-step kernels use, so the buffers written are
// bit-identical to those of n_steps rounds of ppo_synth_env_step / ppo_cartpole_step + ppo_act.
#include "ppo_common.h"
#include <stdlib.h>

#define ENV_SYNTH 0
#define ENV_CARTPOLE 1

struct RolloutArgs {
    const float* params;
    int S, A;
    float action_var;
    int64_t n_envs;
    int64_t n_steps;
    float* states;        // [n_steps * n_envs, S]
    float* rewards;       // [n_steps * n_envs]
    uint8_t* terminals;
    void* actions;        // int32 [..] or float [.., A]
    float* logp;
    float* values;
    uint64_t act_seed, act_offset0;
    // synthetic env
    int reward_kind;
    float p_terminal;
    int64_t step0, horizon;
    uint64_t env_seed;
    // CartPole env (its own storage, see ppo_cartpole_step)
    double* cp_state;
    int* cp_steps;
    unsigned* cp_episodes;
    float* fin_sum;
    int* fin_cnt;
    float* obs_carry;     // [n_envs, 4]: observation before the first step (in) / after the last (out)
    int max_steps;
};

__host__ __device__ inline int stage_ld(int S) { return S | 1; }  // odd row stride: conflict-free column walks

template <int ENV, bool CONT, int AMAX, int TM>
__global__ __launch_bounds__(ENV == ENV_SYNTH ? 192 : 128) void rollout_kernel(const RolloutArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int S = a.S, A = a.A;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // c: this lane's env in the tile; h: which of the 64 / TM lanes that hold the same env it is
    constexpr int COPIES = 64 / TM;
    const int lane = threadIdx.x & 63, h = lane / TM, c = lane % TM;
    const int64_t m = (int64_t)blockIdx.x * TM + c;   // this lane's env
    const bool valid = m < a.n_envs;
    const int64_t N = a.n_envs;

    const NetLayout PA = net_layout(0, S, A);
    const NetLayout PC = net_layout(PA.n, S, 1);
    const int actor_floats = TileNet<TM>::floats(S, AMAX);
    const int critic_floats = TileNet<TM>::floats(S, 1);
    TileNet<TM> NA, NC;
    const int SLD = stage_ld(S);
    float* const stage = smem + actor_floats + critic_floats;      // [2][TM][SLD]
    NA.stage(smem, a.params, PA, S, A, AMAX);
    NC.stage(smem + actor_floats, a.params, PC, S, 1, 1);

    // ---- observation of step 0 -> stage[0] (+ the buffer's row 0)
    double cp[4] = {0.0, 0.0, 0.0, 0.0};
    int cp_steps = 0;
    unsigned cp_ep = 0;
    float cp_fin_sum = 0.f;
    int cp_fin_cnt = 0;
    const bool env_lane = wave == 0 && h == 0 && valid;   // CartPole: the lane that owns env m
    if constexpr (ENV == ENV_CARTPOLE) {
        if (env_lane) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cp[k] = a.cp_state[m * 4 + k];
                const float o = a.obs_carry[m * 4 + k];
                stage[c * SLD + k] = o;
                a.states[m * 4 + k] = o;
            }
            cp_steps = a.cp_steps[m];
            cp_ep = a.cp_episodes[m];
            if (a.fin_sum) cp_fin_sum = a.fin_sum[m];
            if (a.fin_cnt) cp_fin_cnt = a.fin_cnt[m];
        }
    }
    // synthetic env: draws of step t for this lane's env, by a wave of its own (the draws depend on
    // (seed, step, env) only, so step t + 1 is drawn while the other two waves work on step t).  Block 0
    // (reward, terminal) and the observation blocks 1.. are dealt round-robin to the lanes that hold
    // the env.
    auto synth_generate = [&](int64_t t) {
        if (!valid) return;
        const int64_t row = t * N + m;
        float* const srow = stage + ((t & 1) * TM + c) * SLD;
        const int nblk = 1 + ((S + 3) >> 2);
        for (int b = h; b < nblk; b += COPIES) {
            float v[4];
            uint32_t w2;
            synth_env_block(a.env_seed, a.step0 + t, m, b, v, w2);
            if (b == 0) {
                float reward;
                bool term;
                synth_reward_terminal_of(v, w2, a.step0 + t, a.reward_kind, a.p_terminal, a.horizon, reward, term);
                a.rewards[row] = reward;
                a.terminals[row] = term ? 1 : 0;
            } else {
                const int j = b - 1;
                for (int i = 0; i < 4 && 4 * j + i < S; ++i) {
                    srow[4 * j + i] = v[i];
                    a.states[row * S + 4 * j + i] = v[i];
                }
            }
        }
    };
    if constexpr (ENV == ENV_SYNTH) {
        if (wave == 2) synth_generate(0);
    }

    for (int64_t t = 0; t < a.n_steps; ++t) {
        __syncthreads();   // stage[t & 1] holds the observations of step t (and, first time, the images)
        const int64_t row = t * N + m;
        const RowFromLds x{stage + ((t & 1) * TM + c) * SLD, valid};
        if (wave == 2) {
            // ---------------- synthetic env: next step's draws
            if (t + 1 < a.n_steps) synth_generate(t + 1);
        } else if (wave == 1) {
            // ---------------- critic: value of the observation
            float v[1];
            NC.template forward<1>(S, 1, x, lane, v);
            if (valid && h == 0) a.values[row] = v[0];
        } else {
            // ---------------- actor: sample the action, then (CartPole) step the env with it
            float o[AMAX];
            NA.template forward<AMAX>(S, A, x, lane, o);
            float logp, ent, act_c[AMAX];
            int action;
            void* arow = CONT ? (void*)((float*)a.actions + (size_t)t * N * A) : (void*)((int*)a.actions + (size_t)t * N);
            policy_head_act<CONT, AMAX, true>(o, A, a.action_var, a.act_seed, a.act_offset0 + t, m, valid, h,
                                              arow, logp, ent, action, act_c);
            if (valid && h == 0) a.logp[row] = logp;
            if constexpr (ENV == ENV_CARTPOLE) {
                if (env_lane) {
                    const bool fell = cartpole_advance(cp, action);
                    const int steps = cp_steps + 1;
                    const bool done = fell || steps >= a.max_steps;
                    a.rewards[row] = 1.0f;
                    a.terminals[row] = done ? 1 : 0;
                    if (done) {
                        cp_fin_sum += (float)steps;
                        cp_fin_cnt += 1;
                        cartpole_reset_draw(a.env_seed, cp_ep, m, cp);
                        cp_steps = 0;
                        cp_ep += 1;
                    } else {
                        cp_steps = steps;
                    }
                    // observation the next select_action sees: next row, or the carry after the last step
                    float* const srow = stage + (((t + 1) & 1) * TM + c) * SLD;
                    float* const grow = t + 1 < a.n_steps ? a.states + (row + N) * 4 : a.obs_carry + m * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float ob = (float)cp[k];
                        srow[k] = ob;
                        grow[k] = ob;
                    }
                }
            }
        }
    }
    if constexpr (ENV == ENV_CARTPOLE) {
        if (env_lane) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.cp_state[m * 4 + k] = cp[k];
            a.cp_steps[m] = cp_steps;
            a.cp_episodes[m] = cp_ep;
            if (a.fin_sum) a.fin_sum[m] = cp_fin_sum;
            if (a.fin_cnt) a.fin_cnt[m] = cp_fin_cnt;
        }
    }
}

template <int ENV, bool CONT, int AMAX, int TM>
static int launch_rollout_tm(const RolloutArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(TileNet<TM>::floats(a.S, AMAX) + TileNet<TM>::floats(a.S, 1) +
                                2 * TM * stage_ld(a.S)) * sizeof(float);
    if (lds > 160 * 1024) return PPO_E_UNSUPPORTED;
    static size_t configured = 0;   // per process: one process drives one device
    if (lds > configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)rollout_kernel<ENV, CONT, AMAX, TM>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    const int64_t tiles = (a.n_envs + TM - 1) / TM;
    if (tiles > 0x7fffffff) return PPO_E_BADARG;
    rollout_kernel<ENV, CONT, AMAX, TM><<<(int)tiles, ENV == ENV_SYNTH ? 192 : 128, lds, st>>>(a);
    return (int)hipGetLastError();
}

// 16-env tiles by default: measured faster than 32-env tiles both where the step chain is latency-bound
// (4096 envs: 0.38 against 0.56 ms for 128 steps) and where the chip is full (65536 envs: 5.1 against
// 5.4 ms).  PPO_ROLLOUT_TILE=32 selects the other shape (same bits).
static int rollout_tile(int64_t) {
    static const int forced = [] {
        const char* e = getenv("PPO_ROLLOUT_TILE");
        return e ? atoi(e) : 0;
    }();
    return forced == 32 ? 32 : 16;
}

template <int ENV, bool CONT, int AMAX>
static int launch_rollout(const RolloutArgs& a, hipStream_t st) {
    if (rollout_tile(a.n_envs) == 16) return launch_rollout_tm<ENV, CONT, AMAX, 16>(a, st);
    return launch_rollout_tm<ENV, CONT, AMAX, 32>(a, st);
}

extern "C" int ppo_synth_rollout(const float* params, int state_dim, int action_dim, int continuous,
                                 float action_var, int64_t n_envs, int64_t n_steps, int reward_kind,
                                 float p_terminal, int64_t step0, int64_t horizon, uint64_t env_seed,
                                 uint64_t act_seed, uint64_t act_offset0, float* states, float* rewards,
                                 uint8_t* terminals, void* actions, float* logprobs, float* values,
                                 void* stream) {
    if (!params || !states || !rewards || !terminals || !actions || !logprobs || !values ||
        n_steps <= 0 || n_envs <= 0 || horizon < 1)
        return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 || action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    RolloutArgs a{};
    a.params = params; a.S = state_dim; a.A = action_dim; a.action_var = action_var;
    a.n_envs = n_envs; a.n_steps = n_steps;
    a.states = states; a.rewards = rewards; a.terminals = terminals; a.actions = actions;
    a.logp = logprobs; a.values = values;
    a.act_seed = act_seed; a.act_offset0 = act_offset0;
    a.reward_kind = reward_kind; a.p_terminal = p_terminal; a.step0 = step0; a.horizon = horizon;
    a.env_seed = env_seed;
    hipStream_t st = (hipStream_t)stream;
    const int amax = action_dim <= 4 ? 4 : 8;   // the head widths ppo_act instantiates
    if (continuous) {
        if (amax == 4) return launch_rollout<ENV_SYNTH, true, 4>(a, st);
        return launch_rollout<ENV_SYNTH, true, 8>(a, st);
    }
    if (amax == 4) return launch_rollout<ENV_SYNTH, false, 4>(a, st);
    return launch_rollout<ENV_SYNTH, false, 8>(a, st);
}

extern "C" int ppo_cartpole_rollout(const float* params, double* state, int* steps, unsigned* episodes,
                                    float* fin_sum, int* fin_cnt, float* obs_carry, int64_t n_envs,
                                    int64_t n_steps, int max_episode_steps, uint64_t env_seed,
                                    uint64_t act_seed, uint64_t act_offset0, float* states,
                                    float* rewards, uint8_t* terminals, int* actions, float* logprobs,
                                    float* values, void* stream) {
    if (!params || !state || !steps || !episodes || !obs_carry || !states || !rewards || !terminals ||
        !actions || !logprobs || !values || n_envs <= 0 || n_steps <= 0 || max_episode_steps < 1)
        return PPO_E_BADARG;
    RolloutArgs a{};
    a.params = params; a.S = 4; a.A = 2; a.action_var = 0.f;
    a.n_envs = n_envs; a.n_steps = n_steps;
    a.states = states; a.rewards = rewards; a.terminals = terminals; a.actions = actions;
    a.logp = logprobs; a.values = values;
    a.act_seed = act_seed; a.act_offset0 = act_offset0;
    a.env_seed = env_seed;
    a.cp_state = state; a.cp_steps = steps; a.cp_episodes = episodes; a.fin_sum = fin_sum;
    a.fin_cnt = fin_cnt; a.obs_carry = obs_carry; a.max_steps = max_episode_steps;
    return launch_rollout<ENV_CARTPOLE, false, 4>(a, (hipStream_t)stream);
}
