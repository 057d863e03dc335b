// Pieces shared by the two fused forward/backward kernels (update.hip: forward/backward wave pairs on
// 32-sample tiles; update_sym.hip: symmetric waves on 16-sample tiles): the launch arguments, the
// per-sample head inputs, and the distribution head + loss + gradient w.r.t. the head's inputs.
#pragma once
#include "ppo_common.h"

#define FB_GRID 256   // at most one workgroup per CU, split between the actor and the critic

struct FwdBwdArgs {
    unsigned long long* stamps;
    const float* params;
    int S, A;
    float action_var;
    const float* states;
    const void* actions;
    const float* old_logp;
    const float* adv;
    const float* ret;
    int64_t n;
    float inv_n, clip_lo, clip_hi;
    float* out_logp;
    float* out_val;
    float* out_ent;
    float* out_surr;
    // Seeded form only (update_vjp.hip: the vector-Jacobian product of evaluate()): the caller's
    // gradients w.r.t. log-prob, value and entropy per sample, each may be null (= zeros); they
    // replace the PPO loss's own seeds.
    const float* seed_logp;
    const float* seed_val;
    const float* seed_ent;
    float* partial_actor;   // [n_actor_wg][stride_actor]
    float* partial_critic;  // [FB_GRID - n_actor_wg][stride_critic]
    int n_actor_wg;         // workgroups 0 .. n_actor_wg-1 train the actor, the rest the critic
    int n_critic_wg;        // (host side only: the grid is n_actor_wg + n_critic_wg workgroups)
    int* ws_header;         // [0] <- n_actor_wg, [1] <- critic workgroups: read back by the reduction kernels
    int stride_actor, stride_critic;
};

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

// ---------------------------------------------------------------------------- head + loss
// Distribution head, PPO.py:233-240 loss terms and the gradient w.r.t. the head's inputs for one
// sample per lane (both half-waves hold the same sample and compute the same values).
// Per-sample inputs of the head, fetched at the top of the tile loop so that their global-memory
// latency passes under the two MLP layers.
template <int HEAD, int AOUT>
struct HeadIn {
    float ret, old_lp, adv;
    int action;
    float act[HEAD == HEAD_GAUSSIAN ? AOUT : 1];
};

template <int HEAD, int AOUT>
__device__ __forceinline__ HeadIn<HEAD, AOUT> load_head_inputs(const FwdBwdArgs& a, int A, int64_t m,
                                                               bool valid) {
    HeadIn<HEAD, AOUT> in;
    in.ret = 0.f; in.old_lp = 0.f; in.adv = 0.f; in.action = 0;
    if constexpr (HEAD == HEAD_VALUE) {
#ifndef PPO_FWDBWD_SEEDED   // (the seeded form has no returns / old log-probs / advantages)
        in.ret = valid ? a.ret[m] : 0.f;
#endif
    } else {
#ifndef PPO_FWDBWD_SEEDED
        in.old_lp = valid ? a.old_logp[m] : 0.f;
        in.adv = valid ? a.adv[m] : 0.f;
#endif
        if constexpr (HEAD == HEAD_CATEGORICAL) {
            in.action = valid ? reinterpret_cast<const int*>(a.actions)[m] : 0;
        } else {
            const float* ap = reinterpret_cast<const float*>(a.actions);
#pragma unroll
            for (int k = 0; k < AOUT; ++k) in.act[k] = (valid && k < A) ? ap[m * A + k] : 0.f;
        }
    }
    return in;
}

template <int HEAD, int AOUT>
__device__ __forceinline__ void head_loss_grad(const FwdBwdArgs& a, int A, const float (&o)[AOUT],
                                               const HeadIn<HEAD, AOUT>& in,
                                               int64_t m, bool valid, bool writer,
                                               float (&dout)[AOUT], double& loss_acc) {
    if constexpr (HEAD == HEAD_VALUE) {
        // 0.5 * MseLoss(state_values, rewards), PPO.py:240; d/dV of its mean = (V - R)/n
        const float R = in.ret;
        const float diff = o[0] - R;
#ifdef PPO_FWDBWD_SEEDED
        dout[0] = (valid && a.seed_val) ? a.seed_val[m] : 0.f;
#else
        dout[0] = valid ? diff * a.inv_n : 0.f;
#endif
        if (valid && writer) {
            loss_acc += (double)diff * (double)diff;
            if (a.out_val) a.out_val[m] = o[0];
        }
    } else {
        const float old_lp = in.old_lp;
        const float adv = in.adv;
        float logp, ent;
        CatOut<AOUT> cat;
        GaussOut<AOUT> gs;
        int action = 0;
        if constexpr (HEAD == HEAD_CATEGORICAL) {
            action = in.action;
            categorical_head<AOUT>(o, A, cat);
            logp = cat.L[0];
#pragma unroll
            for (int k = 1; k < AOUT; ++k) logp = (k == action) ? cat.L[k] : logp;
            ent = cat.ent;
        } else {
            gaussian_head<AOUT>(o, A, a.action_var, in.act, gs);
            logp = gs.logp;
            ent = gs.ent;
        }
        const float ratio = head_exp(logp - old_lp);
        const float s1 = ratio * adv;
        const float s2 = fminf(fmaxf(ratio, a.clip_lo), a.clip_hi) * adv;
        const float surr = fminf(s1, s2);
        // torch.min splits ties 1/2 + 1/2 and clamp passes gradient on the closed interval: inside
        // the clip range the whole gradient flows through ratio; outside it only when the
        // unclipped branch is the smaller one.
        const bool inside = ratio >= a.clip_lo && ratio <= a.clip_hi;
        const bool gate = inside || (adv >= 0.f ? ratio <= a.clip_hi : ratio >= a.clip_lo);
#ifdef PPO_FWDBWD_SEEDED
        const float dlogp = (valid && a.seed_logp) ? a.seed_logp[m] : 0.f;
        const float dent = (valid && a.seed_ent) ? a.seed_ent[m] : 0.f;
        (void)gate;
#else
        const float dlogp = (valid && gate) ? -(adv * ratio) * a.inv_n : 0.f;
        const float dent = valid ? -0.01f * a.inv_n : 0.f;
#endif
        if constexpr (HEAD == HEAD_CATEGORICAL) {
            float pa = cat.pass[0], wq = 0.f;
#pragma unroll
            for (int k = 0; k < AOUT; ++k) {
                pa = (k == action) ? cat.pass[k] : pa;
                if (k < A) wq += (cat.L[k] + cat.pass[k]) * cat.q[k];
            }
#pragma unroll
            for (int k = 0; k < AOUT; ++k) {
                const float onehot = (k == action) ? 1.f : 0.f;
                const float dz_logp = pa * (onehot - cat.q[k]);
                const float dz_ent = -(cat.q[k] * ((cat.L[k] + cat.pass[k]) - wq));
                dout[k] = k < A ? dlogp * dz_logp + dent * dz_ent : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < AOUT; ++k)
                dout[k] = k < A ? dlogp * head_div(gs.zed[k], gs.sigma) * (1.0f - gs.mu[k] * gs.mu[k]) : 0.f;
        }
        if (valid && writer) {
            loss_acc += (double)(-surr - 0.01f * ent);
            if (a.out_logp) a.out_logp[m] = logp;
            if (a.out_ent) a.out_ent[m] = ent;
            if (a.out_surr) a.out_surr[m] = surr;
        }
    }
}


// update_sym.hip: the symmetric-wave form of the fused kernel (16-sample tiles), for state_dim <= 31
bool fwdbwd_sym_supported(int state_dim, int action_dim);
bool fwdbwd_sym_default(int state_dim, int action_dim, int64_t n);   // shapes on which it is the faster one
int launch_fwdbwd_sym(const FwdBwdArgs& a, int continuous, hipStream_t st);
