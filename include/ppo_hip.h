/*
 * ppo_hip.h - C ABI of the MI355X (gfx950) PPO engine.
 *
 * The reference (Katherine121/PPO, a fork of PPO-PyTorch) has no FFI: its hot path is the Python
 * class surface of PPO.py.  Each entry point below replaces the arithmetic of one piece of that
 * file; the citation after "replaces" names it.  The host-side mirror of the class surface
 * (ppo_amd/ppo.py) is the only caller.
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t, or a negative PPO_E_* code;
 *   - all pointers are DEVICE pointers borrowed from the caller (no ownership is taken) unless the
 *     parameter name ends in _host;
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*); nothing here
 *     synchronises, allocates or frees device memory;
 *   - thread-compatible, not thread-safe: one workspace per concurrent caller;
 *   - one process drives ONE device (the data-parallel design is one process per GPU): the library
 *     caches per-kernel launch attributes (dynamic LDS size) per process, not per device.
 *
 * Parameter vector layout (P floats, fp32, state_dict order of ActorCritic, PPO.py:49-73):
 *   actor.0.weight[64,S] actor.0.bias[64] actor.2.weight[64,64] actor.2.bias[64]
 *   actor.4.weight[A,64] actor.4.bias[A]  then the same six tensors of the critic with A -> 1.
 *   Every weight is [out,in] row-major, exactly as torch.nn.Linear stores it.
 */
#ifndef PPO_HIP_H
#define PPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_HIDDEN 64          /* hidden width of both MLPs, PPO.py:50-52,68-70 */
#define PPO_MAX_STATE_DIM 63   /* S + 1 (bias column) must fit two 32-wide MFMA tiles */
#define PPO_MAX_ACTION_DIM 8

#define PPO_E_BADARG (-1)      /* shape / pointer / size outside what the kernels assume */
#define PPO_E_WORKSPACE (-2)   /* workspace smaller than ppo_workspace_bytes() */
#define PPO_E_UNSUPPORTED (-3) /* S or A beyond PPO_MAX_* */
#define PPO_E_TOO_LARGE (-4)   /* size beyond the fixed capacity of an allocation-free entry point */

/* ABI version of this header (bumped on any signature change). */
int ppo_hip_abi_version(void);

/* Human-readable text for a return code of any function here. */
const char* ppo_hip_strerror(int code);

/* Number of parameters P for (S, A); n_actor_out receives the size of the actor block. */
int64_t ppo_param_count(int state_dim, int action_dim, int64_t* n_actor_out);

/* Bytes of scratch the functions taking (workspace, workspace_bytes) need for buffers of up to
 * max_samples samples. */
size_t ppo_workspace_bytes(int state_dim, int action_dim, int64_t max_samples);

/* ---- rollout -------------------------------------------------------------------------------
 * Batched ActorCritic.act for n states at once.  replaces: PPO.py:86-100 (act) as called by
 * PPO.select_action, PPO.py:175-198.  Samples an action per state from Categorical(softmax) or
 * N(tanh(mean), diag(action_var)), and writes action, log-prob of that action, and the critic
 * value.  Randomness: Philox4x32-10 keyed by `seed`, counter (state index, rng_offset < 2^56, block):
 * every (rng_offset, state index) pair owns a disjoint range of counter values, so the noise is
 * independent across states, steps and action dimensions; data-parallel ranks pass different seeds.
 * The reference's action stream (torch global CPU generator) is not reproducible and not a parity
 * target; log-prob and value for the (state, action) pairs written ARE.
 *   actions: int32[n] (discrete) or float[n, A] (continuous).
 *   action_var: fp32 value of std*std (PPO.py:46); ignored when continuous == 0.
 */
int ppo_act(const float* params, int state_dim, int action_dim, int continuous, float action_var,
            const float* states, int64_t n, uint64_t seed, uint64_t rng_offset,
            void* actions, float* logprobs, float* values, void* stream);

/* ActorCritic.evaluate forward only: log-prob of GIVEN actions, value, entropy per sample.
 * replaces: PPO.py:102-121.  Any of the three outputs may be NULL. */
int ppo_evaluate(const float* params, int state_dim, int action_dim, int continuous,
                 float action_var, const float* states, const void* actions, int64_t n,
                 float* logprobs, float* values, float* entropy, void* stream);

/* Vector-Jacobian product of ppo_evaluate w.r.t. the parameters: given dL/dlogprob, dL/dvalue and
 * dL/dentropy per sample (each float[n], or NULL for zeros), writes dL/dparams (flat, state_dict order)
 * into grad_params_out, which must hold param_count + 2 floats (the last two are scratch).
 * replaces: loss.backward() through policy.evaluate, PPO.py:227,244 (autograd over PPO.py:102-121), for
 * any loss a caller builds from evaluate()'s outputs; ppo_epoch_fwdbwd is this with PPO.py:233-240's
 * own seeds computed in the kernel.  The clamp of the Categorical's probabilities passes gradient on the
 * closed interval and the Gaussian's entropy has none, as in torch.  workspace: ppo_workspace_bytes(). */
int ppo_evaluate_backward(const float* params, int state_dim, int action_dim, int continuous,
                          float action_var, const float* states, const void* actions, int64_t n,
                          const float* grad_logprobs, const float* grad_values,
                          const float* grad_entropy, float* grad_params_out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- returns -------------------------------------------------------------------------------
 * Monte-Carlo discounted returns with terminal resets over ONE flat sequence, fp64 carry,
 * rounded once to fp32, plus the moments of the rounded returns.
 * replaces: PPO.py:202-211 (reverse loop + torch.tensor(..., float32)) and the reductions of
 * PPO.py:212.
 *   rewards: float[n] when rewards_are_f64 == 0, double[n] otherwise (the reference accumulates
 *   Python floats, i.e. doubles).  terminals: uint8[n].
 *   moments_out: double[3] = {count, mean, M2 = sum (x-mean)^2} of the fp32 returns.
 */
int ppo_returns_scan_flat(const void* rewards, int rewards_are_f64, const uint8_t* terminals,
                          int64_t n, double gamma, float* returns_out, double* moments_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ppo_returns_scan_flat without a workspace argument and without any allocation.  No counterpart in
 * the reference: it exists for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  The scan's scratch (block
 * aggregates, carries, block moments) is a static device array inside the library, sized for
 * n <= PPO_SCAN_NOALLOC_MAX (394 KB on every GPU the library runs on); larger n returns
 * PPO_E_TOO_LARGE.  Results are identical to ppo_returns_scan_flat's, bit for bit.  Every call on
 * one GPU uses that GPU's one array: calls that overlap in time on the same GPU (different
 * streams or threads) race; calls on one stream, or on different GPUs, do not.  Nothing has to
 * be cleared between calls.  `stream` must belong to the current device (PPO_E_BADARG otherwise):
 * the array used is the current device's. */
#define PPO_SCAN_NOALLOC_MAX (1 << 24)
int ppo_returns_scan_flat_noalloc(const void* rewards, int rewards_are_f64, const uint8_t* terminals,
                                  int64_t n, double gamma, float* returns_out, double* moments_out,
                                  void* stream);

/* Same recurrence for a time-major [n_steps, n_envs] rollout of independent envs: column e is
 * env e's trajectory; a column starts from carry 0 at its last step exactly as the flat buffer
 * does at its end (PPO.py:203).  Equals the flat scan of the env-major concatenation with the
 * last step of every env flagged terminal. */
int ppo_returns_scan_rows(const float* rewards_tn, const uint8_t* terminals_tn, int64_t n_steps,
                          int64_t n_envs, double gamma, float* returns_tn, double* moments_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Normalised returns and advantages from device-resident moments (no host round trip):
 *   std = sqrt(M2/(count-1)); returns_norm = (returns - mean) / (std + 1e-7);
 *   advantages = returns_norm - old_values.
 * replaces: PPO.py:212 (elementwise part) and PPO.py:221.  returns_norm may alias returns. */
int ppo_normalize_advantage(const float* returns, const float* old_values, int64_t n,
                            const double* moments, float* returns_norm, float* advantages,
                            void* stream);

/* ---- update --------------------------------------------------------------------------------
 * One epoch of evaluate + clipped-surrogate loss + backward over n_local samples, leaving
 * per-workgroup gradient partials in the workspace, preceded by a small header that records how the
 * workgroups were split between actor and critic (the reduction functions read it back, so they
 * have to be given the workspace of the preceding ppo_epoch_fwdbwd).  replaces: PPO.py:227-244 (evaluate, ratio,
 * surr1/surr2, loss, loss.mean().backward()).  Gradients are those of
 *   (1/n_global) * sum_t [ -min(r_t A_t, clamp(r_t,1-eps,1+eps) A_t) + 0.5 (V_t-R_t)^2 - 0.01 H_t ]
 * so that the sum over data-parallel shards is the reference's gradient.
 * Optional per-sample outputs (each may be NULL): out_logprobs, out_values, out_entropy,
 * out_surrogate (= min(surr1, surr2)).
 */
int ppo_epoch_fwdbwd(const float* params, int state_dim, int action_dim, int continuous,
                     float action_var, const float* states, const void* actions,
                     const float* old_logprobs, const float* advantages, const float* returns_norm,
                     int64_t n_local, int64_t n_global, float eps_clip,
                     float* out_logprobs, float* out_values, float* out_entropy,
                     float* out_surrogate, void* workspace, size_t workspace_bytes, void* stream);

/* Deterministic fixed-order sum of the partials left by ppo_epoch_fwdbwd.
 * grad_out: float[P + 2]; grad_out[P] = shard's sum of (-min(..) - 0.01 H)/n_global,
 * grad_out[P+1] = shard's 0.5*sum (V-R)^2 / n_global; their sum over shards is loss.mean(). */
int ppo_grad_reduce(int state_dim, int action_dim, const void* workspace, size_t workspace_bytes,
                    float* grad_out, void* stream);

/* torch.optim.Adam step (no weight decay, no amsgrad) with two learning-rate groups.
 * replaces: PPO.py:139-142,245 (torch/optim/adam.py single-tensor path).  `step` is the 1-based
 * count of optimizer steps including this one (it persists across update() calls).
 * loss_out (may be NULL): receives grad[n_params] + grad[n_params+1], i.e. loss.mean() of the
 * epoch whose (all-reduced) gradient vector this is; `grad` therefore holds n_params + 2 floats. */
int ppo_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n_params, int64_t n_actor, int64_t step, double lr_actor,
                  double lr_critic, double beta1, double beta2, double eps, float* loss_out,
                  void* stream);

/* Kernel timings of the K-epoch loop (no counterpart in the reference; what a profiler would show,
 * available to the caller without one).  After ppo_kernel_timings_enable(1), every
 * ppo_update_epochs call (k_epochs <= 256) brackets each of its launches with events on the launch
 * stream; ppo_kernel_timings waits for the last such call and writes out[0] = mean milliseconds of
 * the fused forward/backward launch (the replacement of PPO.py:227-244), out[1] = mean milliseconds
 * of the reduce + Adam launch (PPO.py:245), out[2] = launches of each kind timed (n >= 3).
 * ppo_kernel_timings_enable(0) switches the events off again. */
int ppo_kernel_timings_enable(int on);
int ppo_kernel_timings(float* out, int n);

/* ---- single-env latency path -------------------------------------------------------------------
 * PPO.select_action as the reference's loop calls it (PPO.py:175-198, train.py:176): ONE state on
 * the host in, ONE action on the host out.  host_state: state_dim floats of ordinary host memory
 * (copied into the launch's arguments).  host_action_out (int32, or action_dim floats) and
 * host_ticket (one word, may be NULL) must be PAGE-LOCKED host buffers (hipHostMalloc / torch
 * pin_memory), which the kernel writes directly.  The kernel files the observation in state_row,
 * computes exactly ppo_act's result for it (n = 1, same bits) and writes the
 * action to action_row and to host_action_out, then `ticket` to *host_ticket; the call returns when
 * it sees the ticket there (a spin on host memory: no stream wait).  `ticket` must differ from the
 * word's current value (count calls, skipping nothing: any change will do).  With host_ticket ==
 * NULL the stream is waited for instead.  One launch, no copy commands.  state_row / action_row /
 * logprob_row / value_row are the rollout buffer's device rows for this step; value_row is
 * complete in stream order (not necessarily when the call returns). */
int ppo_act_host(const float* params, int state_dim, int action_dim, int continuous, float action_var,
                 const float* host_state, float* state_row, uint64_t seed, uint64_t rng_offset,
                 void* action_row, float* logprob_row, float* value_row, void* host_action_out,
                 unsigned* host_ticket, unsigned ticket, void* stream);

/* The same step with launch, kernel start and weight load taken out of it.  A session keeps ONE kernel
 * waiting: it loads the weights, then polls a mailbox in page-locked host memory (owned by the
 * session) for requests and serves them one after the other; it ends by itself after about wait_us
 * microseconds without a request, or when the session is drained, and a step that finds no kernel
 * waiting launches one.  _step posts the request (state, rows, seed / rng_offset and action_var as in
 * ppo_act) and returns when the action (int32, or action_dim floats) is in action_out (ordinary host
 * memory); when it returns, all four rows of the step are visible to every stream.  The results are
 * ppo_act's, bit for bit.  The session's kernels run on a stream of its own; a kernel launched after
 * _create / _drain is ordered behind the work already enqueued on after_stream.
 * _drain withdraws the waiting kernel and waits for the session's stream: call it before the
 * weights behind `params` change, before the rows' storage moves, and before exit (_destroy does).
 * The waiting kernel holds the weights it loaded at its launch and carries the buffer columns, the seed
 * and the variance in its arguments: params_version is the caller's count of writes to the vector
 * behind `params` (any number that changes whenever the contents do); a step whose params,
 * params_version, columns, seed or variance differ from those the waiting kernel was launched under
 * withdraws that kernel and is served by one launched now.  _step addresses rows by pointer: a call
 * whose four rows are row r (0 <= r < 2^32) of the columns of the binding in force is served as that
 * binding's step r, so a caller that walks down its buffer keeps its waiting kernel; any other call
 * binds anew with its rows as row 0.
 * One session serves one caller thread.  No counterpart in the reference: PPO.py:175-198 pays a
 * full launch per step. */
typedef struct ppo_act_session ppo_act_session;
int ppo_act_session_create(ppo_act_session** out, int state_dim, int action_dim, int continuous,
                           int wait_us);
int ppo_act_session_step(ppo_act_session* session, const float* params, uint64_t params_version,
                         float action_var, const float* host_state, uint64_t seed, uint64_t rng_offset, float* state_row,
                         void* action_row, float* logprob_row, float* value_row, void* action_out,
                         void* after_stream);
/* The same step with its unchanging arguments bound beforehand (a host loop's per-step cost is mostly
 * argument marshalling in the binding).  _bind stores the weights, their version, the variance, the
 * host state / action buffers, the seed, the four buffer columns (dense rows: state_dim floats; one
 * int32 or action_dim floats; one float; one float) and the stream a newly launched kernel is
 * ordered behind; _step_bound(row, rng_offset), 0 <= row < 2^32, is ppo_act_session_step on row `row`
 * of those columns.  Bind again whenever one of the bound things changes (the next step then replaces
 * the waiting kernel). */
int ppo_act_session_bind(ppo_act_session* session, const float* params, uint64_t params_version,
                         float action_var, const float* host_state, void* action_out, uint64_t seed,
                         float* states, void* actions, float* logprobs, float* values,
                         void* after_stream);
int ppo_act_session_step_bound(ppo_act_session* session, int64_t row, uint64_t rng_offset);
int ppo_act_session_drain(ppo_act_session* session);
int ppo_act_session_destroy(ppo_act_session* session);

/* ---- synthetic vectorised environment (measurement harness) ---------------------------------
 * One lockstep step of n_envs synthetic envs: observations ~ N(0,1), reward 1.0 (kind 0) or
 * ~N(0,1) (kind 1), terminal ~ Bernoulli(p_terminal) or, when p_terminal < 0, exactly at steps
 * with (step % horizon) == horizon-1.  No counterpart in the reference (its envs are gym's,
 * train.py:53); this stands in for env.step (train.py:177) in bench.py and the tests. */
int ppo_synth_env_step(float* states_out, float* rewards_out, uint8_t* terminals_out,
                       int64_t n_envs, int state_dim, int reward_kind, float p_terminal,
                       int64_t step, int64_t horizon, uint64_t seed, void* stream);

/* A whole lockstep rollout of the synthetic envs in ONE launch: a persistent kernel in which a
 * workgroup owns a tile of envs for all n_steps steps (one wave the actor, one the critic, one the
 * env's draws).  The buffers it leaves are bit-identical to n_steps rounds of ppo_synth_env_step into
 * rows [t*n_envs, (t+1)*n_envs) of the time-major buffers followed by ppo_act on those rows
 * (rng_offset = act_offset0 + t).  Stands in for the loop of train.py:173-181 in the measurement
 * harness. */
int ppo_synth_rollout(const float* params, int state_dim, int action_dim, int continuous,
                      float action_var, int64_t n_envs, int64_t n_steps, int reward_kind,
                      float p_terminal, int64_t step0, int64_t horizon, uint64_t env_seed,
                      uint64_t act_seed, uint64_t act_offset0, float* states, float* rewards,
                      uint8_t* terminals, void* actions, float* logprobs, float* values,
                      void* stream);

/* ---- device-side vectorised CartPole-v1 ---------------------------------------------------------
 * The env on the other side of select_action: env.reset() (train.py:170) and env.step(action)
 * (train.py:177) of gym's CartPole-v1 (train.py:19,136) for n_envs independent envs in lockstep.
 * state [n,4] fp64, steps [n] int32, episodes [n] uint32 are the envs' own storage; fin_sum [n] f32
 * and fin_cnt [n] int32 (both may be NULL) accumulate the returns / number of finished episodes per
 * env for logging.  ppo_cartpole_step integrates one step, writes reward (1.0) and terminal flag
 * (pole fell, cart left the track, or max_episode_steps reached), resets finished envs in the
 * same call and writes the observation the next select_action has to see. */
int ppo_cartpole_reset(double* state, int* steps, unsigned* episodes, float* fin_sum, int* fin_cnt,
                       float* obs_out, int64_t n_envs, uint64_t seed, void* stream);
int ppo_cartpole_step(double* state, int* steps, unsigned* episodes, float* fin_sum, int* fin_cnt,
                      const int* actions, float* obs_out, float* rewards_out,
                      uint8_t* terminals_out, int64_t n_envs, int max_episode_steps, uint64_t seed,
                      void* stream);
/* env.step(action) of ONE env (n_envs = 1) as the reference's loop calls it (train.py:177): the action
 * (0 or 1) rides in the launch's arguments; host_out (6 floats: observation, reward, terminal flag as
 * 0.0 / 1.0) and host_ticket (one word) are PAGE-LOCKED host buffers the kernel writes directly; the
 * call returns when `ticket` (any value other than the word's current one) is in *host_ticket.  One
 * launch, no copy commands, no stream wait; the same step as ppo_cartpole_step, bit for bit. */
int ppo_cartpole_step_host(double* state, int* steps, unsigned* episodes, float* fin_sum, int* fin_cnt,
                           int action, float* host_out, unsigned* host_ticket, unsigned ticket,
                           int max_episode_steps, uint64_t seed, void* stream);

/* n_steps lockstep steps of the policy on the device-side CartPole envs, enqueued by one call: row t
 * of the time-major buffers (states [n_steps*n_envs, 4], actions int32, logprobs, values, rewards,
 * terminals) receives what ppo_act and ppo_cartpole_step produce at step t; obs_carry [n_envs, 4]
 * holds the observation the policy sees first (in) and the one after the last step (out).
 * The loop of train.py:173-183 for n_envs envs at once. */
int ppo_cartpole_rollout(const float* params, double* state, int* steps, unsigned* episodes,
                         float* fin_sum, int* fin_cnt, float* obs_carry, int64_t n_envs,
                         int64_t n_steps, int max_episode_steps, uint64_t env_seed, uint64_t act_seed,
                         uint64_t act_offset0, float* states, float* rewards, uint8_t* terminals,
                         int* actions, float* logprobs, float* values, void* stream);

/* Fused ppo_grad_reduce + ppo_adam_step for the single-process case (no all-reduce in between):
 * same arithmetic, one launch.  grad_out (P + 2 floats) still receives the reduced gradient. */
int ppo_reduce_adam_step(int state_dim, int action_dim, const void* workspace, size_t workspace_bytes,
                         float* grad_out, float* params, float* exp_avg, float* exp_avg_sq,
                         int64_t step, double lr_actor, double lr_critic, double beta1, double beta2,
                         double eps, float* loss_out, void* stream);

/* The whole K-epoch loop of PPO.update (PPO.py:224-245) for a single process, enqueued by one
 * call: k_epochs x (ppo_epoch_fwdbwd with n_global = n, ppo_reduce_adam_step with step
 * first_step + e).  losses_out (may be NULL) receives k_epochs floats.  For small buffers (the
 * reference's own 900-6000-step updates) the per-epoch cost is launch latency, which this halves. */
int ppo_update_epochs(float* params, int state_dim, int action_dim, int continuous, float action_var,
                      const float* states, const void* actions, const float* old_logprobs,
                      const float* advantages, const float* returns_norm, int64_t n, float eps_clip,
                      int k_epochs, float* grad_out, float* exp_avg, float* exp_avg_sq,
                      int64_t first_step, double lr_actor, double lr_critic, double beta1,
                      double beta2, double eps, float* losses_out, void* workspace,
                      size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPO_HIP_H */
