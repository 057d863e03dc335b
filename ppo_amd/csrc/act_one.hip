// select_action as train.py:176 calls it: ONE state per call, the action back on the host before the
// env can step.  What counts here is the time from "state known" to "action known", not throughput.
//
// Kernel.  A tile kernel would stage 17 KB of weights into LDS to use one column of a 32-column
// MFMA; here a wave per network holds one hidden unit per lane and reads its weight rows straight
// from global memory.  The fp32 MFMA adds its k-slices as a chain of fmaf (one rounding each, lower
// lane group first), so the same chain on the VALU, in the tile's unit order, gives the tile
// kernels' result bit for bit:
//   layer 1: inputs ascending, then the bias;  layer 2: bias first, then the units in the order
//   0,4,1,5,2,6,3,7 | 8,12,...;  out layer: the four partial sums of out_layer_act.
// The activation a lane needs from another lane comes by v_readlane (layer 2) or through 64 floats
// of LDS (out layer).  Wave 0 = actor (+ head, sampling, the host copy of the action and the
// completion notice), wave 1 = critic.
//
// Two ways to hand it a state:
//  * ppo_act_host: the state travels in the launch's arguments; the host spins on a ticket word in
//    page-locked memory instead of waiting for the stream.
//  * ppo_act_session_*: the launch itself leaves the critical path.  A kernel is launched BEFORE
//    its state exists: it loads its weights, then polls a mailbox in page-locked host memory for
//    the request, serves it and goes back to polling, its weight rows still in registers.  It always
//    terminates on its own: when no request has come for about a millisecond (configurable), or when a
//    control word of the mailbox withdraws it; a step that finds no kernel waiting launches one.
//    Request numbers are 1 .. 2^31 - 1.  Kernels of one session run on the session's own stream, one
//    after the other, which is what lets them pass "last request served" through a device word.
//    A read of host memory is a PCIe round trip of more than a microsecond, and a store that has to
//    be acknowledged before the next one may go costs as much again, so the protocol is built to need
//    ONE read round trip and NO acknowledged store per step: request and reply travel as 16-bit
//    halves, each in a dword of its own whose upper half is the low 16 bits of the request number.
//    A dword is read and written whole, so every dword says by itself which request it belongs to:
//    the poll IS the fetch of the request (a poll that finds every dword on the same new number has
//    the whole request, whatever order the host's stores and the device's loads were performed in),
//    and the host accepts a reply once every dword of it carries the number it posted.  What stays
//    the same from step to step (buffer columns, seed, variance) travels in the launch's arguments.
#include "ppo_common.h"
#include <stddef.h>
#include <string.h>

struct ActRequest {   // one step's rows and random stream (immediate form: in the launch's arguments)
    float* state_row;
    void* action_row;
    float* logprob_row;
    float* value_row;
    uint64_t seed, rng_offset;
    float action_var;
    unsigned seq;
    unsigned pad[2];
};

// Tagged halves: dword = (request number & 0xffff) << 16 | 16 bits of payload.
// Request payload, in halves: [0..1] row (32 bits), [2..5] rng_offset (64 bits), [6 + 2u, 7 + 2u] input u.
#define ACT_REQ_HALVES(S) (6 + 2 * (S))
#define ACT_REQ_MAX 192                // >= ACT_REQ_HALVES(PPO_MAX_STATE_DIM), three dwords per lane at most
static_assert(ACT_REQ_HALVES(PPO_MAX_STATE_DIM) <= ACT_REQ_MAX, "request does not fit three dwords per lane");

struct ActMailbox {   // page-locked host memory
    unsigned req[ACT_REQ_MAX];      // host -> device: the request as tagged halves
    unsigned ctrl;                  // host -> device: ACT_CANCEL_BIT withdraws the waiter
    unsigned pad0[15];
    uint64_t reply[8];              // device -> host: the action words as tagged halves (int32 in word 0 for
                                    //   a discrete policy), two dwords per word (one 8-byte store each)
    unsigned unserved;              // device -> host: kernels that gave up waiting so far (only grows)
    unsigned pad1[15];
};

struct ActBound {   // what the waiting kernel is launched with: the buffer's columns (dense rows) and the stream of draws
    float* states;
    char* actions;
    float* logprobs;
    float* values;
    uint64_t seed;
    float action_var;
};

struct OneState {   // the observation, passed by value: it reaches the device with the launch itself
    float v[PPO_MAX_STATE_DIM + 1];
};

#define ACT_CANCEL_BIT 0x80000000u
#define ACT_MAX_POLLS (1u << 16)

template <class T>
__device__ __forceinline__ T host_load(const T* p) {   // a load that always goes to the host's memory
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// mb == nullptr: serve (req, state) from the arguments now, report through host_action_out / host_ticket.
// mb != nullptr: wait until the mailbox holds a request with a number other than words[0] (words: device
// memory, [0] = last request served, [1] = kernels that gave up), serve it on row `row` of the bound
// columns, report through the mailbox.
template <bool CONT, int AMAX, bool WAIT>
__device__ __forceinline__ void act_one_body(const float* __restrict__ params, int S, int A,
                                             ActRequest req, const OneState* state,
                                             void* host_action_out, unsigned* host_ticket,
                                             ActMailbox* mb_arg, unsigned* words, unsigned wait_ticks,
                                             const ActBound* bound) {
    // every multiply and add below is rounded as written (a prescale product must be rounded before
    // it enters the chain, as the tile kernels' staged images are)
#pragma clang fp contract(off)
    ActMailbox* const mb = WAIT ? mb_arg : nullptr;
    __shared__ float h2s[2][HID];
    __shared__ float xs[HID];
    __shared__ unsigned go, go_row;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), u = threadIdx.x & 63;
    const int aout = wave ? 1 : A;
    const NetLayout PA = net_layout(0, S, A);
    const NetLayout P = wave ? net_layout(PA.n, S, 1) : PA;

    // weight rows of this lane's unit: issued up front (in the waiting form: before the state exists)
    f32x4 w2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w2[i] = *reinterpret_cast<const f32x4*>(params + P.w2 + u * HID + 4 * i);
    const float b1 = params[P.b1 + u], b2 = params[P.b2 + u];
    float w1a[8];   // the first eight inputs' weights (all of them for most envs)
#pragma unroll
    for (int j = 0; j < 8; ++j) w1a[j] = j < S ? params[P.w1 + u * S + j] : 0.f;
    float w1b[24];  // inputs 8 .. 31 (the widest observations of the reference's envs are 24 - 28 values)
    if (S > 8) {
#pragma unroll
        for (int j = 0; j < 24; ++j) w1b[j] = 8 + j < S ? params[P.w1 + u * S + 8 + j] : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 24; ++j) w1b[j] = 0.f;
    }
    // out layer: lane 4 a + g sums class g of output a (16 units, ascending)
    const int oa = u >> 2, og = u & 3;
    const bool out_lane = oa < aout;
    float w3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w3[i] = out_lane ? params[P.w3 + oa * HID + u16(og, i >> 2, i & 3)] : 0.f;
    float b3[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) b3[a] = a < aout ? params[P.b3 + a] : 0.f;

    unsigned served = 0;   // (waiting form) the last request served, by this kernel or the one before it
    if constexpr (WAIT) served = words[0];
    for (;;) {
    unsigned expect = 0;
    float xv = 0.f;
    if constexpr (WAIT) {
        if (wave == 0) {
            // The whole wave polls: lane i fetches dwords i, i + 64 and i + 128 of the request, lane 0 the
            // control word too, all in flight together: one round trip to the host's memory per poll.
            const int nd = ACT_REQ_HALVES(S);
            const bool has0 = u < nd, has1 = u + 64 < nd, has2 = u + 128 < nd;
            unsigned g = 0, w0 = 0, w1 = 0, w2 = 0;
            const uint64_t t0 = wall_clock64();
            for (unsigned it = 0; it < ACT_MAX_POLLS; ++it) {
                if (has0) w0 = host_load(&mb->req[u]);
                if (has1) w1 = host_load(&mb->req[u + 64]);
                if (has2) w2 = host_load(&mb->req[u + 128]);
                const unsigned ctl = u == 0 ? host_load(&mb->ctrl) : 0u;
                if (__builtin_amdgcn_readfirstlane(ctl) & ACT_CANCEL_BIT) break;
                const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane(w0) >> 16;
                const bool torn = (has0 && (w0 >> 16) != t) || (has1 && (w1 >> 16) != t) || (has2 && (w2 >> 16) != t);
                const unsigned ahead = (t - served) & 0xffffu;   // requests are numbered consecutively
                if (ahead != 0 && !__any(torn)) { g = served + ahead; break; }
                if ((it & 7) == 7 && wall_clock64() - t0 > wait_ticks) break;
            }
            if (g) {
                // lane u's input: halves 6 + 2u and 7 + 2u
                const int i0 = 6 + 2 * u, i1 = i0 + 1;
                unsigned lo = __shfl(w0, i0 & 63), hi = __shfl(w0, i1 & 63);
                if (nd > 64) {   // (wave-uniform)
                    const unsigned lo1 = __shfl(w1, i0 & 63), hi1 = __shfl(w1, i1 & 63);
                    const unsigned lo2 = __shfl(w2, i0 & 63), hi2 = __shfl(w2, i1 & 63);
                    lo = (i0 >> 6) == 0 ? lo : ((i0 >> 6) == 1 ? lo1 : lo2);
                    hi = (i1 >> 6) == 0 ? hi : ((i1 >> 6) == 1 ? hi1 : hi2);
                }
                if (u < S) xs[u] = __uint_as_float((lo & 0xffffu) | (hi << 16));
                auto half = [&](int i) { return (uint64_t)((unsigned)__builtin_amdgcn_readlane(w0, i) & 0xffffu); };
                req.rng_offset = half(2) | half(3) << 16 | half(4) << 32 | half(5) << 48;
                if (u == 0) go_row = (unsigned)(half(0) | half(1) << 16);
            }
            if (u == 0) go = g;
        }
        __syncthreads();
        expect = go;
        if (!expect) {   // nobody came within the waiting time, or the waiter was withdrawn: say so and leave
            if (threadIdx.x == 0) {
                const unsigned n = words[1] + 1;
                words[1] = n;
                __hip_atomic_store(&mb->unserved, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (u < S) xv = xs[u];
        const size_t row = go_row;
        req.state_row = bound->states + row * S;
        req.action_row = bound->actions + row * (sizeof(float) * (CONT ? (size_t)A : 1));
        req.logprob_row = bound->logprobs + row;
        req.value_row = bound->values + row;
        req.seed = bound->seed;
        req.action_var = bound->action_var;
    } else {
        if (u < S) xv = state->v[u];
    }
    if (wave == 0 && u < S) req.state_row[u] = xv;
    const int xb = __float_as_int(xv);

    float z = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)   // a zero weight (input >= S) leaves z as it is
        z = fmaf(w1a[j] * TANH_PRESCALE, __int_as_float(__builtin_amdgcn_readlane(xb, j)), z);
    if (S > 8) {
#pragma unroll
        for (int j = 0; j < 24; ++j)
            z = fmaf(w1b[j] * TANH_PRESCALE, __int_as_float(__builtin_amdgcn_readlane(xb, 8 + j)), z);
    }
    for (int k0 = 32; k0 < S; k0 += 8) {   // eight loads in flight
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = k0 + j < S ? params[P.w1 + u * S + k0 + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            z = fmaf(w[j] * TANH_PRESCALE, __int_as_float(__builtin_amdgcn_readlane(xb, (k0 + j) & 63)), z);
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
            v = ((p0 + p1) + (p2 + p3)) + b3[a];
        }
        o[a] = v;
    }
    float logp = 0.f, ent, act_c[AMAX];
    int action = 0;
    if (wave) {
        if (u == 0) req.value_row[0] = o[0];
    } else {
        policy_head_act<CONT, AMAX, true>(o, A, req.action_var, req.seed, req.rng_offset, 0, true, u,
                                          req.action_row, logp, ent, action, act_c);
        if (u == 0) req.logprob_row[0] = logp;
    }
    __syncthreads();   // the critic's row is written too: the reply below vouches for the whole step
    if (threadIdx.x == 0) {
        if constexpr (WAIT) {
            words[0] = expect;
            // the four rows (device memory: acknowledged within the device) become visible to every
            // stream before the reply leaves; the reply's own stores wait for nothing
            __threadfence_system();
            const unsigned tag = expect << 16;
            auto put = [&](int k, unsigned word) {
                const uint64_t v = (uint64_t)(tag | (word & 0xffffu)) | (uint64_t)(tag | (word >> 16)) << 32;
                __hip_atomic_store(&mb->reply[k], v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            };
            if constexpr (CONT) {
#pragma unroll
                for (int k = 0; k < AMAX; ++k) if (k < A) put(k, __float_as_uint(act_c[k]));
            } else {
                put(0, (unsigned)action);
            }
        } else {
            if constexpr (CONT) {
#pragma unroll
                for (int k = 0; k < AMAX; ++k) if (k < A) reinterpret_cast<float*>(host_action_out)[k] = act_c[k];
            } else {
                reinterpret_cast<int*>(host_action_out)[0] = action;
            }
            if (host_ticket) {
                __threadfence_system();
                __hip_atomic_store(host_ticket, req.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if constexpr (!WAIT) break;
    // the waiting form goes back to its mailbox (the barrier above separates this request's reads of
    // xs / go / h2s from the next one's writes)
    served = expect;
    }
}

// Two entry points of the same body: the immediate form carries request and state in its arguments
// (320 bytes); the waiting form's launch is on the caller's critical path, so it carries 88.
template <bool CONT, int AMAX>
__global__ __launch_bounds__(128) void act_one_kernel(const float* __restrict__ params, int S, int A,
                                                      const ActRequest req, const OneState state,
                                                      void* host_action_out, unsigned* host_ticket) {
    act_one_body<CONT, AMAX, false>(params, S, A, req, &state, host_action_out, host_ticket, nullptr, nullptr, 0,
                                    nullptr);
}

template <bool CONT, int AMAX>
__global__ __launch_bounds__(128) void act_wait_kernel(const float* __restrict__ params, int S, int A,
                                                       ActMailbox* mb, unsigned* words, unsigned wait_ticks,
                                                       const ActBound bound) {
    act_one_body<CONT, AMAX, true>(params, S, A, ActRequest{}, nullptr, nullptr, nullptr, mb, words, wait_ticks,
                                   &bound);
}

static int launch_one(const float* params, int S, int A, int continuous, const ActRequest& req,
                      const OneState& state, void* host_action_out, unsigned* host_ticket, ActMailbox* mb,
                      unsigned* words, unsigned wait_ticks, const ActBound& bound, hipStream_t st) {
#define ACT_ONE(C, AM)                                                                              \
    if (mb) act_wait_kernel<C, AM><<<1, 128, 0, st>>>(params, S, A, mb, words, wait_ticks, bound);       \
    else act_one_kernel<C, AM><<<1, 128, 0, st>>>(params, S, A, req, state, host_action_out, host_ticket)
    if (continuous) {
        if (A <= 4) { ACT_ONE(true, 4); } else { ACT_ONE(true, 8); }
    } else {
        if (A <= 4) { ACT_ONE(false, 4); } else { ACT_ONE(false, 8); }
    }
#undef ACT_ONE
    return (int)hipGetLastError();
}

// Spin until *word == want; every so often ask the stream, so that a kernel that ended without
// writing the word (a fault) surfaces as its error instead of a hang.
static int spin_for(const unsigned* word, unsigned want, hipStream_t st) {
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return 0;
        if ((spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return __atomic_load_n(word, __ATOMIC_ACQUIRE) == want ? 0 : (int)hipErrorUnknown;
            if (q != hipErrorNotReady) return (int)q;
        }
    }
}

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
    ActRequest req{};
    req.state_row = state_row; req.action_row = action_row; req.logprob_row = logprob_row;
    req.value_row = value_row; req.seed = seed; req.rng_offset = rng_offset; req.action_var = action_var;
    req.seq = ticket;
    const int rc = launch_one(params, state_dim, action_dim, continuous, req, state, host_action_out,
                              host_ticket, nullptr, nullptr, 0, ActBound{}, st);
    if (rc) return rc;
    if (!host_ticket) return (int)hipStreamSynchronize(st);
    return spin_for(host_ticket, ticket, st);
}

// ---------------------------------------------------------------------------- sessions
struct ppo_act_session {
    ActMailbox* mb;
    unsigned* words;          // device: [0] last request served, [1] kernels that gave up
    hipStream_t stream;       // the session's own stream (non-blocking)
    hipEvent_t after;         // orders the first kernel after a drain behind the caller's stream
    int S, A, continuous;
    unsigned seq;             // last request posted
    unsigned seen_unserved;
    int live;                 // a kernel has been launched and is not yet known to have ended
    unsigned wait_ticks;
    // ppo_act_session_bind: everything of a step that does not change from step to step
    struct Binding {
        const float* params;
        uint64_t version;
        ActBound dev;         // what the waiting kernels carry in their arguments
        const float* host_state;
        void* action_out;
        void* after_stream;
        bool set;
    } bound, launched;        // launched: what the kernels now in flight were launched under
};

static bool same_launch(const ppo_act_session::Binding& a, const ppo_act_session::Binding& b) {
    return a.params == b.params && a.version == b.version && a.dev.states == b.dev.states &&
           a.dev.actions == b.dev.actions && a.dev.logprobs == b.dev.logprobs && a.dev.values == b.dev.values &&
           a.dev.seed == b.dev.seed && memcmp(&a.dev.action_var, &b.dev.action_var, sizeof(float)) == 0;
}

static int session_launch(ppo_act_session* s) {
    const ActRequest none{};
    const OneState no_state{};
    const int rc = launch_one(s->launched.params, s->S, s->A, s->continuous, none, no_state, nullptr, nullptr, s->mb,
                              s->words, s->wait_ticks, s->launched.dev, s->stream);
    if (!rc) s->live = 1;
    return rc;
}

extern "C" int ppo_act_session_create(ppo_act_session** out, int state_dim, int action_dim, int continuous,
                                      int wait_us) {
    if (!out || wait_us < 1 || wait_us > 100000) return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 || action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    ppo_act_session* s = new ppo_act_session();
    s->S = state_dim; s->A = action_dim; s->continuous = continuous;
    s->wait_ticks = (unsigned)wait_us * 100u;   // wall_clock64 counts at 100 MHz
    // The protocol needs fine-grained coherent host memory (the device's system-scope loads must see the
    // host's stores and the host the reply): asked for explicitly, whatever HIP_HOST_COHERENT says.
    hipError_t e = hipHostMalloc((void**)&s->mb, sizeof(ActMailbox), hipHostMallocCoherent);
    if (e == hipSuccess) e = hipMalloc((void**)&s->words, 2 * sizeof(unsigned));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->after, hipEventDisableTiming);
    // cleared on the session's own (non-blocking) stream, ahead of its first kernel
    if (e == hipSuccess) e = hipMemsetAsync(s->words, 0, 2 * sizeof(unsigned), s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        if (s->mb) (void)hipHostFree(s->mb);
        if (s->words) (void)hipFree(s->words);
        if (s->stream) (void)hipStreamDestroy(s->stream);
        if (s->after) (void)hipEventDestroy(s->after);
        delete s;
        return (int)e;
    }
    memset(s->mb, 0, sizeof(ActMailbox));
    *out = s;
    return 0;
}

// Withdraw the waiting kernel (if any) and wait until every kernel of the session has ended: after
// this the rows written so far are visible to any stream, and the weights may change.
extern "C" int ppo_act_session_drain(ppo_act_session* s) {
    if (!s) return PPO_E_BADARG;
    if (s->live) __atomic_store_n(&s->mb->ctrl, ACT_CANCEL_BIT, __ATOMIC_RELEASE);
    const hipError_t e = hipStreamSynchronize(s->stream);
    __atomic_store_n(&s->mb->ctrl, 0u, __ATOMIC_RELEASE);
    s->live = 0;
    s->seen_unserved = __atomic_load_n(&s->mb->unserved, __ATOMIC_ACQUIRE);
    return (int)e;
}

extern "C" int ppo_act_session_destroy(ppo_act_session* s) {
    if (!s) return 0;
    const int rc = ppo_act_session_drain(s);
    (void)hipStreamDestroy(s->stream);
    (void)hipEventDestroy(s->after);
    (void)hipHostFree(s->mb);
    (void)hipFree(s->words);
    delete s;
    return rc;
}

// What stays the same from step to step is bound once: the weights and their version, the variance,
// the host's state / action buffers, the seed and the four buffer columns (dense rows: state_dim
// floats, one int32 or action_dim floats, one float, one float).  The columns, the seed and the
// variance travel in the waiting kernel's launch arguments, the weights are loaded at its launch:
// a step under another binding than the waiting kernel's withdraws that kernel first.
extern "C" int ppo_act_session_bind(ppo_act_session* s, const float* params, uint64_t params_version,
                                    float action_var, const float* host_state, void* action_out,
                                    uint64_t seed, float* states, void* actions, float* logprobs,
                                    float* values, void* after_stream) {
    if (!s || !params || !host_state || !action_out || !states || !actions || !logprobs || !values)
        return PPO_E_BADARG;
    s->bound.params = params;
    s->bound.version = params_version;
    s->bound.dev.states = states;
    s->bound.dev.actions = (char*)actions;
    s->bound.dev.logprobs = logprobs;
    s->bound.dev.values = values;
    s->bound.dev.seed = seed;
    s->bound.dev.action_var = action_var;
    s->bound.host_state = host_state;
    s->bound.action_out = action_out;
    s->bound.after_stream = after_stream;
    s->bound.set = true;
    return 0;
}

extern "C" int ppo_act_session_step_bound(ppo_act_session* s, int64_t row, uint64_t rng_offset) {
    if (!s || !s->bound.set || row < 0 || row > 0xffffffffll) return PPO_E_BADARG;
    // The kernel that waits for this request was launched earlier, under the binding of that moment.
    // Another binding - other weights, or the same vector written since (the caller's version number
    // moved), other columns, seed or variance - or a used-up counter: withdraw it and start over with a
    // kernel launched now.
    if (!same_launch(s->bound, s->launched) || s->seq >= 0x7ffffff0u) {
        const int rc = ppo_act_session_drain(s);
        if (rc) return rc;
        s->launched = s->bound;
        if (s->seq >= 0x7ffffff0u) {
            s->seq = 0;
            memset(s->mb->req, 0, sizeof(s->mb->req));                                  // no dword may carry an old number
            hipError_t e = hipMemsetAsync(s->words, 0, sizeof(unsigned), s->stream);   // the drained stream is idle
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    ActMailbox* const mb = s->mb;
    const unsigned k = ++s->seq;
    const unsigned tag = k << 16;
    {   // the request as tagged halves; no store has to be ordered against another
        unsigned* const q = mb->req;
        auto put = [&](int i, unsigned half) { __atomic_store_n(&q[i], tag | (half & 0xffffu), __ATOMIC_RELAXED); };
        put(0, (unsigned)row); put(1, (unsigned)row >> 16);
        put(2, (unsigned)rng_offset); put(3, (unsigned)(rng_offset >> 16));
        put(4, (unsigned)(rng_offset >> 32)); put(5, (unsigned)(rng_offset >> 48));
        const float* const x = s->bound.host_state;
        for (int u = 0; u < s->S; ++u) {
            unsigned w;
            memcpy(&w, &x[u], 4);
            put(6 + 2 * u, w); put(7 + 2 * u, w >> 16);
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
    }
    int rc = 0;
    if (!s->live) {                                   // nobody is waiting for this request
        // first kernel since the drain: it must see what the caller's stream did meanwhile (new weights)
        hipError_t e = hipEventRecord(s->after, (hipStream_t)s->bound.after_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s->stream, s->after, 0);
        if (e != hipSuccess) return (int)e;
        if ((rc = session_launch(s))) return rc;
    }
    const int nw = s->continuous ? s->A : 1;          // action words; two tagged dwords each
    unsigned r[16];
    const unsigned* const reply = reinterpret_cast<const unsigned*>(mb->reply);
    for (unsigned spins = 1;; ++spins) {
        bool all = true;
        for (int i = 0; i < 2 * nw; ++i) {
            r[i] = __atomic_load_n(&reply[i], __ATOMIC_RELAXED);
            all = all && (r[i] & 0xffff0000u) == tag;
        }
        if (all) break;
        const unsigned gave_up = __atomic_load_n(&mb->unserved, __ATOMIC_ACQUIRE);
        if (gave_up != s->seen_unserved) {            // the waiting kernel left before the request came
            s->seen_unserved = gave_up;
            if ((rc = session_launch(s))) return rc;
        }
        if ((spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(s->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return (int)q;
            if (q == hipSuccess && (__atomic_load_n(&reply[0], __ATOMIC_RELAXED) & 0xffff0000u) != tag &&
                __atomic_load_n(&mb->unserved, __ATOMIC_ACQUIRE) == s->seen_unserved)
                return (int)hipErrorUnknown;          // the kernel ended, it neither served nor gave up
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    unsigned* const out = (unsigned*)s->bound.action_out;
    for (int i = 0; i < nw; ++i) out[i] = (r[2 * i] & 0xffffu) | (r[2 * i + 1] << 16);
    return 0;
}

// One step with everything given per call.  Served through the binding: a call whose rows are row r of
// the columns bound so far (same weights, version, variance, seed, host buffers and stream) is that
// binding's step r - a caller that walks down its buffer row by row keeps its waiting kernel - and any
// other call binds anew, with its rows as row 0.
extern "C" int ppo_act_session_step(ppo_act_session* s, const float* params, uint64_t params_version,
                                    float action_var,
                                    const float* host_state, uint64_t seed, uint64_t rng_offset,
                                    float* state_row, void* action_row, float* logprob_row,
                                    float* value_row, void* action_out, void* after_stream) {
    if (!s || !params || !host_state || !state_row || !action_row || !logprob_row || !value_row || !action_out)
        return PPO_E_BADARG;
    const ptrdiff_t action_bytes = (ptrdiff_t)sizeof(float) * (s->continuous ? s->A : 1);
    int64_t row = -1;
    const ppo_act_session::Binding& b = s->bound;
    if (b.set && b.params == params && b.version == params_version && b.dev.seed == seed &&
        memcmp(&b.dev.action_var, &action_var, sizeof(float)) == 0 && b.host_state == host_state &&
        b.action_out == action_out && b.after_stream == after_stream) {
        const ptrdiff_t r = logprob_row - b.dev.logprobs;
        if (r >= 0 && r <= 0xffffffffll && value_row == b.dev.values + r && state_row == b.dev.states + r * s->S &&
            (char*)action_row == b.dev.actions + r * action_bytes)
            row = r;
    }
    if (row < 0) {
        const int rc = ppo_act_session_bind(s, params, params_version, action_var, host_state, action_out, seed,
                                            state_row, action_row, logprob_row, value_row, after_stream);
        if (rc) return rc;
        row = 0;
    }
    return ppo_act_session_step_bound(s, row, rng_offset);
}
