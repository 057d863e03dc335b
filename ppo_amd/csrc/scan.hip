// Monte-Carlo return scan, moments and normalisation (HBM-bound stage of PPO.update).
//
// The recurrence of PPO.py:204-208 is R_t = r_t + gamma * (terminal_t ? 0 : R_{t+1}) in fp64.
// Each step is the affine map x -> b + a*x with (a, b) = (terminal ? 0 : gamma, r); maps compose
// associatively, so a flat sequence is scanned in three launches:
//   1. every block composes the maps of its 2048 elements (wave shuffles + LDS)        -> aggregate
//   2. one wave turns the block aggregates into the carry entering each block          -> carries
//   3. every block re-composes, gives each thread the exact carry entering its 8 elements, and
//      runs the reference's own serial recurrence (separate multiply and add, as Python does)
//      on them; it rounds to fp32 once (PPO.py:211) and leaves fp64 moments of the rounded values.
// Only the carries go through re-associated arithmetic (relative error ~1e-16 before the fp32
// rounding); everything inside a thread's 8 elements is the reference's operation order.
#include "ppo_common.h"
#include <cstdlib>

#define SCAN_BLOCK 256
#define SCAN_E 8
#define SCAN_CHUNK (SCAN_BLOCK * SCAN_E)

struct Aff {  // x -> b + a * x
    double a, b;
};

__device__ __forceinline__ Aff compose(const Aff& l, const Aff& r) {  // l after r (l is to the left)
    Aff o;
    o.a = l.a * r.a;
    o.b = l.b + l.a * r.b;
    return o;
}

__device__ __forceinline__ Aff shfl_down_aff(const Aff& v, int off) {
    Aff o;
    o.a = __shfl_down(v.a, off);
    o.b = __shfl_down(v.b, off);
    return o;
}

// inclusive suffix composition over the 64 lanes: lane i gets F_i o F_{i+1} o ... o F_63
__device__ __forceinline__ Aff wave_suffix(Aff v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Aff o = shfl_down_aff(v, off);
        if (lane + off < 64) v = compose(v, o);
    }
    return v;
}

struct Mom {  // count, mean, M2
    double n, mean, m2;
};

__device__ __forceinline__ Mom merge(const Mom& x, const Mom& y) {  // Chan et al. pairwise merge
    if (y.n == 0.0) return x;
    if (x.n == 0.0) return y;
    Mom o;
    o.n = x.n + y.n;
    const double d = y.mean - x.mean;
    o.mean = x.mean + d * (y.n / o.n);
    o.m2 = x.m2 + y.m2 + d * d * (x.n * y.n / o.n);
    return o;
}

// Merge over a group of `nwaves` consecutive waves (all threads of the workgroup call it: one barrier
// inside); lds: >= nwaves entries of the group's own.  Lanes merge by shuffles, then the waves in order.
__device__ __forceinline__ Mom group_merge(Mom v, Mom* lds, int wave_in_group, int nwaves) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Mom o;
        o.n = __shfl_down(v.n, off);
        o.mean = __shfl_down(v.mean, off);
        o.m2 = __shfl_down(v.m2, off);
        if (lane + off < 64) v = merge(v, o);
    }
    if (lane == 0) lds[wave_in_group] = v;
    __syncthreads();
    Mom t = lds[0];
    for (int w = 1; w < nwaves; ++w) t = merge(t, lds[w]);
    return t;
}

__device__ __forceinline__ Mom block_merge(Mom v, Mom* lds /* >= 4 */) {
    return group_merge(v, lds, threadIdx.x >> 6, (int)(blockDim.x >> 6));
}

template <typename RT>
__device__ __forceinline__ void load_elems(const RT* __restrict__ rewards,
                                           const uint8_t* __restrict__ terminals, int64_t base,
                                           int64_t n, double (&r)[SCAN_E], bool (&term)[SCAN_E]) {
#pragma unroll
    for (int j = 0; j < SCAN_E; ++j) {
        const int64_t i = base + j;
        const bool ok = i < n;
        r[j] = ok ? (double)rewards[i] : 0.0;
        term[j] = ok ? terminals[i] != 0 : false;
    }
}

// composite of one thread's elements: maps carry at the chunk's end to R at its first element.
// Out-of-range elements are the identity (a=1,b=0).
__device__ __forceinline__ Aff thread_aff(const double (&r)[SCAN_E], const bool (&term)[SCAN_E],
                                          int64_t base, int64_t n, double gamma) {
    Aff f;
    f.a = 1.0;
    f.b = 0.0;
#pragma unroll
    for (int j = SCAN_E - 1; j >= 0; --j) {
        if (base + j < n) {
            const double a = term[j] ? 0.0 : gamma;
            f.b = r[j] + a * f.b;
            f.a = a * f.a;
        }
    }
    return f;
}

template <typename RT>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_flat_reduce(const RT* __restrict__ rewards,
                                                               const uint8_t* __restrict__ terminals,
                                                               int64_t n, double gamma,
                                                               Aff* __restrict__ block_aff) {
    __shared__ Aff wav[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_E;
    double r[SCAN_E];
    bool term[SCAN_E];
    load_elems(rewards, terminals, base, n, r, term);
    const Aff s = wave_suffix(thread_aff(r, term, base, n, gamma), lane);
    if (lane == 0) wav[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        Aff t = wav[0];
        for (int w = 1; w < SCAN_BLOCK / 64; ++w) t = compose(t, wav[w]);
        block_aff[blockIdx.x] = t;
    }
}

// One wave: carry entering block b (the return just right of its last element) for all blocks.
__device__ __forceinline__ void carries_of_blocks(const Aff* block_aff, int nb, double* carries, int lane) {
    const int per = (nb + 63) / 64;
    const int lo = lane * per, hi = min(nb, lo + per);
    Aff f;
    f.a = 1.0;
    f.b = 0.0;
    for (int b = hi - 1; b >= lo; --b) f = compose(block_aff[b], f);  // maps carry at hi to R at lo
    const Aff s = wave_suffix(f, lane);
    Aff x = shfl_down_aff(s, 1);  // composition of everything right of this lane's segment
    if (lane == 63) { x.a = 1.0; x.b = 0.0; }
    double carry = x.b;  // applied to the sequence end, where the return is 0 (PPO.py:203)
    for (int b = hi - 1; b >= lo; --b) {
        carries[b] = carry;
        carry = block_aff[b].b + block_aff[b].a * carry;
    }
}

__global__ __launch_bounds__(64) void scan_flat_carries(const Aff* __restrict__ block_aff, int nb,
                                                        double* __restrict__ carries) {
    carries_of_blocks(block_aff, nb, carries, threadIdx.x);
}

// A thread's 8 elements from the carry entering them: the reference's own recurrence, one rounding per
// multiply and per add (Python floats), rounded to fp32 once (PPO.py:211); leaves the thread's moments.
__device__ __forceinline__ Mom apply_elems(const double (&r)[SCAN_E], const bool (&term)[SCAN_E], int64_t base,
                                           int64_t n, double gamma, double d, float* __restrict__ returns) {
    float out[SCAN_E];
    double sum = 0.0;
    int cnt = 0;
#pragma unroll
    for (int j = SCAN_E - 1; j >= 0; --j) {
        if (base + j < n) {
            if (term[j]) d = 0.0;
            d = __dadd_rn(r[j], __dmul_rn(gamma, d));
            out[j] = (float)d;
            sum += (double)out[j];
            ++cnt;
        }
    }
    Mom m;
    m.n = (double)cnt;
    m.mean = cnt ? sum / cnt : 0.0;
    m.m2 = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_E; ++j) {
        if (base + j < n) {
            returns[base + j] = out[j];
            const double dd = (double)out[j] - m.mean;
            m.m2 += dd * dd;
        }
    }
    return m;
}

template <typename RT>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_flat_apply(const RT* __restrict__ rewards,
                                                              const uint8_t* __restrict__ terminals,
                                                              int64_t n, double gamma,
                                                              const double* __restrict__ carries,
                                                              float* __restrict__ returns,
                                                              Mom* __restrict__ block_mom) {
    __shared__ Aff wav[SCAN_BLOCK / 64];
    __shared__ double wcarry[SCAN_BLOCK / 64];
    __shared__ Mom mom_lds[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_E;
    double r[SCAN_E];
    bool term[SCAN_E];
    load_elems(rewards, terminals, base, n, r, term);
    const Aff s = wave_suffix(thread_aff(r, term, base, n, gamma), lane);
    if (lane == 0) wav[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = carries[blockIdx.x];
        for (int w = SCAN_BLOCK / 64 - 1; w >= 0; --w) {
            wcarry[w] = c;
            c = wav[w].b + wav[w].a * c;
        }
    }
    __syncthreads();
    Aff x = shfl_down_aff(s, 1);
    if (lane == 63) { x.a = 1.0; x.b = 0.0; }
    const double d0 = x.b + x.a * wcarry[wave];
    const Mom m = apply_elems(r, term, base, n, gamma, d0, returns);
    const Mom bm = block_merge(m, mom_lds);
    if (threadIdx.x == 0) block_mom[blockIdx.x] = bm;
}

// Time-major [n_steps, n_envs]: one thread per env walks its column backwards (exactly the
// reference's serial order, so the result is bit-exact); loads and stores of a step are coalesced
// across envs.  The recurrence itself is two dependent fp64 operations per step; what has to be
// hidden is memory latency, so the rewards / terminals of ROWS_PF steps are fetched into registers
// before the chain runs over them (ROWS_PF loads in flight per lane instead of one).  One wave per
// workgroup, so that a few thousand envs already spread over many CUs.
#define ROWS_BLOCK 64
#define ROWS_PF 32
__global__ __launch_bounds__(ROWS_BLOCK) void scan_rows_kernel(const float* __restrict__ rewards,
                                                               const uint8_t* __restrict__ terminals,
                                                               int64_t n_steps, int64_t n_envs,
                                                               double gamma,
                                                               float* __restrict__ returns,
                                                               Mom* __restrict__ block_mom) {
    __shared__ Mom mom_lds[4];
    const int64_t e = (int64_t)blockIdx.x * ROWS_BLOCK + threadIdx.x;
    Mom m;
    m.n = 0.0;
    m.mean = 0.0;
    m.m2 = 0.0;
    if (e < n_envs) {
        double d = 0.0, shift = 0.0, s1 = 0.0, s2 = 0.0;
        for (int64_t hi = n_steps; hi > 0; hi -= ROWS_PF) {   // steps hi-1 .. hi-ROWS_PF, backwards
            float r[ROWS_PF];
            uint8_t tm[ROWS_PF];
#pragma unroll
            for (int j = 0; j < ROWS_PF; ++j) {
                const int64_t t = hi - 1 - j;
                const int64_t i = (t >= 0 ? t : 0) * n_envs + e;
                r[j] = rewards[i];
                tm[j] = terminals[i];
            }
#pragma unroll
            for (int j = 0; j < ROWS_PF; ++j) {
                const int64_t t = hi - 1 - j;
                if (t >= 0) {
                    if (tm[j]) d = 0.0;
                    d = __dadd_rn((double)r[j], __dmul_rn(gamma, d));
                    const float o = (float)d;
                    returns[t * n_envs + e] = o;
                    if (t == n_steps - 1) shift = (double)o;
                    const double y = (double)o - shift;
                    s1 += y;
                    s2 += y * y;
                }
            }
        }
        m.n = (double)n_steps;
        m.mean = shift + s1 / m.n;
        m.m2 = fmax(s2 - s1 * s1 / m.n, 0.0);
    }
    const Mom bm = block_merge(m, mom_lds);
    if (threadIdx.x == 0) block_mom[blockIdx.x] = bm;
}

// The block moments merged in sequence order by 256 threads (`group` = their wave group of four, see
// group_merge); thread 0 of the group returns the total.
__device__ __forceinline__ Mom finalize_moments(const Mom* block_mom, int nb, int t256, Mom* lds) {
    Mom m;
    m.n = 0.0;
    m.mean = 0.0;
    m.m2 = 0.0;
    // contiguous segments keep the merge order equal to the sequence order
    const int per = (nb + 255) / 256;
    const int lo = t256 * per, hi = min(nb, lo + per);
    for (int b = lo; b < hi; ++b) m = merge(m, block_mom[b]);
    return group_merge(m, lds, t256 >> 6, 4);
}

__global__ __launch_bounds__(256) void moments_finalize(const Mom* __restrict__ block_mom, int nb,
                                                        double* __restrict__ out3) {
    __shared__ Mom mom_lds[4];
    const Mom t = finalize_moments(block_mom, nb, threadIdx.x, mom_lds);
    if (threadIdx.x == 0) {
        out3[0] = t.n;
        out3[1] = t.mean;
        out3[2] = t.m2;
    }
}

// The whole flat scan in ONE launch for sequences of up to SCAN_SMALL_MAX steps (the reference's buffers
// are 900-6000 steps): one workgroup of 1024 threads acts as four 256-thread blocks of the three-launch
// form above - same composition order per block, the same carries_of_blocks, apply_elems and moment
// merges - with the block aggregates, carries and block moments in LDS instead of a workspace, so the
// returns and moments are bit-identical to the three-launch form's.
#define SCAN_SMALL_BLOCKS 4
#define SCAN_SMALL_MAX (SCAN_SMALL_BLOCKS * SCAN_CHUNK)
template <typename RT>
__global__ __launch_bounds__(SCAN_SMALL_BLOCKS * SCAN_BLOCK) void scan_flat_small(
    const RT* __restrict__ rewards, const uint8_t* __restrict__ terminals, int64_t n, double gamma,
    float* __restrict__ returns, double* __restrict__ out3) {
    constexpr int WPB = SCAN_BLOCK / 64;                 // waves per virtual block
    __shared__ Aff wav[SCAN_SMALL_BLOCKS * WPB];
    __shared__ Aff block_aff[SCAN_SMALL_BLOCKS];
    __shared__ double carries[SCAN_SMALL_BLOCKS];
    __shared__ double wcarry[SCAN_SMALL_BLOCKS * WPB];
    __shared__ Mom mom_lds[SCAN_SMALL_BLOCKS * WPB];
    __shared__ Mom block_mom[SCAN_SMALL_BLOCKS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vb = threadIdx.x / SCAN_BLOCK, tl = threadIdx.x % SCAN_BLOCK;   // virtual block, thread in it
    const int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    const int64_t base = (int64_t)vb * SCAN_CHUNK + (int64_t)tl * SCAN_E;
    double r[SCAN_E];
    bool term[SCAN_E];
    load_elems(rewards, terminals, base, n, r, term);
    const Aff s = wave_suffix(thread_aff(r, term, base, n, gamma), lane);
    if (lane == 0) wav[wave] = s;
    __syncthreads();
    if (tl == 0) {                                       // (scan_flat_reduce's last step)
        Aff t = wav[vb * WPB];
        for (int w = 1; w < WPB; ++w) t = compose(t, wav[vb * WPB + w]);
        block_aff[vb] = t;
    }
    __syncthreads();
    if (wave == 0) carries_of_blocks(block_aff, nb, carries, lane);
    __syncthreads();
    if (tl == 0 && vb < nb) {                            // (scan_flat_apply's carry per wave)
        double c = carries[vb];
        for (int w = WPB - 1; w >= 0; --w) {
            wcarry[vb * WPB + w] = c;
            c = wav[vb * WPB + w].b + wav[vb * WPB + w].a * c;
        }
    }
    __syncthreads();
    Aff x = shfl_down_aff(s, 1);
    if (lane == 63) { x.a = 1.0; x.b = 0.0; }
    const double d0 = x.b + x.a * wcarry[wave];          // (a block past the end holds no element: unused)
    const Mom m = apply_elems(r, term, base, n, gamma, d0, returns);
    const Mom bm = group_merge(m, mom_lds + vb * WPB, wave % WPB, WPB);
    if (tl == 0) block_mom[vb] = bm;
    __syncthreads();
    // (moments_finalize: its 256 threads are virtual block 0; the other groups merge nothing)
    const Mom t = finalize_moments(block_mom, vb == 0 ? nb : 0, tl, mom_lds + vb * WPB);
    if (threadIdx.x == 0) {
        out3[0] = t.n;
        out3[1] = t.mean;
        out3[2] = t.m2;
    }
}

// PPO.py:212 (elementwise) + PPO.py:221.  fp32 arithmetic exactly as the reference's tensors:
// mean and std are fp32 scalars, 1e-7 is added to the fp32 std, the division is IEEE.
// returns_norm may be the same buffer as returns (each thread reads an element before it writes
// it), so neither of the two is declared __restrict__.  16-byte accesses when all four pointers
// are 16-byte aligned (VEC), one element per thread otherwise.
template <bool VEC>
__global__ __launch_bounds__(256) void normalize_adv_kernel(const float* returns,
                                                            const float* __restrict__ old_values,
                                                            int64_t n,
                                                            const double* __restrict__ moments,
                                                            float* returns_norm,
                                                            float* __restrict__ adv) {
    const double cnt = moments[0];
    const float mean = (float)moments[1];
    const float stdv = (float)sqrt(moments[2] / fmax(cnt - 1.0, 1.0));
    const float denom = stdv + 1e-7f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            const f32x4 r = reinterpret_cast<const f32x4*>(returns)[i];
            const f32x4 v = reinterpret_cast<const f32x4*>(old_values)[i];
            f32x4 rn, ad;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rn[j] = (r[j] - mean) / denom;
                ad[j] = rn[j] - v[j];
            }
            reinterpret_cast<f32x4*>(returns_norm)[i] = rn;
            reinterpret_cast<f32x4*>(adv)[i] = ad;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
            const float rn = (returns[i] - mean) / denom;
            returns_norm[i] = rn;
            adv[i] = rn - old_values[i];
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float rn = (returns[i] - mean) / denom;
            returns_norm[i] = rn;
            adv[i] = rn - old_values[i];
        }
    }
}

// ---------------------------------------------------------------------------- host side
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// workspace carve for the scan: [Aff nb][double nb][Mom nb]
size_t ppo_scan_ws_bytes(int64_t n) {
    const size_t nb = (size_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK) + 1;
    const size_t nb_rows = (size_t)((n + ROWS_BLOCK - 1) / ROWS_BLOCK) + 1;   // >= one block per 64 envs
    const size_t m = nb > nb_rows ? nb : nb_rows;
    return align_up(m * sizeof(Aff), 256) + align_up(m * sizeof(double), 256) +
           align_up(m * sizeof(Mom), 256);
}

// PPO_SCAN_SMALL=0 keeps short sequences on the three-launch form (cross-check of the two forms).
static bool scan_small_off() {
    static const bool off = [] { const char* e = getenv("PPO_SCAN_SMALL"); return e && e[0] == '0'; }();
    return off;
}

// The three launches of the flat scan + the moments merge, on scratch carved by the caller:
// aff[nb + 1], carries[nb + 1], mom[nb + 1] with nb = ceil(n / SCAN_CHUNK).  Every entry a kernel
// reads has been written by an earlier kernel of the same call: the scratch needs no clearing.
static int launch_scan_flat(const void* rewards, int rewards_are_f64, const uint8_t* terminals,
                            int64_t n, double gamma, float* returns_out, double* moments_out,
                            Aff* aff, double* carries, Mom* mom, hipStream_t st) {
    const int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    if (n <= SCAN_SMALL_MAX && !scan_small_off()) {   // one launch instead of four; same bits
        if (rewards_are_f64)
            scan_flat_small<double><<<1, SCAN_SMALL_BLOCKS * SCAN_BLOCK, 0, st>>>((const double*)rewards, terminals, n,
                                                                               gamma, returns_out, moments_out);
        else
            scan_flat_small<float><<<1, SCAN_SMALL_BLOCKS * SCAN_BLOCK, 0, st>>>((const float*)rewards, terminals, n,
                                                                              gamma, returns_out, moments_out);
        return (int)hipGetLastError();
    }
    if (rewards_are_f64) {
        scan_flat_reduce<double><<<nb, SCAN_BLOCK, 0, st>>>((const double*)rewards, terminals, n,
                                                            gamma, aff);
        scan_flat_carries<<<1, 64, 0, st>>>(aff, nb, carries);
        scan_flat_apply<double><<<nb, SCAN_BLOCK, 0, st>>>((const double*)rewards, terminals, n,
                                                           gamma, carries, returns_out, mom);
    } else {
        scan_flat_reduce<float><<<nb, SCAN_BLOCK, 0, st>>>((const float*)rewards, terminals, n,
                                                           gamma, aff);
        scan_flat_carries<<<1, 64, 0, st>>>(aff, nb, carries);
        scan_flat_apply<float><<<nb, SCAN_BLOCK, 0, st>>>((const float*)rewards, terminals, n,
                                                          gamma, carries, returns_out, mom);
    }
    moments_finalize<<<1, 256, 0, st>>>(mom, nb, moments_out);
    return (int)hipGetLastError();
}

extern "C" int ppo_returns_scan_flat(const void* rewards, int rewards_are_f64,
                                     const uint8_t* terminals, int64_t n, double gamma,
                                     float* returns_out, double* moments_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!rewards || !terminals || !returns_out || !moments_out || n <= 0 ||
        n > ((int64_t)1 << 40))
        return PPO_E_BADARG;
    if (!workspace || workspace_bytes < ppo_scan_ws_bytes(n)) return PPO_E_WORKSPACE;
    const int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    char* w = (char*)workspace;
    Aff* aff = (Aff*)w;
    w += align_up((size_t)(nb + 1) * sizeof(Aff), 256);
    double* carries = (double*)w;
    w += align_up((size_t)(nb + 1) * sizeof(double), 256);
    Mom* mom = (Mom*)w;
    return launch_scan_flat(rewards, rewards_are_f64, terminals, n, gamma, returns_out, moments_out,
                            aff, carries, mom, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- no-allocation variant
// Same scan for callers that cannot allocate device memory (code under HIP graph capture, tight
// serving loops): its block aggregates, carries and block moments live in a static device array
// instead of a caller-provided workspace.  The array is sized for PPO_SCAN_NOALLOC_MAX elements
// (394 KB, held by every GPU the library runs on until the process ends).  In host code the array's
// name is only a stand-in for the symbol: its device address is looked up once per device with
// hipGetSymbolAddress (each GPU holds its own copy) and only that address is handed to kernels.
#define NOALLOC_NB ((PPO_SCAN_NOALLOC_MAX + SCAN_CHUNK - 1) / SCAN_CHUNK + 1)
#define NOALLOC_AFF_BYTES ((NOALLOC_NB * sizeof(Aff) + 255) / 256 * 256)
#define NOALLOC_CARRY_BYTES ((NOALLOC_NB * sizeof(double) + 255) / 256 * 256)
#define NOALLOC_MOM_BYTES ((NOALLOC_NB * sizeof(Mom) + 255) / 256 * 256)
static __device__ __align__(256) unsigned char scratch[NOALLOC_AFF_BYTES + NOALLOC_CARRY_BYTES + NOALLOC_MOM_BYTES];

#define NOALLOC_MAX_DEVICES 64
static void* scratch_addr[NOALLOC_MAX_DEVICES];  // device address of `scratch` per device, 0 = not looked up yet

extern "C" int ppo_returns_scan_flat_noalloc(const void* rewards, int rewards_are_f64,
                                             const uint8_t* terminals, int64_t n, double gamma,
                                             float* returns_out, double* moments_out, void* stream) {
    if (!rewards || !terminals || !returns_out || !moments_out || n <= 0) return PPO_E_BADARG;
    if (n > (int64_t)PPO_SCAN_NOALLOC_MAX) return PPO_E_TOO_LARGE;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return PPO_E_BADARG;
    // the scratch address used below is the CURRENT device's: a stream of another device would run the
    // kernels there, on an address that belongs to this one
    hipDevice_t stream_dev = dev;
    if (stream && hipStreamGetDevice((hipStream_t)stream, &stream_dev) == hipSuccess && (int)stream_dev != dev)
        return PPO_E_BADARG;
    if (!scratch_addr[dev]) {  // first call on this device (which is current)
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return PPO_E_BADARG;
        scratch_addr[dev] = p;
    }
    char* w = (char*)scratch_addr[dev];
    Aff* aff = (Aff*)w;
    double* carries = (double*)(w + NOALLOC_AFF_BYTES);
    Mom* mom = (Mom*)(w + NOALLOC_AFF_BYTES + NOALLOC_CARRY_BYTES);
    return launch_scan_flat(rewards, rewards_are_f64, terminals, n, gamma, returns_out, moments_out,
                            aff, carries, mom, (hipStream_t)stream);
}

extern "C" int ppo_returns_scan_rows(const float* rewards_tn, const uint8_t* terminals_tn,
                                     int64_t n_steps, int64_t n_envs, double gamma,
                                     float* returns_tn, double* moments_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!rewards_tn || !terminals_tn || !returns_tn || !moments_out || n_steps <= 0 ||
        n_envs <= 0 || n_steps * n_envs > ((int64_t)1 << 40))
        return PPO_E_BADARG;
    if (!workspace || workspace_bytes < ppo_scan_ws_bytes(n_steps * n_envs)) return PPO_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((n_envs + ROWS_BLOCK - 1) / ROWS_BLOCK);
    Mom* mom = (Mom*)workspace;   // nb block moments; ppo_scan_ws_bytes() covers (nb + 1) * sizeof(Mom)
    scan_rows_kernel<<<nb, ROWS_BLOCK, 0, st>>>(rewards_tn, terminals_tn, n_steps, n_envs, gamma,
                                                returns_tn, mom);
    moments_finalize<<<1, 256, 0, st>>>(mom, nb, moments_out);
    return (int)hipGetLastError();
}

extern "C" int ppo_normalize_advantage(const float* returns, const float* old_values, int64_t n,
                                       const double* moments, float* returns_norm,
                                       float* advantages, void* stream) {
    if (!returns || !old_values || !moments || !returns_norm || !advantages || n <= 0)
        return PPO_E_BADARG;
    const bool vec = (((uintptr_t)returns | (uintptr_t)old_values | (uintptr_t)returns_norm |
                       (uintptr_t)advantages) & 15) == 0;
    const int64_t want = ((vec ? (n + 3) / 4 : n) + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    if (vec)
        normalize_adv_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(returns, old_values, n, moments,
                                                                         returns_norm, advantages);
    else
        normalize_adv_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(returns, old_values, n, moments,
                                                                          returns_norm, advantages);
    return (int)hipGetLastError();
}
