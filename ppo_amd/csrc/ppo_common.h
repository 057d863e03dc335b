// Shared device code of the gfx950 PPO kernels: parameter layout, the wave-level MLP tile
// (fp32 MFMA, sample-on-lane orientation), the distribution heads, Philox.
//
// Orientation used everywhere: one wave owns a tile of 32 samples.  Lane l holds sample
// c = l & 31; h = l >> 5 selects which half of the hidden units the lane's registers carry.
// A 64-unit activation vector of the 32 samples is two 32x32 MFMA accumulator tiles
// (f32x16 t[2]): register r of tile t on lane (c, h) is unit 32*t + rho(r, h) of sample c, with
//   rho(r, h) = (r & 3) + 8 * (r >> 2) + 4 * h
// (the C/D layout of v_mfma_f32_32x32x2_f32).  Because the B operand of that instruction wants
// B[k = h][j = c], an accumulator register is directly the B operand of the next layer's k-step
// {rho(r,0), rho(r,1)}: activations never leave registers between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HID PPO_HIDDEN
#define TILE_M 32   // samples per wave tile

#define HEAD_CATEGORICAL 0
#define HEAD_GAUSSIAN 1
#define HEAD_VALUE 2

#define LOG_2PI_F 1.8378770664093453f
#define EPS_PROB_F 1.1920928955078125e-07f  // finfo(float32).eps, torch/distributions/utils.py:121

// ------------------------------------------------------------------ parameter layout
struct NetLayout {  // offsets (floats) of one MLP inside the flat parameter vector
    int w1, b1, w2, b2, w3, b3, n;
};

__host__ __device__ inline NetLayout net_layout(int base, int S, int out) {
    NetLayout L;
    L.w1 = base;
    L.b1 = L.w1 + HID * S;
    L.w2 = L.b1 + HID;
    L.b2 = L.w2 + HID * HID;
    L.w3 = L.b2 + HID;
    L.b3 = L.w3 + out * HID;
    L.n = L.b3 + out - base;
    return L;
}

__host__ __device__ inline int n_actor_params(int S, int A) { return net_layout(0, S, A).n; }
__host__ __device__ inline int n_critic_params(int S) { return net_layout(0, S, 1).n; }

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ tanh
// On gfx950 the fp32 MFMA runs on the same datapath as the fp32 VALU (measured: VALU fillers
// between v_mfma_f32_32x32x2_f32 add their full issue time at 1, 2 and 4 waves per SIMD), so every
// VALU instruction of the 128 tanh evaluations per sample is paid for in matrix throughput.
// tanh(x) = 1 - 2 / (exp(2x) + 1) in four instructions (v_exp, v_add, v_rcp, v_fma); the factor
// 2*log2(e) that v_exp_f32 (base 2) needs is folded into the layer's weights and bias when they are
// staged into LDS.  Saturates correctly (exp -> inf gives 1, exp -> 0 gives -1).  Absolute error
// <= ~1.5e-7 over the whole range, i.e. the rounding noise of a value near 1; the RELATIVE error
// near 0 is not small, which nothing here needs (hidden units feed dot products, the Gaussian mean
// feeds a difference).  -DPPO_TANH_ACCURATE selects a <= 2.5 ulp version (odd polynomial below
// 0.625) at ~3x the instructions.
#define TANH_PRESCALE 2.8853900817779268f  // 2 * log2(e)

// <= 2.5 ulp: odd polynomial below 0.625, 1 - 2/(exp(2x)+1) above.
__device__ __forceinline__ float tanh_accurate(float x) {
#pragma clang fp contract(off)
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = fmaf(p, z, 2.06390887954e-2f);
    p = fmaf(p, z, -5.37397155531e-2f);
    p = fmaf(p, z, 1.33314422036e-1f);
    p = fmaf(p, z, -3.33332819422e-1f);
    const float small = fmaf(p * z, x, x);
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * TANH_PRESCALE) + 1.0f);
    return fabsf(x) < 0.625f ? small : fmaf(-2.0f, r, 1.0f);
}

__device__ __forceinline__ float tanh_prescaled(float xs) {  // xs = x * TANH_PRESCALE
#ifdef PPO_TANH_ACCURATE
    return tanh_accurate(xs * (1.0f / TANH_PRESCALE));
#else
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f);
    return fmaf(-2.0f, r, 1.0f);
#endif
}

__device__ __forceinline__ float tanh_f32(float x) { return tanh_prescaled(x * TANH_PRESCALE); }

// tanh of the Gaussian head's mean (PPO.py:55).  Unlike a hidden unit, whose error is averaged away
// by the next layer's dot product, an error d in the mean reaches the log-prob multiplied by
// (a - mu) / sigma^2 (up to ~5-30 for std 0.6-0.1): the head uses the accurate form (A <= 8 values
// per sample, against 128 hidden units).  -DPPO_HEAD_TANH_FAST selects the 4-instruction form.
__device__ __forceinline__ float tanh_head(float x) {
#ifdef PPO_HEAD_TANH_FAST
    return tanh_f32(x);
#else
    return tanh_accurate(x);
#endif
}

__device__ __forceinline__ f32x16 tanh16_prescaled(f32x16 v) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = tanh_prescaled(v[r]);
    return o;
}

// The same values with the add and the final fma written on register pairs (v_pk_add_f32 /
// v_pk_fma_f32: 16 vector instructions fewer per call).  Measured in the fused update kernel
// (same-box A/B): -0.5 % per epoch for observations of up to 15 values, +1.6 % at 26, so the caller
// chooses per shape class.
__device__ __forceinline__ f32x16 tanh16_prescaled_pk(f32x16 v) {
#ifdef PPO_TANH_ACCURATE
    return tanh16_prescaled(v);
#else
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 e = {__builtin_amdgcn_exp2f(v[r]), __builtin_amdgcn_exp2f(v[r + 1])};
        e = e + (f32x2){1.0f, 1.0f};
        const f32x2 rc = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
        const f32x2 t = __builtin_elementwise_fma(rc, (f32x2){-2.0f, -2.0f}, (f32x2){1.0f, 1.0f});
        o[r] = t.x;
        o[r + 1] = t.y;
    }
    return o;
#endif
}

// ------------------------------------------------------------------ LDS image of one MLP
// w1t: [2*KS1][64]   row k (< S: input k; == S: bias b1; > S: zero), column = hidden unit; x TANH_PRESCALE
// w2f: [64][W2_LD]   W2[out][in] x TANH_PRESCALE (as is b2).  Forward A operand: lane (c, h) reads
//                    four consecutive `in` of row `out` = c (+32) with one 16-byte load, which feeds
//                    four k-steps, because registers 4q..4q+3 of an accumulator tile are the four
//                    consecutive units 8q + 4h .. +3.
// w2b: [64][W2_LD]   W2^T, [in][out], unscaled (dgrad A operand, same access pattern); absent for
//                    forward-only kernels
// b2 : [64], w3: [AOUT][64] (rows >= A zero), b3: [AOUT]
// W2_LD = 68: rows 16-byte aligned, and eight consecutive rows start 4 banks apart, so a 16-byte
// access per lane is conflict-free.
#define W2_LD 68
struct NetLds {
    float* w1t;
    float* w2f;
    float* w2b;
    float* b2;
    float* w3;
    float* b3;
};

__host__ __device__ inline int ks1_of(int S) { return (S + 2) / 2; }  // ceil((S+1)/2) k-steps

__host__ __device__ inline int net_lds_floats(int S, int aout, bool bwd) {
    return 2 * ks1_of(S) * HID + HID * W2_LD * (bwd ? 2 : 1) + HID + aout * HID + ((aout + 3) & ~3);
}

__device__ inline NetLds carve_net_lds(float* base, int S, int aout, bool bwd) {
    NetLds L;
    L.w1t = base;
    L.w2f = L.w1t + 2 * ks1_of(S) * HID;
    L.w2b = bwd ? L.w2f + HID * W2_LD : nullptr;
    L.b2 = L.w2f + HID * W2_LD * (bwd ? 2 : 1);
    L.w3 = L.b2 + HID;
    L.b3 = L.w3 + aout * HID;
    return L;
}

// Block-wide fill; caller synchronises afterwards.  Every global load of the first pass (8 elements of W1
// and of W2 per thread - all of them for 512 threads - and the small tensors) is issued before the first
// LDS store, so that the fill costs one memory latency instead of one per tensor and trip (the weights
// were written by the previous launch: they come from beyond this XCD's L2).  BATCHED = false is the plain
// form, one tensor and one trip after the other (same image).
template <bool BATCHED = true>
__device__ inline void fill_net_lds(const NetLds& L, const float* __restrict__ params,
                                    const NetLayout& P, int S, int A, int aout, bool bwd) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n1 = 2 * ks1_of(S) * HID, n2 = HID * HID, n3 = aout * HID;
    auto w1_elem = [&](int i) {   // image row k (< S: input k; == S: bias b1; > S: zero), column = hidden unit
        const int k = i >> 6, o = i & 63;
        return k < S ? params[P.w1 + o * S + k] : (k == S ? params[P.b1 + o] : 0.f);
    };
    auto put_w1 = [&](int i, float v) { L.w1t[i] = v * TANH_PRESCALE; };
    auto put_w2 = [&](int i, float v) {   // W2[out = i>>6][in = i&63]
        L.w2f[(i >> 6) * W2_LD + (i & 63)] = v * TANH_PRESCALE;
        if (bwd) L.w2b[(i & 63) * W2_LD + (i >> 6)] = v;
    };
    auto w3_elem = [&](int i) { return (i >> 6) < A ? params[P.w3 + i] : 0.f; };
    if constexpr (!BATCHED) {
        for (int i = tid; i < n1; i += nt) put_w1(i, w1_elem(i));
        for (int i = tid; i < n2; i += nt) put_w2(i, params[P.w2 + i]);
        for (int i = tid; i < HID; i += nt) L.b2[i] = params[P.b2 + i] * TANH_PRESCALE;
        for (int i = tid; i < n3; i += nt) L.w3[i] = w3_elem(i);
        for (int i = tid; i < aout; i += nt) L.b3[i] = i < A ? params[P.b3 + i] : 0.f;
        return;
    }
    constexpr int U = 8;
    float r1[U], r2[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int i = tid + j * nt;
        r1[j] = i < n1 ? w1_elem(i) : 0.f;
        r2[j] = i < n2 ? params[P.w2 + i] : 0.f;
    }
    const float rb2 = tid < HID ? params[P.b2 + tid] : 0.f;
    const float r3 = tid < n3 ? w3_elem(tid) : 0.f;
    const float rb3 = tid < A ? params[P.b3 + tid] : 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int i = tid + j * nt;
        if (i < n1) put_w1(i, r1[j]);
        if (i < n2) put_w2(i, r2[j]);
    }
    if (tid < HID) L.b2[tid] = rb2 * TANH_PRESCALE;
    if (tid < n3) L.w3[tid] = r3;
    if (tid < aout) L.b3[tid] = rb3;
    // what the first pass did not reach (fewer than 512 threads)
    for (int i = tid + U * nt; i < n1; i += nt) put_w1(i, w1_elem(i));
    for (int i = tid + U * nt; i < n2; i += nt) put_w2(i, params[P.w2 + i]);
    for (int i = tid + nt; i < HID; i += nt) L.b2[i] = params[P.b2 + i] * TANH_PRESCALE;
    for (int i = tid + nt; i < n3; i += nt) L.w3[i] = w3_elem(i);
    for (int i = tid + nt; i < aout; i += nt) L.b3[i] = i < A ? params[P.b3 + i] : 0.f;
}

// Layer 2 of the forward pass: Z2 = W2 H1 + b2 (prescaled), tanh.  H1 stays in registers as the B
// operand; the A operand comes from w2f, one 16-byte LDS load per four k-steps and row half.
template <bool PK_TANH = false>
__device__ __forceinline__ void mlp_layer2(const NetLds& L, int c, int h, const f32x16 (&H1)[2],
                                           f32x16 (&H2)[2]) {
    f32x16 y0, y1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&L.b2[8 * q + 4 * h]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&L.b2[32 + 8 * q + 4 * h]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y0[4 * q + j] = b0[j];
            y1[4 * q + j] = b1[j];
        }
    }
    const float* row0 = L.w2f + c * W2_LD + 4 * h;
    const float* row1 = row0 + 32 * W2_LD;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(row0 + 32 * st + 8 * q);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(row1 + 32 * st + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = H1[st][4 * q + j];  // unit 32*st + 8q + 4h + j
                y0 = mfma32(a0[j], b, y0);
                y1 = mfma32(a1[j], b, y1);
            }
        }
    }
    H2[0] = PK_TANH ? tanh16_prescaled_pk(y0) : tanh16_prescaled(y0);
    H2[1] = PK_TANH ? tanh16_prescaled_pk(y1) : tanh16_prescaled(y1);
}

// ------------------------------------------------------------------ forward of one 32-sample tile
// x(k) returns input k (< S) of this lane's sample (0 for a lane without a sample).
template <class X>
__device__ __forceinline__ void mlp_fwd_tile_x(const NetLds& L, int S, const X& x, int c, int h,
                                               f32x16 (&H1)[2], f32x16 (&H2)[2]) {
    f32x16 z0 = {0}, z1 = {0};
    const int ks1 = ks1_of(S);
    for (int s = 0; s < ks1; ++s) {
        const int k = 2 * s + h;
        float b = (k == S) ? 1.0f : 0.0f;          // ones column carries the bias
        if (k < S) b = x(k);
        const float a0 = L.w1t[k * HID + c];
        const float a1 = L.w1t[k * HID + 32 + c];
        z0 = mfma32(a0, b, z0);
        z1 = mfma32(a1, b, z1);
    }
    H1[0] = tanh16_prescaled(z0);
    H1[1] = tanh16_prescaled(z1);
    mlp_layer2(L, c, h, H1, H2);
}

struct RowFromGlobal {  // states: [n, S] row-major; m = this lane's sample; valid = m < n
    const float* __restrict__ states;
    int S;
    int64_t m;
    bool valid;
    __device__ __forceinline__ float operator()(int k) const { return valid ? states[m * S + k] : 0.f; }
};

struct RowFromLds {  // this lane's sample row in an LDS staging tile
    const float* row;
    bool valid;
    __device__ __forceinline__ float operator()(int k) const { return valid ? row[k] : 0.f; }
};

__device__ __forceinline__ void mlp_fwd_tile(const NetLds& L, const float* __restrict__ states,
                                             int S, int64_t m, bool valid, int c, int h,
                                             f32x16 (&H1)[2], f32x16 (&H2)[2]) {
    const RowFromGlobal x{states, S, m, valid};
    mlp_fwd_tile_x(L, S, x, c, h, H1, H2);
}

// The value lane ^ 32 holds (v_permlane32_swap: a register move between the wave's halves, no LDS).
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);  // r[0]: lower half's values, r[1]: upper half's
}

// o[a] = b3[a] + sum_u W3[a][u] * H2[u]; each lane sums its 32 units as two chains of packed fmas
// (even and odd registers), the two halves are added.  (The update kernels' forward; act() uses
// out_layer_act.)
template <int AOUT>
__device__ __forceinline__ void out_layer(const NetLds& L, int A, const f32x16 (&H2)[2], int h,
                                          float (&o)[AOUT]) {
#pragma unroll
    for (int a = 0; a < AOUT; ++a) {
        float acc = 0.f;
        if (a < A) {
            f32x2 acc2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&L.w3[a * HID + 32 * t + 8 * q + 4 * h]);
                    acc2 = __builtin_elementwise_fma((f32x2){w.x, w.y}, (f32x2){H2[t][4 * q + 0], H2[t][4 * q + 1]}, acc2);
                    acc2 = __builtin_elementwise_fma((f32x2){w.z, w.w}, (f32x2){H2[t][4 * q + 2], H2[t][4 * q + 3]}, acc2);
                }
            }
            acc = acc2.x + acc2.y;
            acc += other_half(acc);
            acc += L.b3[a];
        }
        o[a] = acc;
    }
}

// The out layer of act(): the same sum taken as four partial sums P_g, one per unit class
// g = (unit bit 2) + 2 * (unit bit 0), each a chain of fmaf over the class's units in ascending order,
// then o = (P_0 + P_1) + (P_2 + P_3) + b3.  A lane of the 32-sample tile holds two classes (g = h and
// h + 2) and a lane of the 16-sample tile below holds one, so both tile shapes produce this sum bit
// for bit (two independent chains here instead of one, too).
template <int AOUT>
__device__ __forceinline__ void out_layer_act(const NetLds& L, int A, const f32x16 (&H2)[2], int h,
                                              float (&o)[AOUT]) {
#pragma unroll
    for (int a = 0; a < AOUT; ++a) {
        float pa = 0.f, pb = 0.f;
        if (a < A) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&L.w3[a * HID + 32 * t + 8 * q + 4 * h]);
                    pa = fmaf(w.x, H2[t][4 * q + 0], pa);
                    pb = fmaf(w.y, H2[t][4 * q + 1], pb);
                    pa = fmaf(w.z, H2[t][4 * q + 2], pa);
                    pb = fmaf(w.w, H2[t][4 * q + 3], pb);
                }
            }
            pa += other_half(pa);
            pb += other_half(pb);
            pa = (pa + pb) + L.b3[a];
        }
        o[a] = pa;
    }
}

// ------------------------------------------------------------------ forward on a 16-sample tile
// The same forward pass on v_mfma_f32_16x16x4_f32: a wave owns 16 samples, lane l holds sample
// c = l & 15, and g = l >> 4 selects the quarter of the hidden units its registers carry.  Used where
// a step's latency matters more than throughput (half the matrix and tanh work per wave and step).
// Both fp32 MFMA shapes add their k-slices one after the other, lower lane group first, each with
// one rounding (fmaf); the 32-sample tile's layer 2 therefore adds the units in the order
// 0,4,1,5 | 2,6,3,7 | 8,12,9,13 | ...  Here lane group g carries the units
//     u16(g, mb, r) = 16 mb + 8 (r >> 1) + 4 (g & 1) + 2 (r & 1) + (g >> 1)      (mb, r = 0..3),
// which makes register 4 mb + r of an activation directly the B operand of k-step 4 mb + r in that
// very order, and makes a lane's units one class of out_layer_act: every output bit equals the
// 32-sample tile's.
__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ inline int u16(int g, int mb, int r) {
    return 16 * mb + 8 * (r >> 1) + 4 * (g & 1) + 2 * (r & 1) + (g >> 1);
}

// LDS image of one MLP for the 16-sample tile (floats):
// w1 : [KS][4][64]     k-step s, output block mb, lane (c, g): W1s[u16(c >> 2, mb, c & 3)][k = 4 s + g]
//                      (k == S: the bias, k > S: 0), x TANH_PRESCALE
// w2 : [4][4][64][4]   output block mb, k-step group i4, lane, e: W2s[u16(c >> 2, mb, c & 3)][u16(g, i4, e)]
// b2 : [4][4][4]       [mb][g][r] = b2s[u16(g, mb, r)]
// w3 : [AOUT][4][4][4] [a][mb][g][r] = W3[a][u16(g, mb, r)] (rows >= A zero);  b3: [AOUT]
struct Net16 {
    float* w1;
    float* w2;
    float* b2;
    float* w3;
    float* b3;
};

__host__ __device__ inline int ks16_of(int S) { return (S + 4) / 4; }  // ceil((S+1)/4) k-steps

__host__ __device__ inline int net16_lds_floats(int S, int aout) {
    return ks16_of(S) * 256 + 4096 + 64 + aout * 64 + ((aout + 3) & ~3);
}

__device__ inline Net16 carve_net16(float* base, int S, int aout) {
    Net16 L;
    L.w1 = base;
    L.w2 = L.w1 + ks16_of(S) * 256;
    L.b2 = L.w2 + 4096;
    L.w3 = L.b2 + 64;
    L.b3 = L.w3 + aout * 64;
    return L;
}

// Block-wide fill; caller synchronises afterwards.
__device__ inline void fill_net16(const Net16& L, const float* __restrict__ params, const NetLayout& P,
                                  int S, int A, int aout) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < ks16_of(S) * 256; i += nt) {
        const int l = i & 63, mb = (i >> 6) & 3, s = i >> 8;
        const int uo = u16((l & 15) >> 2, mb, l & 3), k = 4 * s + (l >> 4);
        float v = 0.f;
        if (k < S) v = params[P.w1 + uo * S + k];
        else if (k == S) v = params[P.b1 + uo];
        L.w1[i] = v * TANH_PRESCALE;
    }
    for (int i = tid; i < 4096; i += nt) {
        const int e = i & 3, l = (i >> 2) & 63, i4 = (i >> 8) & 3, mb = i >> 10;
        const int uo = u16((l & 15) >> 2, mb, l & 3), k = u16(l >> 4, i4, e);
        L.w2[i] = params[P.w2 + uo * HID + k] * TANH_PRESCALE;
    }
    for (int i = tid; i < 64; i += nt)
        L.b2[i] = params[P.b2 + u16((i >> 2) & 3, i >> 4, i & 3)] * TANH_PRESCALE;
    for (int i = tid; i < aout * 64; i += nt) {
        const int a = i >> 6;
        L.w3[i] = a < A ? params[P.w3 + a * HID + u16((i >> 2) & 3, (i >> 4) & 3, i & 3)] : 0.f;
    }
    for (int i = tid; i < aout; i += nt) L.b3[i] = i < A ? params[P.b3 + i] : 0.f;
}

// The value lane ^ 16 holds (v_permlane16_swap: a register move between neighbouring 16-lane rows).
__device__ __forceinline__ float other_row(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);  // r[0]: even rows' values, r[1]: odd rows'
}

// Forward of one 16-sample tile up to the pre-head outputs o[]; x(k) as in mlp_fwd_tile_x.
template <int AOUT, class X>
__device__ __forceinline__ void mlp_fwd_out16(const Net16& L, int S, int A, const X& x, int lane,
                                              float (&o)[AOUT]) {
    const int g = lane >> 4;
    f32x4 z[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ks = ks16_of(S);
    for (int s = 0; s < ks; ++s) {
        const int k = 4 * s + g;
        float b = (k == S) ? 1.0f : 0.0f;   // ones column carries the bias
        if (k < S) b = x(k);
        const float* w = L.w1 + s * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) z[mb] = mfma16x4(w[64 * mb], b, z[mb]);
    }
    f32x4 H1[4], y[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) H1[mb][r] = tanh_prescaled(z[mb][r]);
        y[mb] = *reinterpret_cast<const f32x4*>(L.b2 + 16 * mb + 4 * g);
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        f32x4 a4[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            a4[mb] = *reinterpret_cast<const f32x4*>(L.w2 + ((mb * 4 + i4) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) y[mb] = mfma16x4(a4[mb][e], H1[i4][e], y[mb]);
        }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) y[mb][r] = tanh_prescaled(y[mb][r]);
    }
#pragma unroll
    for (int a = 0; a < AOUT; ++a) {
        float p = 0.f;
        if (a < A) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(L.w3 + ((a * 4 + mb) * 4 + g) * 4);
                p = fmaf(w.x, y[mb][0], p);
                p = fmaf(w.y, y[mb][1], p);
                p = fmaf(w.z, y[mb][2], p);
                p = fmaf(w.w, y[mb][3], p);
            }
            p += other_row(p);     // P_g + P_(g^1): P_0 + P_1 on rows 0-1, P_2 + P_3 on rows 2-3
            p += other_half(p);
            p += L.b3[a];
        }
        o[a] = p;
    }
}

// ------------------------------------------------------------------ either tile shape behind one name
// One MLP's LDS image and forward for either tile shape.
template <int TM> struct TileNet;
template <> struct TileNet<32> {
    NetLds L;
    static __host__ __device__ int floats(int S, int aout) { return (net_lds_floats(S, aout, false) + 3) & ~3; }
    __device__ void stage(float* base, const float* params, const NetLayout& P, int S, int A, int aout) {
        L = carve_net_lds(base, S, aout, false);
        fill_net_lds(L, params, P, S, A, aout, false);
    }
    template <int AOUT, class X>
    __device__ __forceinline__ void forward(int S, int A, const X& x, int lane, float (&o)[AOUT]) const {
        f32x16 H1[2], H2[2];
        mlp_fwd_tile_x(L, S, x, lane & 31, lane >> 5, H1, H2);
        out_layer_act<AOUT>(L, A, H2, lane >> 5, o);
    }
};
template <> struct TileNet<16> {
    Net16 L;
    static __host__ __device__ int floats(int S, int aout) { return (net16_lds_floats(S, aout) + 3) & ~3; }
    __device__ void stage(float* base, const float* params, const NetLayout& P, int S, int A, int aout) {
        L = carve_net16(base, S, aout);
        fill_net16(L, params, P, S, A, aout);
    }
    template <int AOUT, class X>
    __device__ __forceinline__ void forward(int S, int A, const X& x, int lane, float (&o)[AOUT]) const {
        mlp_fwd_out16<AOUT>(L, S, A, x, lane, o);
    }
};

// ------------------------------------------------------------------ heads
// The head functions below switch floating-point contraction off: every multiply and add is
// rounded as written, so the same pre-head outputs give bit-identical log-probs in every kernel that
// inlines them, whatever the surrounding code (act, rollout and evaluate kernels also form those
// outputs in one order, out_layer_act; the update kernels' out_layer sums in another: equal to rounding).
// exp / log / divide of the distribution heads.  The heads are one dependent chain per sample (no
// instruction-level parallelism to hide latencies), so they use the hardware's base-2 exp / log and
// reciprocal (<= 1 ulp each) instead of the ~10-instruction correctly rounded library sequences;
// -DPPO_HEAD_EXACT_MATH restores those.
#ifdef PPO_HEAD_EXACT_MATH
__device__ __forceinline__ float head_exp(float x) { return expf(x); }
__device__ __forceinline__ float head_log(float x) { return logf(x); }
__device__ __forceinline__ float head_div(float a, float b) { return a / b; }
#else
__device__ __forceinline__ float head_exp(float x) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
__device__ __forceinline__ float head_log(float x) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
}
__device__ __forceinline__ float head_div(float a, float b) {
#pragma clang fp contract(off)
    return a * __builtin_amdgcn_rcpf(b);
}
#endif

// Categorical over softmax probabilities, restating torch/distributions/categorical.py:71
// (renormalise), utils.py:121-137 (logits = log(clamp(probs, eps, 1-eps))), :150-161.
template <int AOUT>
struct CatOut {
    float q[AOUT];     // renormalised probabilities
    float L[AOUT];     // clamped log-probabilities
    float pass[AOUT];  // 1 where the clamp passes gradient (closed interval), else 0
    float ent;
};

template <int AOUT>
__device__ __forceinline__ void categorical_head(const float (&o)[AOUT], int A, CatOut<AOUT>& out) {
#pragma clang fp contract(off)
    float mx = o[0];
#pragma unroll
    for (int a = 1; a < AOUT; ++a) if (a < A) mx = fmaxf(mx, o[a]);
    float e[AOUT], s = 0.f;
#pragma unroll
    for (int a = 0; a < AOUT; ++a) { e[a] = a < A ? head_exp(o[a] - mx) : 0.f; s += e[a]; }
    float s2 = 0.f;
#pragma unroll
    for (int a = 0; a < AOUT; ++a) { e[a] = head_div(e[a], s); s2 += e[a]; }
    float ent = 0.f;
#pragma unroll
    for (int a = 0; a < AOUT; ++a) {
        const float q = head_div(e[a], s2);
        const float qc = fminf(fmaxf(q, EPS_PROB_F), 1.0f - EPS_PROB_F);
        const float L = head_log(qc);
        out.q[a] = q;
        out.L[a] = L;
        out.pass[a] = (q >= EPS_PROB_F && q <= 1.0f - EPS_PROB_F) ? 1.f : 0.f;
        if (a < A) ent += L * q;
    }
    out.ent = -ent;
}

// Diagonal Gaussian around tanh(o): torch/distributions/multivariate_normal.py:255-273 with
// scale_tril = diag(sqrt(var)) (:193).
template <int AOUT>
struct GaussOut {
    float mu[AOUT];
    float zed[AOUT];  // (a - mu) / sigma
    float logp, ent, sigma;
};

template <int AOUT>
__device__ __forceinline__ void gaussian_head(const float (&o)[AOUT], int A, float action_var,
                                              const float (&act)[AOUT], GaussOut<AOUT>& out) {
#pragma clang fp contract(off)
    const float sigma = sqrtf(action_var);
    const float ls = logf(sigma);
    float M = 0.f, hld = 0.f;
#pragma unroll
    for (int a = 0; a < AOUT; ++a) {
        if (a < A) {
            const float mu = tanh_head(o[a]);
            const float z = head_div(act[a] - mu, sigma);
            out.mu[a] = mu;
            out.zed[a] = z;
            M += z * z;
            hld += ls;
        } else {
            out.mu[a] = 0.f;
            out.zed[a] = 0.f;
        }
    }
    out.sigma = sigma;
    out.logp = -0.5f * ((float)A * LOG_2PI_F + M) - hld;
    out.ent = 0.5f * (float)A * (1.0f + LOG_2PI_F) + hld;
}

// ------------------------------------------------------------------ Philox4x32-10
// 128-bit counter = (index: 64 bits | stream: 56 bits | block: 8 bits).  `index` is the sample / env,
// `stream` the step (or episode) number, and successive next() calls of one generator advance only
// `block`: the draws of different (stream, index) pairs never share a counter value, however many
// blocks one of them consumes (at most 17 here: one block + 16 for a 63-dimensional observation).
struct Philox {
    uint32_t c[4], k[2];
    __device__ Philox(uint64_t seed, uint64_t stream, uint64_t index) {
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        c[0] = (uint32_t)index; c[1] = (uint32_t)(index >> 32);
        c[2] = (uint32_t)stream; c[3] = (uint32_t)(stream >> 32) << 8;
    }
    __device__ void next(uint32_t (&out)[4]) {
        uint32_t x0 = c[0], x1 = c[1], x2 = c[2], x3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0;
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0;
            const uint32_t y1 = (uint32_t)p1;
            const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1;
            const uint32_t y3 = (uint32_t)p0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
        ++c[3];  // next block of this (stream, index)
    }
    // continue with block b (< 256) of this (stream, index): blocks are independent draws
    __device__ void set_block(uint32_t b) { c[3] = (c[3] & 0xFFFFFF00u) | b; }
};

__device__ __forceinline__ float u01(uint32_t x) {  // (0, 1]
#pragma clang fp contract(off)
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Two standard normals from two 32-bit words (Box-Muller).  The draws are noise, not a parity
// target (the reference's come from torch's generator), so the hardware's base-2 log, square root
// and sine / cosine of a fraction of a turn (<= 1 ulp / ~1e-6 absolute) replace the correctly
// rounded library routines: 8 instructions instead of ~170, which is what bounded the rollout
// kernel at wide observations and the Gaussian head's single-state latency.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
#pragma clang fp contract(off)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(a)));  // sqrt(-2 ln u)
    const float turn = u01(b);                                                                       // angle / 2 pi
    n0 = r * __builtin_amdgcn_cosf(turn);
    n1 = r * __builtin_amdgcn_sinf(turn);
}

// ------------------------------------------------------------------ policy head of act()
// From the actor's pre-head outputs o[] of sample m: sample an action (SAMPLE) or read the given one,
// and return its log-prob and the entropy (ActorCritic.act / evaluate, PPO.py:86-121).  Both
// half-waves hold the same sample and compute the same values; the h == 0 lane stores the action.
// action_out (discrete) / act[] (continuous) return the action to the caller as well.
template <bool CONT, int AMAX, bool SAMPLE>
__device__ __forceinline__ void policy_head_act(const float (&o)[AMAX], int A, float action_var,
                                                uint64_t seed, uint64_t rng_offset, int64_t m,
                                                bool valid, int h, void* __restrict__ actions_io,
                                                float& logp, float& ent, int& action_out,
                                                float (&act)[AMAX]) {
#pragma clang fp contract(off)
    action_out = 0;
    if constexpr (CONT) {
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
        action_out = action;
    }
}

// ------------------------------------------------------------------ synthetic env draws
// Env e at step `step` draws from Philox(seed ^ C, step, e): block 0 -> reward (first normal) and
// terminal (third word); block 1 + j -> observation entries 4j .. 4j+3 (standard normals).
#define SYNTH_ENV_KEY 0x9E3779B97F4A7C15ull

// Block b of env e at `step`: its four normals and its third word.  One instruction stream for
// every block, so lanes that draw different blocks of one env run side by side.
__device__ __forceinline__ void synth_env_block(uint64_t seed, int64_t step, int64_t e, int b,
                                                float (&v)[4], uint32_t& word2) {
    Philox rng(seed ^ SYNTH_ENV_KEY, (uint64_t)step, (uint64_t)e);
    rng.set_block((uint32_t)b);
    uint32_t u[4];
    rng.next(u);
    box_muller(u[0], u[1], v[0], v[1]);
    box_muller(u[2], u[3], v[2], v[3]);
    word2 = u[2];
}

__device__ __forceinline__ void synth_reward_terminal_of(const float (&v)[4], uint32_t word2, int64_t step,
                                                         int reward_kind, float p_terminal,
                                                         int64_t horizon, float& reward, bool& term) {
    reward = reward_kind == 0 ? 1.0f : v[0];
    if (p_terminal < 0.f) term = (step % horizon) == horizon - 1;
    else term = (1.0f - u01(word2)) < p_terminal;
}

__device__ __forceinline__ void synth_env_reward_terminal(uint64_t seed, int64_t step, int64_t e,
                                                          int reward_kind, float p_terminal,
                                                          int64_t horizon, float& reward, bool& term) {
    float v[4];
    uint32_t w2;
    synth_env_block(seed, step, e, 0, v, w2);
    synth_reward_terminal_of(v, w2, step, reward_kind, p_terminal, horizon, reward, term);
}

__device__ __forceinline__ void synth_env_state_block(uint64_t seed, int64_t step, int64_t e, int j,
                                                      float (&v)[4]) {
    uint32_t w2;
    synth_env_block(seed, step, e, 1 + j, v, w2);
}

// ------------------------------------------------------------------ CartPole-v1 dynamics
// One Euler step of the published cart-pole equations (tau = 0.02 s, force +-10 N) on the fp64
// state (x, x_dot, theta, theta_dot); returns true when the pole fell or the cart left the track.
__device__ __forceinline__ bool cartpole_advance(double (&s)[4], int action) {
#pragma clang fp contract(off)
    const double gravity = 9.8, masspole = 0.1, total_mass = 1.1, length = 0.5,
                 polemass_length = 0.05, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    double x = s[0], xd = s[1], th = s[2], thd = s[3];
    const double force = action == 1 ? force_mag : -force_mag;
    const double ct = cos(th), st = sin(th);
    const double temp = (force + polemass_length * (thd * thd) * st) / total_mass;
    const double thacc = (gravity * st - ct * temp) /
                         (length * (4.0 / 3.0 - masspole * (ct * ct) / total_mass));
    const double xacc = temp - polemass_length * thacc * ct / total_mass;
    x = x + tau * xd;
    xd = xd + tau * xacc;
    th = th + tau * thd;
    thd = thd + tau * thacc;
    s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
    return x < -x_thr || x > x_thr || th < -theta_thr || th > theta_thr;
}

// reset draw of env e for its episode number ep: four values U(-0.05, 0.05)
#define CARTPOLE_RESET_KEY 0xC2B2AE3D27D4EB4Full
__device__ __forceinline__ void cartpole_reset_draw(uint64_t seed, unsigned ep, int64_t e, double (&s)[4]) {
    Philox rng(seed ^ CARTPOLE_RESET_KEY, (uint64_t)ep, (uint64_t)e);
    uint32_t u[4];
    rng.next(u);
#pragma unroll
    for (int k = 0; k < 4; ++k)   // 32 random bits -> [0, 1) exactly representable in fp64
        s[k] = ((double)u[k] * (1.0 / 4294967296.0) - 0.5) * 0.1;
}
