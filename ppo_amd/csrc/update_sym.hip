// Fused evaluate + clipped-surrogate loss + backward, SYMMETRIC-wave form (opt-in alternative to the
// forward/backward-pair kernel of update.hip, for state_dim <= 31; PPO_FWDBWD=sym): every wave runs the whole forward AND backward pass of its own tiles in one
// instruction stream, and there is no workgroup barrier inside the tile loop.
//
// Why: on gfx950 a wave with an fp32 MFMA ready always wins the SIMD's issue slot over a wave with a
// VALU instruction ready.  With dedicated forward / backward waves coupled by barriers (update.hip)
// every latency of the wave that runs alone is exposed; with two free-running waves of the SAME mixed
// stream per SIMD, whenever one waits (LDS operands, MFMA results, memory) the other issues.  Measured
// on the instruction mix of this network (tools/ubench/symmetric.hip): 14.6 k cycles per 32 samples
// and SIMD against 18.7 k.
//
// A 32-sample tile does not fit that form (two waves per SIMD leave 256 registers and 20 KB of LDS
// per wave), so the tile is 16 samples on v_mfma_f32_16x16x4_f32:
//   lane l = (j = l & 15, q = l >> 4).  A[i][k]: lane supplies A[i = j][k = q]; B[k][n]: lane supplies
//   B[k = q][n = j]; D[i][n]: register r of lane (j, q) is D[i = 4q + r][n = j].
// Orientation: the sample is the MFMA column n, hidden units are rows.  A 64-unit activation vector
// of the 16 samples is four accumulators (f32x4 v[4]): v[t][r] on lane (j, q) is unit 16t + 4q + r of
// sample j.  A k-step takes B[k = q][n = j], so register v[t][r] IS the B operand of the next layer's
// k-step over units {16t + 4q + r : q = 0..3}: forward and dgrad chain through registers.  The matching
// A operands of the four k-steps r = 0..3 are four consecutive weights W[out][16t + 4q .. + 3]: one
// 16-byte LDS load feeds four MFMAs.
// LDS images are laid out for those loads: the 16 x 16 block (row tile, k tile) of a weight matrix is
// 256 consecutive floats with element (i, 4q + r) at 64q + 4i + r, so that lane (i, q) reads 16 bytes
// at 4 * lane - conflict-free by construction.  Weight gradients reduce over samples, so their
// operands pass through a wave-private staging image [unit][sample] with the same block shape,
// element (unit i, sample 4q + s) at 64q + 4(i ^ q) + s: written with 4-byte stores from the
// accumulator layout and read back with 16-byte loads, both conflict-free; four consecutive samples
// are the four k-steps of one load.
#include "ppo_common.h"
#include "update_common.h"

#define SYM_WAVES 8
#define SYM_THREADS (SYM_WAVES * 64)
#define SYM_TILE 16
#define W1_LD 80    // row stride of the layer-1 image: lanes q and q + 1 land 16 banks apart

// Diagnostic build only (-DPPO_PHASE_STAMPS): per-phase s_memtime sums, one row of 16 counters per
// wave, in the buffer registered with ppo_debug_set_stamps().  Never compiled into the product.
#ifdef PPO_PHASE_STAMPS
#define SSTAMP_DECL unsigned long long st_acc[12] = {0}; unsigned long long st_prev = 0;
#define SSTAMP_START { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define SSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[i] += _t - st_prev; st_prev = _t; __builtin_amdgcn_sched_barrier(0); }
#define SSTAMP_FLUSH(a, wave) if ((a).stamps && (threadIdx.x & 63) == 0) { unsigned long long* row = (a).stamps + ((size_t)blockIdx.x * 8 + (wave)) * 16; for (int i = 0; i < 12; ++i) row[i] = st_acc[i]; }
#else
#define SSTAMP_DECL
#define SSTAMP_START
#define SSTAMP(i)
#define SSTAMP_FLUSH(a, wave)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// tanh of four prescaled pre-activations: the two transcendentals per value are scalar, the add and
// the final fma are written on the whole vector so that they become packed fp32 instructions
// (v_pk_add_f32 / v_pk_fma_f32: two values per issue slot).  Same arithmetic as tanh_prescaled().
__device__ __forceinline__ f32x4 tanh4_prescaled(f32x4 xs) {
#ifdef PPO_TANH_ACCURATE
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = tanh_prescaled(xs[r]);
    return o;
#else
    f32x4 e, rc;
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(xs[r]);
    e = e + 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rc[r] = __builtin_amdgcn_rcpf(e[r]);
    return __builtin_elementwise_fma(rc, (f32x4)(-2.0f), (f32x4)(1.0f));
#endif
}

__host__ __device__ inline int sym_k1(int S) { return ((S + 4) / 4) * 4; }   // S + 1 inputs padded to whole k-steps
__host__ __device__ inline int sym_image_floats(int S, int aout) {
    return sym_k1(S) * W1_LD + 2 * HID * HID + HID + aout * HID + ((aout + 3) & ~3);
}
// per wave: dZ image, H1 image (1024 floats each), X tile [16][16 CT + 1], dout^T [16][16]
__host__ __device__ inline int sym_xld(int ct) { return 16 * ct + 1; }
__host__ __device__ inline int sym_wave_floats(int ct) { return 2 * 1024 + ((16 * sym_xld(ct) + 3) & ~3) + 256; }

struct SymImages {
    float* w1;    // [k1][W1_LD]: row k (< S: input k; == S: b1; beyond: 0), column = unit; x TANH_PRESCALE
    float* w2f;   // blocks (out tile, in tile): W2[out][in] x TANH_PRESCALE, forward A operand
    float* w2b;   // blocks (in tile, out tile): W2[out][in], dgrad A operand
    float* b2;    // [64] x TANH_PRESCALE
    float* w3;    // [AOUT][64], rows >= A zero
    float* b3;    // [AOUT]
};

__device__ inline SymImages sym_carve(float* base, int S, int aout) {
    SymImages I;
    I.w1 = base;
    I.w2f = I.w1 + sym_k1(S) * W1_LD;
    I.w2b = I.w2f + HID * HID;
    I.b2 = I.w2b + HID * HID;
    I.w3 = I.b2 + HID;
    I.b3 = I.w3 + aout * HID;
    return I;
}

__device__ inline void sym_fill(const SymImages& I, const float* __restrict__ params, const NetLayout& P,
                                int S, int A, int aout) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int k1 = sym_k1(S);
    for (int i = tid; i < k1 * W1_LD; i += nt) {
        const int k = i / W1_LD, o = i - k * W1_LD;
        float v = 0.f;
        if (o < HID) {
            if (k < S) v = params[P.w1 + o * S + k];
            else if (k == S) v = params[P.b1 + o];
        }
        I.w1[i] = v * TANH_PRESCALE;
    }
    for (int i = tid; i < HID * HID; i += nt) {
        const int out = i >> 6, in = i & 63;
        const float v = params[P.w2 + i];
        I.w2f[((out >> 4) * 4 + (in >> 4)) * 256 + 64 * ((in >> 2) & 3) + 4 * (out & 15) + (in & 3)] = v * TANH_PRESCALE;
        I.w2b[((in >> 4) * 4 + (out >> 4)) * 256 + 64 * ((out >> 2) & 3) + 4 * (in & 15) + (out & 3)] = v;
    }
    for (int i = tid; i < HID; i += nt) I.b2[i] = params[P.b2 + i] * TANH_PRESCALE;
    for (int i = tid; i < aout * HID; i += nt) I.w3[i] = (i >> 6) < A ? params[P.w3 + i] : 0.f;
    for (int i = tid; i < aout; i += nt) I.b3[i] = i < A ? params[P.b3 + i] : 0.f;
}

__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }

// staging image address of (unit 16t + i, sample m): block t, element (i, m) at 64 (m >> 2) + 4 (i ^ (m >> 2)) + (m & 3)
__device__ __forceinline__ int stg_write_off(int t, int q, int r, int j) {
    return t * 256 + 64 * (j >> 2) + 4 * ((4 * q + r) ^ (j >> 2)) + (j & 3);
}

template <int HEAD, int AOUT, int CT>
__device__ __forceinline__ void sym_role(const FwdBwdArgs& a, float* smem) {
    constexpr bool L3_VALU = AOUT <= 2;   // output-layer wgrad on the VALU; wider heads use the MFMA
    constexpr int NU = 4 * CT;            // observation values a lane moves per tile (S <= 16 CT - 1)
    const int S = a.S;
    const int A = HEAD == HEAD_VALUE ? 1 : a.A;
    const int base = HEAD == HEAD_VALUE ? n_actor_params(a.S, a.A) : 0;
    const NetLayout P = net_layout(base, S, A);
    const bool critic = HEAD == HEAD_VALUE;
    const int wg = critic ? blockIdx.x - a.n_actor_wg : blockIdx.x;
    const int nwg = critic ? gridDim.x - a.n_actor_wg : a.n_actor_wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    SSTAMP_DECL
    SSTAMP_START
    const SymImages I = sym_carve(smem, S, AOUT);
    const int XLD = sym_xld(CT);
    float* const wv = smem + ((sym_image_floats(S, AOUT) + 3) & ~3) + wave * sym_wave_floats(CT);
    float* const stg_z = wv;            // dZ2 image, later dZ1 (and H2 while the MFMA output-layer wgrad runs)
    float* const stg_h = wv + 1024;     // H1 image
    float* const xs = wv + 2048;        // X tile [16][XLD]: inputs, 1 at column S, 0 beyond
    float* const dts = xs + ((16 * XLD + 3) & ~3);   // dout^T [16 rows a][16 samples], rows >= AOUT zero
    sym_fill(I, a.params, P, S, A, AOUT);
    for (int i = lane; i < 16 * XLD; i += 64) xs[i] = (i % XLD) == S ? 1.0f : 0.0f;
    for (int i = lane; i < 256; i += 64) dts[i] = 0.f;
    __syncthreads();

    // persistent gradient accumulators
    f32x4 gW2[4][4], gW1[4][CT], gb2[4], gW3v[L3_VALU ? AOUT : 1][4], gW3m[4];
    float gb3[AOUT];
    double loss_acc = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) gW2[t][u] = 0.f;
#pragma unroll
        for (int u = 0; u < CT; ++u) gW1[t][u] = 0.f;
        gb2[t] = 0.f;
        gW3m[t] = 0.f;
#pragma unroll
        for (int k = 0; k < (L3_VALU ? AOUT : 1); ++k) gW3v[k][t] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < AOUT; ++k) gb3[k] = 0.f;

    const int64_t n_tiles = (a.n + SYM_TILE - 1) / SYM_TILE;
    const int64_t tstride = (int64_t)nwg * SYM_WAVES;
    const int64_t tile0 = (int64_t)wg * SYM_WAVES + wave;

    // observation prefetch: lane (row = lane >> 2, c0 = lane & 3) moves columns c0, c0 + 4, ...
    const int xrow = lane >> 2, xc0 = lane & 3;
    float xpre[NU];
    auto prefetch = [&](int64_t tile) {
        const int64_t mi = min(tile * SYM_TILE + xrow, a.n - 1);
        const float* rowp = a.states + mi * S + xc0;
#pragma unroll
        for (int u = 0; u < NU; ++u) xpre[u] = (4 * u + xc0 < S) ? rowp[4 * u] : 0.f;
    };
    if (tile0 < n_tiles) prefetch(tile0);
    // Waves w and w + 4 share a SIMD and would otherwise run the same phases in lockstep (both
    // waiting for LDS / MFMA results at the same moments): the second half of the workgroup starts
    // about half a tile later, and with no barrier in the loop the offset persists.
#ifndef SYM_STAGGER
#define SYM_STAGGER 0
#endif
#ifndef SYM_PRIO_MODE
#define SYM_PRIO_MODE 1
#endif
    if (SYM_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(SYM_STAGGER);

    // loop-invariant LDS addresses of this lane
    const float* const w2f_l = I.w2f + 4 * lane;              // + (t * 4 + t') * 256
    const float* const w2b_l = I.w2b + 4 * lane;
    const float* const zrd = stg_z + 64 * q + 4 * (j ^ q);     // + t * 256: 16-byte read of 4 samples
    const float* const hrd = stg_h + 64 * q + 4 * (j ^ q);

    SSTAMP(8)
    int it = wave >> 2;
    for (int64_t tile = tile0; tile < n_tiles; tile += tstride, ++it) {
        const int64_t m = tile * SYM_TILE + j;
        const bool valid = m < a.n;
        // Issue arbitration between the two waves of a SIMD is by priority, then age: at equal
        // priority the older wave (0-3) always wins and finishes its tiles ~20 % sooner, leaving its
        // partner alone on the SIMD at the end.  Alternating the priority tile by tile, in opposite
        // phase for the two halves of the workgroup, evens them out.
#if SYM_PRIO_MODE == 1
        if (it & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#elif SYM_PRIO_MODE == 2
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
        SSTAMP_START
        // ---------------- observations of this tile into the wave's X tile; next tile's on their way
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (4 * u + xc0 < S) xs[xrow * XLD + 4 * u + xc0] = xpre[u];
        if (tile + tstride < n_tiles) prefetch(tile + tstride);
        const HeadIn<HEAD, AOUT> hin = load_head_inputs<HEAD, AOUT>(a, A, m, valid);
        __builtin_amdgcn_wave_barrier();
        SSTAMP(0)

        // ---------------- layer 1 (bias rides as the ones column of X)
        f32x4 H1[4], H2[4];
        {
            f32x4 z[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) z[t] = 0.f;
            int ks1 = sym_k1(S) >> 2;
            asm volatile("" : "+s"(ks1));
#pragma unroll
            for (int s = 0; s < 4 * CT; ++s) {
                if (s < ks1) {
                    const float b = xs[j * XLD + 4 * s + q];
                    const float* wr = I.w1 + (4 * s + q) * W1_LD + j;
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = mfma16(wr[16 * t], b, z[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) H1[t] = tanh4_prescaled(z[t]);
        }
        SSTAMP(1)
        // ---------------- layer 2
        {
            f32x4 y[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) y[t] = *reinterpret_cast<const f32x4*>(&I.b2[16 * t + 4 * q]);
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                f32x4 a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = *reinterpret_cast<const f32x4*>(w2f_l + (t * 4 + tk) * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t] = mfma16(a4[t][r], H1[tk][r], y[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) H2[t] = tanh4_prescaled(y[t]);
        }
        SSTAMP(2)
        // ---------------- output layer: each lane sums its 16 units, the four lanes of a sample are added
        float o[AOUT];
#pragma unroll
        for (int k = 0; k < AOUT; ++k) {
            float acc = 0.f;
            if (k < A) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&I.w3[k * HID + 16 * t + 4 * q]);
                    acc = fmaf(w.x, H2[t][0], acc);
                    acc = fmaf(w.y, H2[t][1], acc);
                    acc = fmaf(w.z, H2[t][2], acc);
                    acc = fmaf(w.w, H2[t][3], acc);
                }
                acc += xor16(acc);
                acc += xor32(acc);
                acc += I.b3[k];
            }
            o[k] = acc;
        }
        // ---------------- head, loss, gradient w.r.t. the pre-head outputs (same on the 4 lanes of a sample)
        float dout[AOUT];
        head_loss_grad<HEAD, AOUT>(a, A, o, hin, m, valid, q == 0, dout, loss_acc);
#pragma unroll
        for (int k = 0; k < AOUT; ++k) gb3[k] += dout[k];

        SSTAMP(3)
        // ---------------- output-layer weight gradient dW3[k, u] += dout[m, k] H2[m, u]
        if constexpr (L3_VALU) {
#pragma unroll
            for (int k = 0; k < AOUT; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) gW3v[k][t] += dout[k] * H2[t];
        } else {
            // on the MFMA: A = dout^T [k][sample], B = H2 [sample][unit] through the staging images
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg_z[stg_write_off(t, q, r, j)] = H2[t][r];
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < AOUT; ++k) dts[k * 16 + j] = dout[k];
            }
            __builtin_amdgcn_wave_barrier();
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(dts + j * 16 + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(zrd + t * 256);
#pragma unroll
                for (int s = 0; s < 4; ++s) gW3m[t] = mfma16(a4[s], b4[s], gW3m[t]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---------------- output layer backward: dZ2 = (W3^T dout) (1 - H2^2); staged with H1 for wgrad 2
        f32x4 dZ[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 g = 0.f;
#pragma unroll
            for (int k = 0; k < AOUT; ++k) {
                if (k < A) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&I.w3[k * HID + 16 * t + 4 * q]);
                    g += w * dout[k];
                }
            }
            dZ[t] = g * (1.0f - H2[t] * H2[t]);
            gb2[t] += dZ[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                stg_z[stg_write_off(t, q, r, j)] = dZ[t][r];
                stg_h[stg_write_off(t, q, r, j)] = H1[t][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        SSTAMP(4)
        // ---------------- dgrad of layer 2: dH1 = W2^T dZ2 (B operand straight from registers), through tanh
        f32x4 d[4];
        {
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = 0.f;
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                f32x4 a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = *reinterpret_cast<const f32x4*>(w2b_l + (t * 4 + tk) * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = mfma16(a4[t][r], dZ[tk][r], d[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = d[t] * (1.0f - H1[t] * H1[t]);
        }
        SSTAMP(5)
        // ---------------- wgrad of layer 2: dW2[o, i] += sum_m dZ2[m, o] H1[m, i]
        {
            f32x4 a4[4], b4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a4[t] = *reinterpret_cast<const f32x4*>(zrd + t * 256);
                b4[t] = *reinterpret_cast<const f32x4*>(hrd + t * 256);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int to = 0; to < 4; ++to)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) gW2[to][ti] = mfma16(a4[to][s], b4[ti][s], gW2[to][ti]);
        }
        __builtin_amdgcn_wave_barrier();
        SSTAMP(6)
        // ---------------- dZ1 takes dZ2's place; wgrad of layer 1 (+ bias as the ones column)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg_z[stg_write_off(t, q, r, j)] = d[t][r];
        __builtin_amdgcn_wave_barrier();
        {
            f32x4 a4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[t] = *reinterpret_cast<const f32x4*>(zrd + t * 256);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float b = xs[(4 * q + s) * XLD + 16 * ct + j];
#pragma unroll
                    for (int to = 0; to < 4; ++to) gW1[to][ct] = mfma16(a4[to][s], b, gW1[to][ct]);
                }
        }
        __builtin_amdgcn_wave_barrier();
        SSTAMP(7)
    }
    SSTAMP_START

    // ---------------- per-lane partials -> per-unit sums (over the 16 samples j of a lane group)
    auto jsum = [&](float v) {
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        return v;
    };
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gb2[t][r] = jsum(gb2[t][r]);
#pragma unroll
            for (int k = 0; k < (L3_VALU ? AOUT : 1); ++k) gW3v[k][t][r] = jsum(gW3v[k][t][r]);
        }
#pragma unroll
    for (int k = 0; k < AOUT; ++k) gb3[k] = jsum(gb3[k]);
    {
        double l = loss_acc;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) l += __shfl_xor(l, off);
        loss_acc = l;
    }
    __syncthreads();   // every wave is done with the images: LDS is free for the reduction arrays

    // Workgroup reduction in fixed order: four parameter-indexed arrays; waves 0-3 store their sums,
    // then waves 4-7 add theirs to the array of wave - 4; then all threads add the four arrays.
    const int red_stride = P.n + 1;
    float* const red = smem + (wave & 3) * red_stride;
    const int o_w1 = P.w1 - base, o_b1 = P.b1 - base, o_w2 = P.w2 - base, o_b2 = P.b2 - base,
              o_w3 = P.w3 - base, o_b3 = P.b3 - base, o_loss = P.n;
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        if ((wave >> 2) == round) {
            const bool add = round == 1;
            auto put = [&](int idx, float v) { red[idx] = add ? red[idx] + v : v; };
#pragma unroll
            for (int to = 0; to < 4; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int out = 16 * to + 4 * q + r;
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) put(o_w2 + out * HID + 16 * ti + j, gW2[to][ti][r]);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int in = 16 * ct + j;
                        if (in < S) put(o_w1 + out * S + in, gW1[to][ct][r]);
                        else if (in == S) put(o_b1 + out, gW1[to][ct][r]);
                    }
                    if (j == 0) {
                        put(o_b2 + out, gb2[to][r]);
                        if constexpr (L3_VALU) {
#pragma unroll
                            for (int k = 0; k < AOUT; ++k)
                                if (k < A) put(o_w3 + k * HID + out, gW3v[k][to][r]);
                        }
                    }
                }
            if constexpr (!L3_VALU) {
                // D of the output-layer wgrad: row (output) = 4q + r, column (unit) = 16 ti + j
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * q + r;
                        if (k < A) put(o_w3 + k * HID + 16 * ti + j, gW3m[ti][r]);
                    }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < AOUT; ++k)
                    if (k < A) put(o_b3 + k, gb3[k]);
                const float lv = HEAD == HEAD_VALUE ? (float)(0.5 * loss_acc * (double)a.inv_n)
                                                    : (float)(loss_acc * (double)a.inv_n);
                put(o_loss, lv);
            }
        }
        __syncthreads();
    }
    float* dst = HEAD == HEAD_VALUE ? a.partial_critic + (size_t)wg * a.stride_critic
                                    : a.partial_actor + (size_t)wg * a.stride_actor;
    for (int i = threadIdx.x; i < red_stride; i += SYM_THREADS)
        dst[i] = ((smem[i] + smem[red_stride + i]) + smem[2 * red_stride + i]) + smem[3 * red_stride + i];
    SSTAMP(9)
    SSTAMP_FLUSH(a, wave)
}

template <bool CONT, int AMAX, int CT>
__global__ __launch_bounds__(SYM_THREADS, 2) void fwdbwd_sym_kernel(const FwdBwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.ws_header[0] = a.n_actor_wg;
        a.ws_header[1] = (int)gridDim.x - a.n_actor_wg;
    }
    if ((int)blockIdx.x < a.n_actor_wg)
        sym_role<CONT ? HEAD_GAUSSIAN : HEAD_CATEGORICAL, AMAX, CT>(a, smem);
    else
        sym_role<HEAD_VALUE, 1, CT>(a, smem);
}

static size_t sym_lds_bytes(int S, int A, int amax, int ct) {
    const int loop = ((sym_image_floats(S, amax) + 3) & ~3) + SYM_WAVES * sym_wave_floats(ct);
    const int red = 4 * (n_actor_params(S, A) + 1);   // the actor's arrays are the larger ones
    return (size_t)(loop > red ? loop : red) * sizeof(float);
}

template <bool CONT, int AMAX, int CT>
static int launch_sym(const FwdBwdArgs& a, hipStream_t st) {
    const size_t lds = sym_lds_bytes(a.S, a.A, AMAX, CT);
    if (lds > 160 * 1024) return PPO_E_UNSUPPORTED;
    static size_t configured = 0;   // per process: one process drives one device
    if (lds > configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)fwdbwd_sym_kernel<CONT, AMAX, CT>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    fwdbwd_sym_kernel<CONT, AMAX, CT><<<a.n_actor_wg + a.n_critic_wg, SYM_THREADS, lds, st>>>(a);
    return (int)hipGetLastError();
}

bool fwdbwd_sym_supported(int state_dim, int action_dim) {
    if (state_dim + 1 > 32) return false;
    const int amax = action_dim <= 2 ? 2 : (action_dim <= 4 ? 4 : 8);
    return sym_lds_bytes(state_dim, action_dim, amax, state_dim + 1 <= 16 ? 1 : 2) <= 160 * 1024;
}

// Whether this kernel is the default for a shape.  Measured on MI355X (bench.py, sustained clocks):
// 524 288 samples of S = 4 / 2 discrete actions: pair kernel 0.279 ms per epoch, this one 0.288;
// 8.4 M samples of S = 26 / 6 continuous actions: 5.45 and 5.43 ms; reference-sized buffers (900-6000
// samples): 25-34 us per epoch against 33-40 (this kernel's image fill and two-round reduction are
// the longer prologue / epilogue).  Its tile loop is the faster one (9.15 k cycles per 16 samples
// against 18.6 k per 32), its fixed costs and the imbalance between the two waves of a SIMD eat the
// difference: nowhere ahead by a margin worth a shape-dependent default, so it is opt-in
// (PPO_FWDBWD=sym) and kept under test as the alternative form.
bool fwdbwd_sym_default(int, int, int64_t) { return false; }

int launch_fwdbwd_sym(const FwdBwdArgs& a, int continuous, hipStream_t st) {
    const int ct = a.S + 1 <= 16 ? 1 : 2;
    const int amax = a.A <= 2 ? 2 : (a.A <= 4 ? 4 : 8);
#define SYM_DISPATCH_CT(C, AM)                              \
    if (ct == 1) return launch_sym<C, AM, 1>(a, st);        \
    return launch_sym<C, AM, 2>(a, st);
#define SYM_DISPATCH(C)                                     \
    if (amax == 2) { SYM_DISPATCH_CT(C, 2) }                \
    if (amax == 4) { SYM_DISPATCH_CT(C, 4) }                \
    SYM_DISPATCH_CT(C, 8)
    if (continuous) { SYM_DISPATCH(true) }
    SYM_DISPATCH(false)
#undef SYM_DISPATCH
#undef SYM_DISPATCH_CT
}
