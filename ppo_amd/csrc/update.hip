// K-epoch update: fused evaluate + clipped-surrogate loss + backward (one launch per epoch),
// deterministic gradient reduction, two-learning-rate Adam.
//
// fwdbwd kernel.  Up to 256 workgroups (one per CU) of 8 waves; the first n_actor_wg train the actor,
// the rest the critic (the two MLPs share no parameters and, given the precomputed advantages, their loss
// terms are independent: PPO.py:221,240).  On gfx950 the fp32 MFMA runs on the fp32 VALU datapath
// (tools/ubench/mfma_valu_overlap.hip: VALU work between MFMAs adds its full issue time at any
// occupancy), so the kernel's time is (MFMA count x 64 cycles) + (every other instruction), and the
// design goal is (a) as few non-MFMA instructions as possible and (b) two waves per SIMD so that
// one wave's LDS / global / MFMA-result latencies are filled by the other's instructions.
// A workgroup is 4 PAIRS of waves; pair p = waves p (forward wave) and p + 4 (backward wave) share
// tiles of 32 samples:
//   forward wave   layer 1, tanh, layer 2, tanh, output layer, distribution head, loss, gradient
//                  w.r.t. the head inputs, dZ2 = (W3^T dout)(1 - H2^2); keeps the output layer's
//                  and b2's gradient sums; hands dZ2 and H1 to its partner through LDS as
//                  [sample][unit] rows (16-byte stores: four consecutive units per register quad);
//   backward wave  dgrad dH1^T = W2^T dZ2^T, dZ1, and the two big weight gradients (the layer-1
//                  one deferred until the hand-off buffers are released, so that its MFMAs run
//                  under the forward wave's hand-off phase)
//                  dW2 += dZ2^T H1, dW1 += dZ1^T [X 1], accumulated across all its tiles in MFMA
//                  accumulator registers.
// The orientation keeps the sample on the MFMA lane (ppo_common.h), so forward and dgrad chain
// through registers and only the weight gradients - reductions over samples - go through LDS.
// Two workgroup barriers per tile order the hand-off; while the backward wave works on tile i the
// forward wave already runs the forward pass of tile i + 1.
// Nothing is accumulated with atomics: per-wave -> per-workgroup (fixed wave order through LDS) ->
// ppo_grad_reduce (fixed workgroup order), so results are bit-reproducible run to run.
#include "ppo_common.h"
#include "update_common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

#define FB_PAIRS 4
// update_vjp.hip compiles this file once more with PPO_FWDBWD_SEEDED defined: the same kernel with the
// caller's per-sample gradients as seeds (update_common.h), under another name, behind
// ppo_evaluate_backward; everything else of this file belongs to the unseeded build only.
#ifdef PPO_FWDBWD_SEEDED
#define FWDBWD_KERNEL fwdbwd_vjp_kernel
#else
#define FWDBWD_KERNEL fwdbwd_kernel
#endif
#define FB_THREADS (FB_PAIRS * 2 * 64)
// Per-pair hand-off space.  Rows are samples, columns hidden units, row stride W2_LD (= 68: 16-byte
// aligned rows whose starts are 4 banks apart, so 16-byte accesses of 8 consecutive rows and 4-byte
// accesses of 32 consecutive columns are both conflict-free).
//   dzs  [32][W2_LD]  dZ2[m][u2]  forward -> backward; the backward wave later reuses it for dZ1[m][u1]
//   h1s  [32][W2_LD]  H1[m][u1]   forward -> backward (H2 while the MFMA output-layer wgrad runs)
//   dtu  [9][DT_LD]   dout[a][m]  (MFMA output-layer wgrad only; row AOUT stays zero)
#define DT_LD 36
#define STG_MAT (TILE_M * W2_LD)
#define STG_PAIR_FLOATS (2 * STG_MAT + 9 * DT_LD)   // (+ STG_MAT for the private dZ1 buffer when S <= 31)
#define CS_LD 65                                       // row stride of colsum()'s scratch: odd => conflict-free
// 16x16x4 weight-gradient tiles read their operands column-wise from the [sample][unit] rows: k-step s of
// lane group q4 = lane >> 4 is sample NARROW_S(s) + NARROW_Q(q4).  A 4-byte LDS read services lane
// groups {0, 1} and {2, 3} together (32 lanes each, banks = dword address mod 32): their rows are 4
// apart, i.e. 4 * W2_LD = 16 banks apart, and each reads 16 consecutive columns, so the access is
// conflict-free (rows 1 apart, the plain 4s + q4 order, overlap in 12 of 16 banks: 2 LDS cycles more
// per read).  One k-step's four rows {o, o+1, o+4, o+5} stay within 6 rows, so that the matching global
// loads of narrow observations touch one cache line as they do in the plain order.
#define NARROW_S(s) (2 * ((s) & 1) + 8 * ((s) >> 1))
#define NARROW_Q(q4) (4 * ((q4) & 1) + ((q4) >> 1))
// column 4s + q4 of sample c = NARROW_S(s) + NARROW_Q(q4)
__device__ __forceinline__ int narrow_col(int c) {
    const int s = ((c >> 1) & 1) + 2 * ((c >> 3) & 1) + 4 * (c >> 4);
    const int q4 = ((c >> 2) & 1) + 2 * (c & 1);
    return 4 * s + q4;
}

// Diagnostic build only (-DPPO_PHASE_STAMPS): per-phase s_memtime sums, one row of 16 counters per
// wave, written to a buffer registered with ppo_debug_set_stamps().  Never compiled into the product.
#if defined(PPO_PHASE_STAMPS)
static unsigned long long* g_stamp_buf = nullptr;
#ifndef PPO_FWDBWD_SEEDED
extern "C" void ppo_debug_set_stamps(void* p) { g_stamp_buf = (unsigned long long*)p; }
#endif
#define STAMP_DECL unsigned long long st_acc[12] = {0}; unsigned long long st_prev = 0;
#define STAMP_START { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[i] += _t - st_prev; st_prev = _t; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_FLUSH(a, wave) if ((a).stamps && (threadIdx.x & 63) == 0) { unsigned long long* row = (a).stamps + ((size_t)blockIdx.x * 8 + (wave)) * 16; for (int i = 0; i < 12; ++i) row[i] = st_acc[i]; }
// wall-clock marks (100 MHz) of a workgroup's stages, slots 12..15 of every wave's row
#define WG_MARK(a, i) if ((a).stamps && (threadIdx.x & 63) == 0) (a).stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define WG_MARK(a, i)
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH(a, wave)
#endif

// Sum of per-lane partials over the 32 sample columns through `buf` (>= 16*CS_LD floats, private
// to the calling wave): one accumulator tile at a time, written as [register][lane], read back as 32
// consecutive floats per hidden unit by the half wave whose units the tile holds; lane u returns unit
// u's total.  Fixed order, no cross-lane instructions (a wave's LDS operations execute in order).
#define CS_FLOATS (16 * CS_LD)
// (WHOLE: both tiles at once through a buffer of 32*CS_LD floats; the same sums)
template <bool WHOLE = false>
__device__ __forceinline__ float colsum(float* buf, const f32x16 (&v)[2], int lane) {
    const int q = lane & 31;
    if constexpr (WHOLE) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[(16 * t + r) * CS_LD + lane] = v[t][r];
        const float* src = buf + (16 * (lane >> 5) + (q & 3) + 4 * (q >> 3)) * CS_LD + 32 * ((q >> 2) & 1);
        float acc = 0.f;
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) acc += src[cc];
        return acc;
    }
    const float* src = buf + ((q & 3) + 4 * (q >> 3)) * CS_LD + 32 * ((q >> 2) & 1);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[r * CS_LD + lane] = v[t][r];
        if ((lane >> 5) == t) {
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) acc += src[cc];
        }
    }
    return acc;
}

// The edges of a launch - staging the weights with every load in flight at once, the forward waves'
// column sums ahead of the post-loop barrier - are worth 3 - 4 us per launch (-1.3 % per epoch at 524 288
// samples of the CartPole shape, -12 % of a 1600-sample launch).  Observations of 16 - 31 values keep the
// plain forms: with the short edges their TILE LOOP measured 1.2 % slower at 8.4 M samples of the
// HalfCheetah shape (same instruction counts; the loop's register assignment comes out differently),
// which is 20 times what the edges save there.
__host__ __device__ constexpr bool fb_short_edges(int l1) { return l1 != 1; }

// Everything one wave needs to know about its place in the workgroup.
struct WaveCtx {
    NetLds L;
    NetLayout P;
    int base;          // first parameter of this MLP in the flat vector
    int S, A;
    int pair, lane, h, c;
    float* dzs;        // dZ2 [32][W2_LD], later dZ1
    float* h1s;        // H1  [32][W2_LD]
    float* dtu;        // dout [9][DT_LD]
    float* dz1;        // dZ1 [32][W2_LD]: private to the backward wave (S <= 31), else aliases dzs
    float* red;        // [P.n + 1] workgroup reduction buffer (aliases the transposition space)
    int64_t tile0, tstride, n_tiles;
    int n_iter;        // same for every wave of the grid: the barriers match
};

// Workgroup reduction: after the tile loop all of LDS is free; every pair owns a full
// parameter-indexed array red[pair][P.n + 1] into which its two waves store their (disjoint) sums
// with plain stores; one barrier later all threads add the four arrays in fixed order.
#define RED_PUT(idx, val) { ctx.red[(idx)] = (val); }

// ------------------------------------------------------------------------------ forward wave
// L1 = shape class of layer 1: 0: S+1 <= 16, 1: S+1 <= 32, 2: S+1 <= 64
template <int HEAD, int AOUT, int L1>
__device__ __forceinline__ void fwd_wave(const FwdBwdArgs& a, const WaveCtx& ctx) {
    const NetLds& L = ctx.L;
    const int S = ctx.S, A = ctx.A, lane = ctx.lane, h = ctx.h, c = ctx.c;
    float* const dzs = ctx.dzs;
    float* const h1s = ctx.h1s;
    float* const dtu = ctx.dtu;
    constexpr bool L3_VALU = AOUT <= 2;  // output-layer wgrad on the VALU (A*32 FMAs per tile)
    constexpr int KS1MAX = L1 == 0 ? 8 : (L1 == 1 ? 16 : 32);

    // Output-layer wgrad on the MFMA (AOUT > 2): 32x32x2 tiles for AOUT = 4, 16x16x4 tiles for AOUT = 8
    // (half the matrix time; measured -2 % per epoch at 6 continuous actions with the role split
    // retuned, but +2 % at 4 discrete ones, hence the two forms)
    constexpr bool L3_NARROW = AOUT > 4;
    f32x16 gW3[2], gb2[2], gW3v[L3_VALU ? AOUT : 1][2];
    f32x4 gW3n[4];   // L3_NARROW: tile t covers units 16t .. 16t+15, rows = outputs
    float gb3[AOUT];
    double loss_acc = 0.0;
    STAMP_DECL
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gW3[i] = 0.f;
        gb2[i] = 0.f;
#pragma unroll
        for (int k = 0; k < (L3_VALU ? AOUT : 1); ++k) gW3v[k][i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < AOUT; ++k) gb3[k] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) gW3n[i] = 0.f;

    // xf[s] = X_aug[this lane's sample][input 2s+h] (1 at input S, 0 beyond), fetched one tile
    // ahead.  Row pointer + constant offsets: one instruction per load.
    float xf[KS1MAX];
    auto load_xf = [&](int64_t t_m0, int half_s) {
        const int64_t mi = min(t_m0 + c, a.n - 1);
        const float* row = a.states + mi * S + h;
#pragma unroll
        for (int s = 0; s < KS1MAX; ++s) {
            if (s < half_s) xf[s] = row[2 * s];
            else if (s == half_s) xf[s] = (S & 1) ? (h == 0 ? row[2 * s] : 1.0f) : (h == 0 ? 1.0f : 0.0f);
        }
    };
    load_xf(min(ctx.tile0, ctx.n_tiles - 1) * TILE_M, S >> 1);
    for (int it = 0; it < ctx.n_iter; ++it) {
        const int64_t tile = ctx.tile0 + (int64_t)it * ctx.tstride;
        const int64_t m = tile * TILE_M + c;
        const bool valid = m < a.n;
        // loop-invariant, but kept opaque so that the 16 step guards are re-derived with one
        // scalar compare each instead of living in (spilled) scalar registers
        int ks1 = ks1_of(S), half_s = S >> 1;
        asm volatile("" : "+s"(ks1), "+s"(half_s));
        STAMP_START
        const HeadIn<HEAD, AOUT> hin = load_head_inputs<HEAD, AOUT>(a, A, m, valid);

        // ---------------- layer 1 (bias rides as the ones column of X)
        f32x16 H1[2], H2[2];
        {
            f32x16 z0 = {0}, z1 = {0};
#pragma unroll
            for (int s = 0; s < KS1MAX; ++s) {
                if (s < ks1) {
                    const int k = 2 * s + h;
                    z0 = mfma32(L.w1t[k * HID + c], xf[s], z0);
                    z1 = mfma32(L.w1t[k * HID + 32 + c], xf[s], z1);
                }
            }
            // next tile's observations: right away for narrow inputs; for S >= 32 (32 registers) only at
            // the end of the iteration, when the activations are dead
            if constexpr (L1 < 2) load_xf(min(tile + ctx.tstride, ctx.n_tiles - 1) * TILE_M, half_s);
            H1[0] = L1 == 0 ? tanh16_prescaled_pk(z0) : tanh16_prescaled(z0);
            H1[1] = L1 == 0 ? tanh16_prescaled_pk(z1) : tanh16_prescaled(z1);
        }
        STAMP(0)
        // ---------------- layer 2
        mlp_layer2<L1 == 0>(L, c, h, H1, H2);
        STAMP(1)
        float o[AOUT];
        out_layer<AOUT>(L, A, H2, h, o);

        // ---------------- head, loss, gradient w.r.t. the pre-head outputs
        float dout[AOUT];
        head_loss_grad<HEAD, AOUT>(a, A, o, hin, m, valid, h == 0, dout, loss_acc);
        if (h == 0) {
#pragma unroll
            for (int k = 0; k < AOUT; ++k) gb3[k] += dout[k];
        }
        if constexpr (L3_VALU) {  // dW3[k, u] += dout[m, k] * H2[m, u], per-lane partials
#pragma unroll
            for (int k = 0; k < AOUT; ++k) {
                gW3v[k][0] += dout[k] * H2[0];
                gW3v[k][1] += dout[k] * H2[1];
            }
        }

        STAMP(2)
        __syncthreads();  // the partner has finished reading the previous tile's buffers
        STAMP(3)

        float* const hrow = h1s + c * W2_LD + 4 * h;
        if constexpr (!L3_VALU) {
            // output-layer wgrad on the MFMA: dW3[a, u] += sum_m dout[m, a] H2[m, u]; A[i = output][k =
            // sample], B[k = sample][j = unit].  H2 passes through h1s before H1 takes its place.
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = H2[t][4 * q + j];
                    *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q) = v;
                }
            if (h == 0) {
                // narrow tiles: sample c = NARROW_S(s) + NARROW_Q(q4) is stored at column 4s + q4
                const int col = L3_NARROW ? narrow_col(c) : c;
#pragma unroll
                for (int k = 0; k < AOUT; ++k) dtu[k * DT_LD + col] = dout[k];
            }
            if constexpr (L3_NARROW) {
                const int c16 = lane & 15, q4 = lane >> 4;
                const float* arow = dtu + min(c16, AOUT) * DT_LD + q4;          // row AOUT is zero; + 4 s
                const float* bcol = h1s + NARROW_Q(q4) * W2_LD + c16;             // + NARROW_S(s) rows, + 16 t columns
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float av = arow[4 * s];
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4)
                        gW3n[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bcol[NARROW_S(s) * W2_LD + 16 * t4], gW3n[t4], 0, 0, 0);
                }
            } else {
                // k-step s of half-wave h is sample 16h + s
                const float* arow = dtu + min(c, AOUT) * DT_LD + 16 * h;  // row AOUT is zero
                const float* bcol = h1s + (16 * h) * W2_LD + c;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int sm = 4 * g + j;
                        gW3[0] = mfma32(av[j], bcol[sm * W2_LD], gW3[0]);
                        gW3[1] = mfma32(av[j], bcol[sm * W2_LD + 32], gW3[1]);
                    }
                }
            }
        }
        // ---------------- output layer backward: dZ2 = (W3^T dout) * (1 - H2^2), handed over
        float* const zrow = dzs + c * W2_LD + 4 * h;
        f32x16 G[2];
        if constexpr (!L3_VALU) {
            // G[u, m] = sum_a W3[a, u] dout[m, a] on the MFMA: k-step s covers outputs 2s and 2s+1
            // (rows >= A of the LDS copy of W3 are zero); the result lands in the accumulator
            // layout of H2, which is what the elementwise product needs
            G[0] = 0.f;
            G[1] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < AOUT / 2; ++s2) {
                const float b = h ? dout[2 * s2 + 1] : dout[2 * s2];
                const float* wr = L.w3 + (2 * s2 + h) * HID + c;
                G[0] = mfma32(wr[0], b, G[0]);
                G[1] = mfma32(wr[32], b, G[1]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (L3_VALU) {
#pragma unroll
                    for (int k = 0; k < AOUT; ++k) {
                        if (k < A) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(&L.w3[k * HID + 32 * t + 8 * q + 4 * h]);
                            g[0] = fmaf(w.x, dout[k], g[0]);
                            g[1] = fmaf(w.y, dout[k], g[1]);
                            g[2] = fmaf(w.z, dout[k], g[2]);
                            g[3] = fmaf(w.w, dout[k], g[3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = G[t][4 * q + j];
                }
                f32x4 dz4, h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * q + j;
                    const float hv = H2[t][r];
                    const float dz = g[j] * (1.0f - hv * hv);
                    gb2[t][r] += dz;
                    dz4[j] = dz;
                    h4[j] = H1[t][r];
                }
                *reinterpret_cast<f32x4*>(zrow + 32 * t + 8 * q) = dz4;   // units 32t + 8q + 4h .. +3
                *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q) = h4;
            }
        }
        STAMP(4)
        if constexpr (L1 == 2) load_xf(min(tile + ctx.tstride, ctx.n_tiles - 1) * TILE_M, half_s);
        __syncthreads();  // buffers of this tile are complete
        STAMP(5)
    }
    STAMP_START

    // per-lane partials -> per-unit sums, through the image of the forward pass's layer-2 weights: past
    // the loop's last barrier no forward wave reads it any more (the backward waves never do), so this
    // runs while the partner still works on the last tile
    static_assert(FB_PAIRS * CS_FLOATS <= HID * W2_LD, "the four transposition buffers fit the W2 image");
    constexpr bool EARLY = fb_short_edges(L1);
    if constexpr (!EARLY) __syncthreads();   // (the plain form: after the partner, through the pair's hand-off space)
    float* const cs = EARLY ? L.w2f + ctx.pair * CS_FLOATS : dzs;
    const float db2_u = colsum<!EARLY>(cs, gb2, lane);   // d/d b2[lane]
    float dw3_u[L3_VALU ? AOUT : 1];              // d/d W3[k][lane] on the VALU path
    if constexpr (L3_VALU) {
#pragma unroll
        for (int k = 0; k < AOUT; ++k) dw3_u[k] = colsum<!EARLY>(cs, gW3v[k], lane);
    }
#pragma unroll
    for (int k = 0; k < AOUT; ++k) {
        float v = gb3[k];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
        gb3[k] = v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) loss_acc += __shfl_xor(loss_acc, off);
    __syncthreads();  // every wave is done with the tile loop: LDS (weights and hand-off space) is free from here on
    const int o_b2 = ctx.P.b2 - ctx.base, o_w3 = ctx.P.w3 - ctx.base, o_b3 = ctx.P.b3 - ctx.base,
              o_loss = ctx.P.n;
    RED_PUT(o_b2 + lane, db2_u);
    if constexpr (L3_VALU) {
#pragma unroll
        for (int k = 0; k < AOUT; ++k)
            if (k < A) RED_PUT(o_w3 + k * HID + lane, dw3_u[k]);
    } else {
        if constexpr (L3_NARROW) {
            // C/D of the 16x16x4 tile: column (unit) = lane & 15, row (output) = 4 * (lane >> 4) + reg
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int act_row = 4 * (lane >> 4) + r;
                    if (act_row < A) RED_PUT(o_w3 + act_row * HID + 16 * t4 + (lane & 15), gW3n[t4][r]);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int act_row = rho(r, h);
                    if (act_row < A) RED_PUT(o_w3 + act_row * HID + 32 * j + c, gW3[j][r]);
                }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < AOUT; ++k) if (k < A) RED_PUT(o_b3 + k, gb3[k]);
        const float lv = HEAD == HEAD_VALUE ? (float)(0.5 * loss_acc * (double)a.inv_n)
                                            : (float)(loss_acc * (double)a.inv_n);
        RED_PUT(o_loss, lv);
    }
    __syncthreads();  // the four arrays are complete
    STAMP(6)
    STAMP_FLUSH(a, ctx.pair)
}

// ------------------------------------------------------------------------------ backward wave
template <int L1>
__device__ __forceinline__ void bwd_wave(const FwdBwdArgs& a, const WaveCtx& ctx) {
    const NetLds& L = ctx.L;
    const int S = ctx.S, lane = ctx.lane, h = ctx.h, c = ctx.c;
    float* const dzs = ctx.dzs;
    float* const h1s = ctx.h1s;
    constexpr int JT = L1 == 2 ? 2 : 1;
    constexpr bool NARROW = L1 == 0;  // S+1 <= 16: layer-1 wgrad on 16x16x4 tiles (half the MFMA time)
    const int c16 = lane & 15, q4 = lane >> 4;
    f32x16 gW2[2][2], gW1[2][JT];
    f32x4 gW1n[4];  // NARROW: tile it covers outputs 16*it .. 16*it+15, inputs 0..15
    STAMP_DECL
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) gW2[i][j] = 0.f;
#pragma unroll
        for (int j = 0; j < JT; ++j) gW1[i][j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) gW1n[i] = 0.f;

    // per-lane LDS addresses, loop-invariant
    const float* const wrow0 = L.w2b + c * W2_LD + 4 * h;           // W2^T rows c and 32 + c
    const float* const wrow1 = wrow0 + 32 * W2_LD;
    float* const zrow = dzs + c * W2_LD + 4 * h;                     // this lane's sample row
    const float* const hrow = h1s + c * W2_LD + 4 * h;
    const float* const zcol = dzs + (16 * h) * W2_LD + c;            // k-step s <-> sample 16h + s
    const float* const hcol = h1s + (16 * h) * W2_LD + c;

    // S <= 31: dZ1 has a buffer of its own, so the hand-off buffers are released BEFORE the layer-1
    // wgrad runs: its MFMAs overlap with the forward wave's (latency-bound) hand-off phase of the
    // next tile instead of preceding it.  Tile it-1's wgrad therefore runs inside iteration it.
    constexpr bool DEFER = L1 < 2;
    float* const z1row = ctx.dz1 + c * W2_LD + 4 * h;
    const float* const z1col = ctx.dz1 + (16 * h) * W2_LD + c;
    const int nq = NARROW_Q(q4);                                      // narrow tiles: this lane group's first sample
    const float* const z1coln = ctx.dz1 + nq * W2_LD + c16;
    float xp[NARROW ? 1 : 16][JT];
    float xq[NARROW ? 8 : 1];
    auto load_x = [&](int64_t m0) {
        // B operands of the layer-1 wgrad, X_aug[sample][input] (1 at input S, 0 beyond), fetched
        // ahead of the barriers so that they are long there when needed (lines the partner loaded a
        // tile ago).  Wide: xp[s][jt] = X_aug[16h+s][32*jt+c].  Narrow: xq[s] = X_aug[NARROW_S(s)+nq][c16].
        const bool full = m0 + TILE_M <= a.n;
        // (S >= 32 only; the base goes through readfirstlane so that it provably lives in scalar registers)
        auto row_base = [&](int row) {
            const uint64_t p = reinterpret_cast<uint64_t>(a.states + (m0 + row) * S);
            const uint64_t u = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p) |
                               (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32;
            return reinterpret_cast<const char*>(u);
        };
        if constexpr (NARROW) {
            const float fill = c16 == S ? 1.0f : 0.0f;
            const int inc = min(c16, S - 1);
            if (full) {
                const float* p0 = a.states + (m0 + nq) * S + inc;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float v = p0[(int64_t)NARROW_S(s) * S];
                    xq[s] = c16 < S ? v : fill;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int64_t gm = min(m0 + NARROW_S(s) + nq, a.n - 1);
                    const float v = a.states[gm * S + inc];
                    xq[s] = c16 < S ? v : fill;
                }
            }
        } else {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const int in = 32 * jt + c;
                const float fill = in == S ? 1.0f : 0.0f;
                const int inc = min(in, S - 1);
                if (full && L1 == 2) {
                    // 32 loads: (scalar row base of k-step s) + (this lane's 32-bit byte offset, the same
                    // for every tile) instead of 32 loop-invariant 64-bit addresses, which would not fit
                    // next to the 128 accumulator registers
                    const unsigned voff = (unsigned)(16 * h * S + inc) * 4u;
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float v = *reinterpret_cast<const float*>(row_base(s) + voff);
                        xp[s][jt] = in < S ? v : fill;
                    }
                } else if (full) {
                    const float* p0 = a.states + (m0 + 16 * h) * S + inc;
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float v = p0[(int64_t)s * S];
                        xp[s][jt] = in < S ? v : fill;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int64_t gm = min(m0 + 16 * h + s, a.n - 1);
                        const float v = a.states[gm * S + inc];
                        xp[s][jt] = in < S ? v : fill;
                    }
                }
            }
        }
    };
    auto wgrad1 = [&]() {
        // ---------------- wgrad of layer 1 (+ bias as the ones column): dW1[o, i] += dZ1[m, o] X[m, i]
        if constexpr (NARROW) {
            // v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    gW1n[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(z1coln[NARROW_S(s) * W2_LD + 16 * t4],
                                                                     xq[s], gW1n[t4], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a0 = z1col[s * W2_LD];
                const float a1 = z1col[s * W2_LD + 32];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    gW1[0][jt] = mfma32(a0, xp[s][jt], gW1[0][jt]);
                    gW1[1][jt] = mfma32(a1, xp[s][jt], gW1[1][jt]);
                }
            }
        }
    };

    auto wgrad2 = [&]() {
        // ---------------- wgrad of layer 2: dW2[o, i] += sum_m dZ2[m, o] H1[m, i]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = zcol[s * W2_LD];
            const float a1 = zcol[s * W2_LD + 32];
            const float b0 = hcol[s * W2_LD];
            const float b1 = hcol[s * W2_LD + 32];
            gW2[0][0] = mfma32(a0, b0, gW2[0][0]);
            gW2[0][1] = mfma32(a0, b1, gW2[0][1]);
            gW2[1][0] = mfma32(a1, b0, gW2[1][0]);
            gW2[1][1] = mfma32(a1, b1, gW2[1][1]);
        }
    };

    for (int it = 0; it < ctx.n_iter; ++it) {
        const int64_t tile = ctx.tile0 + (int64_t)it * ctx.tstride;
        STAMP_START
        if constexpr (DEFER) {
            if (it > 0) load_x((tile - ctx.tstride) * TILE_M);   // operands of the deferred wgrad
        }
        STAMP(0)
        __syncthreads();  // (forward wave: previous buffers may be overwritten)
        STAMP(1)
        if constexpr (DEFER) {
            if (it > 0) wgrad1();
        }
        __syncthreads();  // buffers of this tile are complete
        STAMP(2)
        // S >= 32 (128 accumulator registers): the layer-2 wgrad runs FIRST, so that dH1's 32 registers
        // are not live across it, and the layer-1 wgrad's 32 operand registers are fetched only after
        // it (their latency passes under the dgrad)
        if constexpr (!DEFER) {
            wgrad2();
            load_x(tile * TILE_M);
        }
        // ---------------- dgrad of layer 2: dH1[in, m] = sum_out W2[out, in] dZ2[m, out]; the four
        // k-steps of group g are output units 8g + 4h .. +3: one 16-byte load per operand
        f32x16 d0 = {0}, d1 = {0};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(zrow + 8 * g);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(wrow0 + 8 * g);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(wrow1 + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d0 = mfma32(a0[j], b4[j], d0);
                d1 = mfma32(a1[j], b4[j], d1);
            }
        }
        STAMP(3)
        // ---------------- through tanh: dZ1 = dH1 * (1 - H1^2), kept in registers until the layer-2
        // wgrad has read dZ2
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(hrow + 8 * q);
            const f32x4 h1 = *reinterpret_cast<const f32x4*>(hrow + 32 + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {   // on register pairs (v_pk_fma_f32 / v_pk_mul_f32)
                const f32x2 one = {1.0f, 1.0f};
                const f32x2 a = {h0[j], h0[j + 1]}, b = {h1[j], h1[j + 1]};
                const f32x2 ta = __builtin_elementwise_fma(-a, a, one), tb = __builtin_elementwise_fma(-b, b, one);
                const f32x2 da = (f32x2){d0[4 * q + j], d0[4 * q + j + 1]} * ta;
                const f32x2 db = (f32x2){d1[4 * q + j], d1[4 * q + j + 1]} * tb;
                d0[4 * q + j] = da.x; d0[4 * q + j + 1] = da.y;
                d1[4 * q + j] = db.x; d1[4 * q + j + 1] = db.y;
            }
        }
        STAMP(4)
        if constexpr (DEFER) wgrad2();
        STAMP(5)
        // ---------------- dZ1 goes to its own buffer (S <= 31) or takes dZ2's place (this wave's
        // own reads of dZ2 are complete: LDS operations of one wave execute in order)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = d0[4 * q + j];
                v1[j] = d1[4 * q + j];
            }
            *reinterpret_cast<f32x4*>(z1row + 8 * q) = v0;
            *reinterpret_cast<f32x4*>(z1row + 32 + 8 * q) = v1;
        }
        if constexpr (!DEFER) wgrad1();
        STAMP(6)
    }
    if constexpr (DEFER) {  // the last tile's layer-1 wgrad
        load_x((ctx.tile0 + (int64_t)(ctx.n_iter - 1) * ctx.tstride) * TILE_M);
        wgrad1();
    }
    STAMP_START

    if constexpr (!fb_short_edges(L1)) __syncthreads();
    __syncthreads();  // every wave is done with the tile loop: LDS (weights and hand-off space) is free from here on
    const int o_w1 = ctx.P.w1 - ctx.base, o_b1 = ctx.P.b1 - ctx.base, o_w2 = ctx.P.w2 - ctx.base;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int out = 32 * i + rho(r, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) RED_PUT(o_w2 + out * HID + 32 * j + c, gW2[i][j][r]);
            if constexpr (!NARROW) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int in = 32 * jt + c;
                    if (in < S) RED_PUT(o_w1 + out * S + in, gW1[i][jt][r])
                    else if (in == S) RED_PUT(o_b1 + out, gW1[i][jt][r])
                }
            }
        }
    if constexpr (NARROW) {
        // C/D of the 16x16x4 tile: column (input) = lane & 15, row (output) = 4 * (lane >> 4) + reg
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int out = 16 * t4 + 4 * q4 + r;
                if (c16 < S) RED_PUT(o_w1 + out * S + c16, gW1n[t4][r])
                else if (c16 == S) RED_PUT(o_b1 + out, gW1n[t4][r])
            }
    }
    __syncthreads();  // the four arrays are complete
    STAMP(7)
    STAMP_FLUSH(a, FB_PAIRS + ctx.pair)
}
#undef RED_PUT

template <int HEAD, int AOUT, int L1>
__device__ __forceinline__ void run_role(const FwdBwdArgs& a, float* smem) {
    WG_MARK(a, 12)
    WaveCtx ctx;
    ctx.S = a.S;
    ctx.A = HEAD == HEAD_VALUE ? 1 : a.A;
    ctx.base = HEAD == HEAD_VALUE ? n_actor_params(a.S, a.A) : 0;
    ctx.P = net_layout(ctx.base, a.S, ctx.A);
    const bool critic = HEAD == HEAD_VALUE;
    const int wg = critic ? blockIdx.x - a.n_actor_wg : blockIdx.x;
    const int nwg = critic ? gridDim.x - a.n_actor_wg : a.n_actor_wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ctx.lane = threadIdx.x & 63;
    ctx.h = ctx.lane >> 5;
    ctx.c = ctx.lane & 31;
    ctx.pair = wave & (FB_PAIRS - 1);
    ctx.L = carve_net_lds(smem, a.S, AOUT, true);
    float* stg = smem + round4(net_lds_floats(a.S, AOUT, true));
    constexpr int PAIR_FLOATS = STG_PAIR_FLOATS + (L1 < 2 ? STG_MAT : 0);
    ctx.dzs = stg + ctx.pair * PAIR_FLOATS;
    ctx.h1s = ctx.dzs + STG_MAT;
    ctx.dtu = ctx.h1s + STG_MAT;
    ctx.dz1 = L1 < 2 ? ctx.dtu + 9 * DT_LD : ctx.dzs;
    const int red_stride = ctx.P.n + 1;
    ctx.red = smem + ctx.pair * red_stride;  // valid once the tile loops are over
    ctx.n_tiles = (a.n + TILE_M - 1) / TILE_M;
    ctx.tstride = (int64_t)nwg * FB_PAIRS;
    ctx.tile0 = (int64_t)wg * FB_PAIRS + ctx.pair;
    ctx.n_iter = (int)((ctx.n_tiles + ctx.tstride - 1) / ctx.tstride);
    fill_net_lds<fb_short_edges(L1)>(ctx.L, a.params, ctx.P, a.S, ctx.A, AOUT, true);
    for (int i = threadIdx.x; i < FB_PAIRS * 9 * DT_LD; i += FB_THREADS)
        stg[(i / (9 * DT_LD)) * PAIR_FLOATS + 2 * STG_MAT + i % (9 * DT_LD)] = 0.f;
    __syncthreads();
    WG_MARK(a, 13)

    if (wave < FB_PAIRS) fwd_wave<HEAD, AOUT, L1>(a, ctx);
    else bwd_wave<L1>(a, ctx);
    WG_MARK(a, 14)

    // (both wave kinds end on the barrier that completes the four per-pair arrays)
    float* dst = HEAD == HEAD_VALUE ? a.partial_critic + (size_t)wg * a.stride_critic
                                    : a.partial_actor + (size_t)wg * a.stride_actor;
    for (int i = threadIdx.x; i < red_stride; i += FB_THREADS)
        dst[i] = ((smem[i] + smem[red_stride + i]) + smem[2 * red_stride + i]) + smem[3 * red_stride + i];
    WG_MARK(a, 15)
}

template <bool CONT, int AMAX, int L1>
__global__ __launch_bounds__(FB_THREADS, 2) void FWDBWD_KERNEL(const FwdBwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.ws_header[0] = a.n_actor_wg;
        a.ws_header[1] = (int)gridDim.x - a.n_actor_wg;
    }
    if ((int)blockIdx.x < a.n_actor_wg)
        run_role<CONT ? HEAD_GAUSSIAN : HEAD_CATEGORICAL, AMAX, L1>(a, smem);
    else
        run_role<HEAD_VALUE, 1, L1>(a, smem);
}

#ifndef PPO_FWDBWD_SEEDED
// grad[p] = sum over the role's workgroups (fixed order, fp64 accumulate).  One block per 64
// parameters: 4 row groups x 64 columns, the 4 partial sums combined in fixed order through LDS.
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float* __restrict__ partial_actor,
                                                          const float* __restrict__ partial_critic,
                                                          int stride_actor, int stride_critic,
                                                          int n_actor, int n_critic, const int* __restrict__ ws_header,
                                                          float* __restrict__ grad) {
    __shared__ double part[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int P = n_actor + n_critic;
    const int rows_actor = ws_header[0], rows_critic = ws_header[1];
    double s = 0.0;
    if (i < P + 2) {
        const float* src;
        int stride, rows;
        if (i < n_actor) { src = partial_actor + i; stride = stride_actor; rows = rows_actor; }
        else if (i < P) { src = partial_critic + (i - n_actor); stride = stride_critic; rows = rows_critic; }
        else if (i == P) { src = partial_actor + n_actor; stride = stride_actor; rows = rows_actor; }
        else { src = partial_critic + n_critic; stride = stride_critic; rows = rows_critic; }
        const int per = (rows + 3) / 4;
        const int lo = grp * per, hi = min(rows, lo + per);
#pragma unroll 8
        for (int g = lo; g < hi; ++g) s += (double)src[(size_t)g * stride];
    }
    part[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < P + 2) grad[i] = (float)(((part[0][col] + part[1][col]) + part[2][col]) + part[3][col]);
}

// torch/optim/adam.py:457-547, single-tensor path; operation order kept (no contraction).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ params,
                                                   const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int n, int n_actor, float neg_step_actor,
                                                   float neg_step_critic, float one_minus_b1,
                                                   float b2, float one_minus_b2, float bc2_sqrt,
                                                   float eps, float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && loss_out) loss_out[0] = grad[n] + grad[n + 1];
    if (i >= n) return;
    const float g = grad[i];
    float mi = m[i], vi = v[i];
    mi = mi + one_minus_b1 * (g - mi);          // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * b2;                               // exp_avg_sq.mul_(beta2)
    vi = vi + (one_minus_b2 * g) * g;           //   .addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float ns = i < n_actor ? neg_step_actor : neg_step_critic;
    params[i] = params[i] + (ns * mi) / denom;  // param.addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mi;
    v[i] = vi;
}

// grad_reduce_kernel + adam_kernel in one launch (single process: nothing happens in between).
__global__ __launch_bounds__(256) void reduce_adam_kernel(const float* __restrict__ partial_actor,
                                                          const float* __restrict__ partial_critic,
                                                          int stride_actor, int stride_critic,
                                                          int n_actor, int n_critic, const int* __restrict__ ws_header,
                                                          float* __restrict__ grad,
                                                          float* __restrict__ params,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float neg_step_actor, float neg_step_critic,
                                                          float one_minus_b1, float b2,
                                                          float one_minus_b2, float bc2_sqrt, float eps,
                                                          float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    __shared__ double part[4][64];
    __shared__ float gsh[64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int P = n_actor + n_critic;
    const int rows_actor = ws_header[0], rows_critic = ws_header[1];
    double s = 0.0;
    if (i < P + 2) {
        const float* src;
        int stride, rows;
        if (i < n_actor) { src = partial_actor + i; stride = stride_actor; rows = rows_actor; }
        else if (i < P) { src = partial_critic + (i - n_actor); stride = stride_critic; rows = rows_critic; }
        else if (i == P) { src = partial_actor + n_actor; stride = stride_actor; rows = rows_actor; }
        else { src = partial_critic + n_critic; stride = stride_critic; rows = rows_critic; }
        const int per = (rows + 3) / 4;
        const int lo = grp * per, hi = min(rows, lo + per);
#pragma unroll 8
        for (int g = lo; g < hi; ++g) s += (double)src[(size_t)g * stride];
    }
    part[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < P + 2) {
        const float g = (float)(((part[0][col] + part[1][col]) + part[2][col]) + part[3][col]);
        grad[i] = g;
        gsh[col] = g;
        if (i < P) {
            float mi = m[i], vi = v[i];
            mi = mi + one_minus_b1 * (g - mi);
            vi = vi * b2;
            vi = vi + (one_minus_b2 * g) * g;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            const float ns = i < n_actor ? neg_step_actor : neg_step_critic;
            params[i] = params[i] + (ns * mi) / denom;
            m[i] = mi;
            v[i] = vi;
        }
    }
    __syncthreads();
    // the two loss terms are entries P and P+1; the host only launches this kernel when both
    // fall into one block
    if (loss_out && threadIdx.x == 0 && (int)blockIdx.x == P / 64)
        loss_out[0] = gsh[P & 63] + gsh[(P + 1) & 63];
}

#endif  // !PPO_FWDBWD_SEEDED

// ---------------------------------------------------------------------------- host side
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
size_t ppo_scan_ws_bytes(int64_t n);  // scan.hip

static inline int stride_of(int n) { return (n + 1 + 63) / 64 * 64; }

// Workgroups per role.  The actor's tiles cost more than the critic's (wider output layer, the
// distribution head, for more than two outputs an extra MFMA weight gradient): measured ratios r
// of per-tile time are 1.12 (2 discrete actions), 1.25 (4) and 1.31 (6 continuous).  Every pair of a role
// runs ceil(tiles / pairs) tiles, so the split minimises max(actor iterations * r, critic
// iterations); at 524 288 samples that is 135 : 121 (31 and 34 iterations: 273.5 us per launch against
// 275.8 for 138 : 118 = 30 and 35 iterations, same-box sweep), at 8.4 M samples of 6 actions 145 : 111.
// The fwd/bwd kernel leaves the split in the workspace header for the reduction kernels.
#define FB_MAX_ROLE_WGS (3 * FB_GRID / 4)
#define FB_WS_HEADER 256
// tile: samples per tile; slots: tiles a workgroup works on at a time (wave pairs, or waves)
static void role_wgs(int A, int64_t n, int tile, int slots, int& n_actor, int& n_critic) {
    static const char* tune = getenv("PPO_ACTOR_WGS");   // tuning aid: fixed split for every shape
    const int64_t tiles = (n + tile - 1) / tile;
    int best = FB_GRID / 2;
    if (tune && atoi(tune) >= FB_GRID - FB_MAX_ROLE_WGS && atoi(tune) <= FB_MAX_ROLE_WGS) {
        best = atoi(tune);
    } else {
        const double r = 1.0 + (A > 4 ? 0.051 : (A > 2 ? 0.062 : 0.06)) * A;
        double best_cost = 1e300;
        for (int na = FB_GRID - FB_MAX_ROLE_WGS; na <= FB_MAX_ROLE_WGS; ++na) {
            const int64_t ia = (tiles + slots * na - 1) / (slots * na);
            const int64_t ic = (tiles + slots * (FB_GRID - na) - 1) / (slots * (FB_GRID - na));
            const double ca = ia * r, cc = (double)ic;
            // ties (small buffers): prefer the split closest to the continuous optimum
            const double cost = (ca > cc ? ca : cc) * (1.0 + 1e-6 * fabs(na - FB_GRID * r / (1.0 + r)));
            if (cost < best_cost) { best_cost = cost; best = na; }
        }
    }
    // Small buffers: a workgroup without a tile would still stage the weights and write a full row
    // of (zero) partial sums that the reduction then reads; launch only the workgroups that get one.
    const int needed = (int)((tiles + slots - 1) / slots);
    n_actor = best < needed ? best : needed;
    n_critic = FB_GRID - best < needed ? FB_GRID - best : needed;
}

// [header | actor partials, FB_MAX_ROLE_WGS rows | critic partials, FB_MAX_ROLE_WGS rows]
static size_t ws_critic_offset(int S, int A) {
    return FB_WS_HEADER + align_up((size_t)FB_MAX_ROLE_WGS * stride_of(n_actor_params(S, A)) * sizeof(float), 256);
}
static size_t update_ws_bytes(int S, int A) {
    return ws_critic_offset(S, A) +
           align_up((size_t)FB_MAX_ROLE_WGS * stride_of(n_critic_params(S)) * sizeof(float), 256);
}

#ifndef PPO_FWDBWD_SEEDED
extern "C" size_t ppo_workspace_bytes(int state_dim, int action_dim, int64_t max_samples) {
    if (state_dim < 1 || action_dim < 1 || max_samples < 1) return 0;
    const size_t a = update_ws_bytes(state_dim, action_dim);
    const size_t b = ppo_scan_ws_bytes(max_samples);
    return a > b ? a : b;
}

extern "C" int64_t ppo_param_count(int state_dim, int action_dim, int64_t* n_actor_out) {
    const int na = n_actor_params(state_dim, action_dim);
    if (n_actor_out) *n_actor_out = na;
    return (int64_t)na + n_critic_params(state_dim);
}

#endif  // !PPO_FWDBWD_SEEDED

static size_t fwdbwd_lds_bytes(int S, int A, int amax, int l1) {
    const int net = round4(net_lds_floats(S, amax, true));
    const int loop = net + FB_PAIRS * (STG_PAIR_FLOATS + (l1 < 2 ? STG_MAT : 0));
    const int red = FB_PAIRS * (n_actor_params(S, A) + 1);  // the actor's arrays are the larger ones
    return (size_t)(loop > red ? loop : red) * sizeof(float);
}

template <bool CONT, int AMAX, int L1>
static int launch_fwdbwd(const FwdBwdArgs& a, hipStream_t st) {
    const size_t lds = fwdbwd_lds_bytes(a.S, a.A, AMAX, L1);
    if (lds > 160 * 1024) return PPO_E_UNSUPPORTED;
    static size_t configured = 0;
    if (lds > configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)FWDBWD_KERNEL<CONT, AMAX, L1>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    FWDBWD_KERNEL<CONT, AMAX, L1><<<a.n_actor_wg + a.n_critic_wg, FB_THREADS, lds, st>>>(a);
    return (int)hipGetLastError();
}

// Role split, workspace carve and launch of the fused kernel for arguments filled by the caller.
static int fwdbwd_dispatch(FwdBwdArgs& a, int continuous, void* workspace, hipStream_t st) {
    const int state_dim = a.S, action_dim = a.A;
#ifdef PPO_PHASE_STAMPS
    a.stamps = g_stamp_buf;
#else
    a.stamps = nullptr;
#endif
    a.stride_actor = stride_of(n_actor_params(state_dim, action_dim));
    a.stride_critic = stride_of(n_critic_params(state_dim));
    a.ws_header = (int*)workspace;
    a.partial_actor = (float*)((char*)workspace + FB_WS_HEADER);
    // Symmetric-wave kernel (update_sym.hip) where selected, in the library variant that carries it; the
    // forward/backward-pair kernel below otherwise.  PPO_FWDBWD=pair / sym forces one of them there
    // (cross-checks, A/B timing).
#if defined(PPO_WITH_SYM) && !defined(PPO_FWDBWD_SEEDED)   // the library variant that carries update_sym.hip
    static const char* mode = getenv("PPO_FWDBWD");
    bool sym = fwdbwd_sym_default(state_dim, action_dim, a.n);
    if (mode && !strcmp(mode, "pair")) sym = false;
    if (mode && !strcmp(mode, "sym")) sym = fwdbwd_sym_supported(state_dim, action_dim);
    if (sym) {
        role_wgs(action_dim, a.n, 16, 8, a.n_actor_wg, a.n_critic_wg);
        a.partial_critic = (float*)((char*)workspace + ws_critic_offset(state_dim, action_dim));
        return launch_fwdbwd_sym(a, continuous, st);
    }
#endif
    role_wgs(action_dim, a.n, TILE_M, FB_PAIRS, a.n_actor_wg, a.n_critic_wg);
    a.partial_critic = (float*)((char*)workspace + ws_critic_offset(state_dim, action_dim));
    const int l1 = state_dim + 1 <= 16 ? 0 : (state_dim + 1 <= 32 ? 1 : 2);
    const int amax = action_dim <= 2 ? 2 : (action_dim <= 4 ? 4 : 8);
#define FB_DISPATCH_L1(C, AM)                                                            \
    if (l1 == 0) return launch_fwdbwd<C, AM, 0>(a, st);                                  \
    if (l1 == 1) return launch_fwdbwd<C, AM, 1>(a, st);                                  \
    return launch_fwdbwd<C, AM, 2>(a, st);
#define FB_DISPATCH(C)                                                                   \
    if (amax == 2) { FB_DISPATCH_L1(C, 2) }                                              \
    if (amax == 4) { FB_DISPATCH_L1(C, 4) }                                              \
    FB_DISPATCH_L1(C, 8)
    if (continuous) { FB_DISPATCH(true) }
    FB_DISPATCH(false)
#undef FB_DISPATCH
#undef FB_DISPATCH_L1
}

#ifdef PPO_FWDBWD_SEEDED
// The vector-Jacobian product of ActorCritic.evaluate (PPO.py:102-121) w.r.t. the parameters: what
// loss.backward() computes in the reference (PPO.py:244) for ANY loss built from evaluate()'s three
// outputs.  Same fused kernel, seeded with the caller's per-sample gradients.
extern "C" int ppo_evaluate_backward(const float* params, int state_dim, int action_dim, int continuous,
                                     float action_var, const float* states, const void* actions, int64_t n,
                                     const float* grad_logprobs, const float* grad_values,
                                     const float* grad_entropy, float* grad_params_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!params || !states || !actions || !grad_params_out || n <= 0) return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 ||
        action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < update_ws_bytes(state_dim, action_dim)) return PPO_E_WORKSPACE;
    FwdBwdArgs a{};
    a.params = params;
    a.S = state_dim;
    a.A = action_dim;
    a.action_var = action_var;
    a.states = states;
    a.actions = actions;
    a.n = n;
    a.inv_n = 1.0f;
    a.clip_lo = 0.f;
    a.clip_hi = 0.f;
    a.seed_logp = grad_logprobs;
    a.seed_val = grad_values;
    a.seed_ent = grad_entropy;
    const int rc = fwdbwd_dispatch(a, continuous, workspace, (hipStream_t)stream);
    if (rc) return rc;
    return ppo_grad_reduce(state_dim, action_dim, workspace, workspace_bytes, grad_params_out, stream);
}
#else
extern "C" int ppo_epoch_fwdbwd(const float* params, int state_dim, int action_dim, int continuous,
                                float action_var, const float* states, const void* actions,
                                const float* old_logprobs, const float* advantages,
                                const float* returns_norm, int64_t n_local, int64_t n_global,
                                float eps_clip, float* out_logprobs, float* out_values,
                                float* out_entropy, float* out_surrogate, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!params || !states || !actions || !old_logprobs || !advantages || !returns_norm ||
        n_local <= 0 || n_global < n_local)
        return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 ||
        action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < update_ws_bytes(state_dim, action_dim)) return PPO_E_WORKSPACE;
    FwdBwdArgs a{};
    a.params = params;
    a.S = state_dim;
    a.A = action_dim;
    a.action_var = action_var;
    a.states = states;
    a.actions = actions;
    a.old_logp = old_logprobs;
    a.adv = advantages;
    a.ret = returns_norm;
    a.n = n_local;
    a.inv_n = 1.0f / (float)n_global;
    // 1-eps_clip and 1+eps_clip are Python doubles handed to clamp on an fp32 tensor (PPO.py:237)
    a.clip_lo = (float)(1.0 - (double)eps_clip);
    a.clip_hi = (float)(1.0 + (double)eps_clip);
    a.out_logp = out_logprobs;
    a.out_val = out_values;
    a.out_ent = out_entropy;
    a.out_surr = out_surrogate;
    return fwdbwd_dispatch(a, continuous, workspace, (hipStream_t)stream);
}

extern "C" int ppo_grad_reduce(int state_dim, int action_dim, const void* workspace,
                               size_t workspace_bytes, float* grad_out, void* stream) {
    if (!workspace || !grad_out) return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 ||
        action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    if (workspace_bytes < update_ws_bytes(state_dim, action_dim)) return PPO_E_WORKSPACE;
    const int na = n_actor_params(state_dim, action_dim), nc = n_critic_params(state_dim);
    const int sa = stride_of(na), sc = stride_of(nc);
    const float* pa = (const float*)((const char*)workspace + FB_WS_HEADER);
    const float* pc = (const float*)((const char*)workspace + ws_critic_offset(state_dim, action_dim));
    const int total = na + nc + 2;
    grad_reduce_kernel<<<(total + 63) / 64, 256, 0, (hipStream_t)stream>>>(pa, pc, sa, sc, na, nc,
                                                                          (const int*)workspace, grad_out);
    return (int)hipGetLastError();
}

extern "C" int ppo_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                             int64_t n_params, int64_t n_actor, int64_t step, double lr_actor,
                             double lr_critic, double beta1, double beta2, double eps,
                             float* loss_out, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || n_params <= 0 || n_actor < 0 ||
        n_actor > n_params || step < 1 || n_params > (1 << 30))
        return PPO_E_BADARG;
    // bias corrections in double, as the reference's Python floats (adam.py:531-536)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float nsa = (float)(-(lr_actor / bc1));
    const float nsc = (float)(-(lr_critic / bc1));
    adam_kernel<<<(int)((n_params + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        params, grad, exp_avg, exp_avg_sq, (int)n_params, (int)n_actor, nsa, nsc,
        (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), bc2_sqrt, (float)eps, loss_out);
    return (int)hipGetLastError();
}

extern "C" int ppo_reduce_adam_step(int state_dim, int action_dim, const void* workspace,
                                    size_t workspace_bytes, float* grad_out, float* params,
                                    float* exp_avg, float* exp_avg_sq, int64_t step, double lr_actor,
                                    double lr_critic, double beta1, double beta2, double eps,
                                    float* loss_out, void* stream) {
    if (!workspace || !grad_out || !params || !exp_avg || !exp_avg_sq || step < 1) return PPO_E_BADARG;
    if (state_dim < 1 || state_dim > PPO_MAX_STATE_DIM || action_dim < 1 ||
        action_dim > PPO_MAX_ACTION_DIM)
        return PPO_E_UNSUPPORTED;
    if (workspace_bytes < update_ws_bytes(state_dim, action_dim)) return PPO_E_WORKSPACE;
    const int na = n_actor_params(state_dim, action_dim), nc = n_critic_params(state_dim);
    const int P = na + nc;
    if ((P / 64) != ((P + 1) / 64)) {
        // the two loss terms fall into different blocks: use the two-launch form
        const int rc = ppo_grad_reduce(state_dim, action_dim, workspace, workspace_bytes, grad_out, stream);
        if (rc) return rc;
        return ppo_adam_step(params, grad_out, exp_avg, exp_avg_sq, P, na, step, lr_actor, lr_critic,
                             beta1, beta2, eps, loss_out, stream);
    }
    const int sa = stride_of(na), sc = stride_of(nc);
    const float* pa = (const float*)((const char*)workspace + FB_WS_HEADER);
    const float* pc = (const float*)((const char*)workspace + ws_critic_offset(state_dim, action_dim));
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    reduce_adam_kernel<<<(P + 2 + 63) / 64, 256, 0, (hipStream_t)stream>>>(
        pa, pc, sa, sc, na, nc, (const int*)workspace, grad_out, params,
        exp_avg, exp_avg_sq,
        (float)(-(lr_actor / bc1)), (float)(-(lr_critic / bc1)), (float)(1.0 - beta1), (float)beta2,
        (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, loss_out);
    return (int)hipGetLastError();
}

// The K-epoch loop of PPO.update (PPO.py:224-245) for one process: K x (ppo_epoch_fwdbwd,
// ppo_reduce_adam_step) enqueued by one call; losses_out[e] receives loss.mean() of epoch e.
// ---------------------------------------------------------------------------- kernel timings
// Events on the launch stream around the launches of the K-epoch loop, switched on by
// ppo_kernel_timings_enable(1).  The events are made on first use and kept (one process drives one
// device); with timings off ppo_update_epochs does not touch any of this.
#define PPO_TIMED_EPOCHS_MAX 256
static struct {
    int on = 0, epochs = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2 * PPO_TIMED_EPOCHS_MAX + 1] = {};
    int made = 0;
} g_timing;

extern "C" int ppo_kernel_timings_enable(int on) {
    g_timing.on = on != 0;
    g_timing.epochs = 0;
    return 0;
}

static int timing_mark(int i, hipStream_t st) {
    while (g_timing.made <= i) {
        const hipError_t e = hipEventCreate(&g_timing.ev[g_timing.made]);
        if (e != hipSuccess) return (int)e;
        ++g_timing.made;
    }
    return (int)hipEventRecord(g_timing.ev[i], st);
}

extern "C" int ppo_kernel_timings(float* out, int n) {
    if (!out || n < 3) return PPO_E_BADARG;
    out[0] = out[1] = out[2] = 0.f;
    const int k = g_timing.epochs;
    if (k == 0) return 0;
    hipError_t e = hipEventSynchronize(g_timing.ev[2 * k]);
    if (e != hipSuccess) return (int)e;
    double fb = 0.0, ra = 0.0;
    for (int i = 0; i < k; ++i) {
        float a = 0.f, b = 0.f;
        if ((e = hipEventElapsedTime(&a, g_timing.ev[2 * i], g_timing.ev[2 * i + 1])) != hipSuccess) return (int)e;
        if ((e = hipEventElapsedTime(&b, g_timing.ev[2 * i + 1], g_timing.ev[2 * i + 2])) != hipSuccess) return (int)e;
        fb += a;
        ra += b;
    }
    out[0] = (float)(fb / k);
    out[1] = (float)(ra / k);
    out[2] = (float)k;
    return 0;
}

extern "C" int ppo_update_epochs(float* params, int state_dim, int action_dim, int continuous,
                                 float action_var, const float* states, const void* actions,
                                 const float* old_logprobs, const float* advantages,
                                 const float* returns_norm, int64_t n, float eps_clip, int k_epochs,
                                 float* grad_out, float* exp_avg, float* exp_avg_sq, int64_t first_step,
                                 double lr_actor, double lr_critic, double beta1, double beta2,
                                 double eps, float* losses_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (k_epochs < 1 || first_step < 1) return PPO_E_BADARG;
    const bool timed = g_timing.on && k_epochs <= PPO_TIMED_EPOCHS_MAX;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (timed) {
        g_timing.epochs = 0;
        if ((rc = timing_mark(0, st))) return rc;
    }
    for (int e = 0; e < k_epochs; ++e) {
        rc = ppo_epoch_fwdbwd(params, state_dim, action_dim, continuous, action_var, states, actions,
                              old_logprobs, advantages, returns_norm, n, n, eps_clip, nullptr,
                              nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
        if (rc) return rc;
        if (timed && (rc = timing_mark(2 * e + 1, st))) return rc;
        rc = ppo_reduce_adam_step(state_dim, action_dim, workspace, workspace_bytes, grad_out, params,
                                  exp_avg, exp_avg_sq, first_step + e, lr_actor, lr_critic, beta1,
                                  beta2, eps, losses_out ? losses_out + e : nullptr, stream);
        if (rc) return rc;
        if (timed && (rc = timing_mark(2 * e + 2, st))) return rc;
    }
    if (timed) g_timing.epochs = k_epochs;
    return 0;
}

extern "C" int ppo_hip_abi_version(void) { return 10; }

extern "C" const char* ppo_hip_strerror(int code) {
    switch (code) {
        case 0: return "success";
        case PPO_E_BADARG: return "ppo_hip: bad argument (null pointer, non-positive size or inconsistent shape)";
        case PPO_E_WORKSPACE: return "ppo_hip: workspace missing or smaller than ppo_workspace_bytes()";
        case PPO_E_UNSUPPORTED: return "ppo_hip: state_dim or action_dim beyond PPO_MAX_STATE_DIM / PPO_MAX_ACTION_DIM";
        case PPO_E_TOO_LARGE: return "ppo_hip: input larger than the fixed capacity of this allocation-free entry point (PPO_SCAN_NOALLOC_MAX elements)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ppo_hip: unknown error";
    }
}
#endif  // PPO_FWDBWD_SEEDED
