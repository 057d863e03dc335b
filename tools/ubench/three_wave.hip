// Microbenchmark: the instruction mix of one fwd+bwd tile of the 64-hidden MLP (32 samples, fp32 MFMA
// 32x32x2), split into forward and backward waves as update.hip does, in two workgroup shapes:
//   NF = 1   8 waves: SIMD p hosts forward wave p and backward wave p + 4; two workgroup barriers per tile
//            (the shape of update.hip's pair kernel)
//   NF = 2  12 waves: SIMD p hosts TWO forward waves (p, p + 4; alternate tiles) and backward wave p + 8;
//            the hand-off is double-buffered (one buffer per forward wave), ONE workgroup barrier per tile
// Question: with a dense MFMA stream (backward wave) always there to fill the pipe and two forward waves
// whose vector work can overlap each other, how close does a tile get to the sum of its issue costs?
// No W2^T image in either shape (dgrad reads W2 column-wise), so that NF = 2 fits 160 KB.
// build: hipcc -O3 --offload-arch=gfx950 three_wave.hip -o three_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD 68
#define MAT (32 * LD)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float tanh4(float xs) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f), 1.0f); }

struct Img { const float* w2f; const float* w1t; const float* w3; };

// forward of one tile up to dout; keeps H1, H2
struct FwdState { f32x16 H1[2], H2[2]; float dout; };

__device__ __forceinline__ void fwd_part1(const Img& I, const float (&xf)[3], int c, int h, FwdState& s) {
    f32x16 z0 = {0}, z1 = {0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        z0 = mfma32(I.w1t[(2 * k + h) * 64 + c], xf[k], z0);
        z1 = mfma32(I.w1t[(2 * k + h) * 64 + 32 + c], xf[k], z1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s.H1[0][r] = tanh4(z0[r]); s.H1[1][r] = tanh4(z1[r]); }
}

__device__ __forceinline__ void fwd_part2(const Img& I, int c, int h, FwdState& s, f32x16 (&gW3v)[2]) {
    f32x16 y0 = {0}, y1 = {0};
    const float* row0 = I.w2f + c * LD + 4 * h;
    const float* row1 = row0 + 32 * LD;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(row0 + 32 * st + 8 * q);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(row1 + 32 * st + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y0 = mfma32(a0[j], s.H1[st][4 * q + j], y0);
                y1 = mfma32(a1[j], s.H1[st][4 * q + j], y1);
            }
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s.H2[0][r] = tanh4(y0[r]); s.H2[1][r] = tanh4(y1[r]); }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&I.w3[32 * t + 8 * q + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = fmaf(w[j], s.H2[t][4 * q + j], acc);
        }
    acc += __shfl_xor(acc, 32);
    float dout = acc;
#pragma unroll
    for (int i = 0; i < 12; ++i) dout = fmaf(dout, 0.999f, 0.001f);   // head stand-in: a dependent chain
    dout = __builtin_amdgcn_exp2f(dout * 0.01f) - 1.0f;
    s.dout = dout;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gW3v[t][r] = fmaf(dout, s.H2[t][r], gW3v[t][r]);
}

template <bool GB2>
__device__ __forceinline__ void fwd_handoff(const Img& I, int c, int h, const FwdState& s, float* dzs, float* h1s, f32x16 (&gb2)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&I.w3[32 * t + 8 * q + 4 * h]);
            f32x4 dz4, h4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * q + j;
                const float hv = s.H2[t][r];
                const float dz = (w[j] * s.dout) * (1.0f - hv * hv);
                if (GB2) gb2[t][r] += dz;
                dz4[j] = dz;
                h4[j] = s.H1[t][r];
            }
            *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 32 * t + 8 * q) = dz4;
            *reinterpret_cast<f32x4*>(h1s + c * LD + 4 * h + 32 * t + 8 * q) = h4;
        }
}

// backward of one tile from the hand-off buffers
template <bool SB2>
__device__ __forceinline__ void bwd_tile(const Img& I, int c, int h, int lane, float* dzs, const float* h1s,
                                         f32x16 (&gW2)[2][2], f32x4 (&gW1n)[4], float (&sb2)[2], const float (&xq)[8]) {
    const int c16 = lane & 15, q4 = lane >> 4;
    const float* zrow = dzs + c * LD + 4 * h;
    const float* hrow = h1s + c * LD + 4 * h;
    // dgrad: A = W2 read column-wise (W2[out = 8g + 4h + j][in = c, 32 + c]), B = dZ2 rows
    f32x16 d0 = {0}, d1 = {0};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(zrow + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wc = I.w2f + (8 * g + 4 * h + j) * LD + c;
            d0 = mfma32(wc[0], b4[j], d0);
            d1 = mfma32(wc[32], b4[j], d1);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(hrow + 8 * q);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(hrow + 32 + 8 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d0[4 * q + j] *= 1.0f - h0[j] * h0[j];
            d1[4 * q + j] *= 1.0f - h1[j] * h1[j];
        }
    }
    // wgrad 2 (column reads of the staged rows); the b2 gradient is the sum of its A operands
    const float* zcol = dzs + (16 * h) * LD + c;
    const float* hcol = h1s + (16 * h) * LD + c;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float a0 = zcol[s * LD], a1 = zcol[s * LD + 32], b0 = hcol[s * LD], b1 = hcol[s * LD + 32];
        if (SB2) { sb2[0] += a0; sb2[1] += a1; }
        gW2[0][0] = mfma32(a0, b0, gW2[0][0]);
        gW2[0][1] = mfma32(a0, b1, gW2[0][1]);
        gW2[1][0] = mfma32(a1, b0, gW2[1][0]);
        gW2[1][1] = mfma32(a1, b1, gW2[1][1]);
    }
    // dZ1 rows over dZ2's space, wgrad 1 on 16x16x4 tiles
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v0, v1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = d0[4 * q + j]; v1[j] = d1[4 * q + j]; }
        *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 8 * q) = v0;
        *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 32 + 8 * q) = v1;
    }
    const float* zc = dzs + q4 * LD + c16;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            gW1n[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[4 * s * LD + 16 * t4], xq[s], gW1n[t4], 0, 0, 0);
}

// stamps per wave: [0] cycles (s_memtime) of the loop, [1] 100 MHz ticks (s_memrealtime) of the loop
template <int NF>
__global__ __launch_bounds__((NF + 1) * 256) void k(const float* __restrict__ x, float* out, unsigned long long* stamps, int n_tiles) {
    extern __shared__ __align__(16) float smem[];
    float* w2f = smem;                  // [64][LD]
    float* w1t = w2f + 64 * LD;         // [8][64]
    float* w3 = w1t + 8 * 64;           // [64]
    float* stg = w3 + 64;               // per SIMD: NF x (dzs, h1s)
    constexpr int NW = (NF + 1) * 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int p = wave & 3, kind = wave >> 2;
    for (int i = threadIdx.x; i < 64 * LD + 8 * 64 + 64; i += blockDim.x) smem[i] = 0.01f * ((i * 37) % 101 - 50);
    __syncthreads();
    const Img I{w2f, w1t, w3};
    float* const pair_stg = stg + p * NF * 2 * MAT;
    const float* xrow = x + ((size_t)(blockIdx.x * 4 + p) * 32 + c) * 4;
    float acc_out = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (kind < NF) {
        // ---------------------------------------------------------------- forward wave `kind`
        f32x16 gW3v[2] = {{0}, {0}}, gb2[2] = {{0}, {0}};
        float xf[3] = {xrow[h], xrow[2 + h], h ? 0.f : 1.f};
        float* dzs = pair_stg + kind * 2 * MAT;
        float* h1s = dzs + MAT;
        FwdState s;
        if (NF == 1) {
            for (int t = 0; t < n_tiles; ++t) {
                fwd_part1(I, xf, c, h, s);
                fwd_part2(I, c, h, s, gW3v);
                __syncthreads();   // partner has finished reading the previous tile
                fwd_handoff<true>(I, c, h, s, dzs, h1s, gb2);
                __syncthreads();   // buffers complete
            }
        } else {
            // tile t (t % 2 == kind): part 1 before barrier t-1 (skipped for t = 0), part 2 before barrier t,
            // hand-off writes after barrier t; the backward wave reads them after barrier t+1
            int nb = 0;
            for (int t = kind; t < n_tiles; t += 2) {
                fwd_part1(I, xf, c, h, s);
                if (t > 0) { __syncthreads(); ++nb; }
                fwd_part2(I, c, h, s, gW3v);
                __syncthreads(); ++nb;
                fwd_handoff<false>(I, c, h, s, dzs, h1s, gb2);
            }
            for (; nb < n_tiles + 2; ++nb) __syncthreads();
        }
        for (int i = 0; i < 16; ++i) acc_out += gW3v[0][i] + gW3v[1][i] + gb2[0][i] + gb2[1][i];
    } else {
        // ---------------------------------------------------------------- backward wave
        f32x16 gW2[2][2] = {{{0}, {0}}, {{0}, {0}}};
        f32x4 gW1n[4] = {{0}, {0}, {0}, {0}};
        float sb2[2] = {0.f, 0.f};
        float xq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xq[i] = xrow[i & 3];
        if (NF == 1) {
            for (int t = 0; t < n_tiles; ++t) {
                __syncthreads();
                __syncthreads();
                bwd_tile<false>(I, c, h, lane, pair_stg, pair_stg + MAT, gW2, gW1n, sb2, xq);
            }
        } else {
            __syncthreads();
            __syncthreads();
            for (int t = 0; t < n_tiles; ++t) {   // step t + 2
                float* dzs = pair_stg + (t & 1) * 2 * MAT;
                bwd_tile<true>(I, c, h, lane, dzs, dzs + MAT, gW2, gW1n, sb2, xq);
                __syncthreads();
            }
        }
        for (int i = 0; i < 16; ++i) acc_out += gW2[0][0][i] + gW2[0][1][i] + gW2[1][0][i] + gW2[1][1][i];
        for (int i = 0; i < 4; ++i) acc_out += gW1n[0][i] + gW1n[1][i] + gW1n[2][i] + gW1n[3][i];
        acc_out += sb2[0] + sb2[1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        stamps[(blockIdx.x * NW + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * NW + wave) * 2 + 1] = r1 - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc_out;
}

template <int NF>
void run(int n_tiles, int launches) {
    constexpr int NW = (NF + 1) * 4;
    float *x, *out; unsigned long long* st;
    (void)hipMalloc(&x, 256 * 4 * 32 * 4 * 4);
    {
        static float hx[256 * 4 * 32 * 4];
        for (int i = 0; i < 256 * 4 * 32 * 4; ++i) hx[i] = (float)rand() / RAND_MAX - 0.5f;
        (void)hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
    }
    (void)hipMalloc(&out, 256 * NW * 64 * 4); (void)hipMalloc(&st, 256 * NW * 16);
    const size_t lds = (64 * LD + 8 * 64 + 64 + 4 * NF * 2 * MAT) * 4;
    (void)hipFuncSetAttribute((const void*)k<NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k<NF>), dim3(256), dim3(NW * 64), lds, 0, x, out, st, n_tiles);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NF>), dim3(256), dim3(NW * 64), lds, 0, x, out, st, n_tiles);
    (void)hipEventRecord(e1, 0);
    const hipError_t err = hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long hs[256 * 12 * 2]; (void)hipMemcpy(hs, st, 256 * NW * 16, hipMemcpyDeviceToHost);
    // per workgroup: the longest wave; then the mean over workgroups
    double cyc = 0, ghz = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long m = 0, rt = 1;
        for (int w = 0; w < NW; ++w) if (hs[(b * NW + w) * 2] > m) { m = hs[(b * NW + w) * 2]; rt = hs[(b * NW + w) * 2 + 1]; }
        cyc += (double)m; ghz += (double)m / (double)rt * 0.1;
    }
    cyc /= 256; ghz /= 256;
    // MFMA cycles per tile: (6 + 64 + 64 + 64) * 64 + 32 * 32 = 13696
    printf("NF=%d (%2d waves, LDS %6zu B, %s): %7.0f cycles per tile per SIMD, MFMA share %.3f, in-kernel clock %.2f GHz, launch %.3f ms = %.2f us per tile\n",
           NF, NW, lds, hipGetErrorString(err), cyc / n_tiles, 13696.0 * n_tiles / cyc, ghz, ms, ms * 1e3 / n_tiles);
    (void)hipFree(x); (void)hipFree(out); (void)hipFree(st);
}

int main(int argc, char** argv) {
    const int n_tiles = argc > 1 ? atoi(argv[1]) : 36;
    const int launches = argc > 2 ? atoi(argv[2]) : 400;
    run<1>(n_tiles, launches);
    run<2>(n_tiles, launches);
    run<1>(n_tiles, launches);
    run<2>(n_tiles, launches);
    return 0;
}
