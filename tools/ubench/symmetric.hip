// Microbenchmark: the instruction mix of one fwd+bwd tile of the 64-hidden MLP (32 samples, fp32
// MFMA 32x32x2, weights as LDS A operands, activations chained through registers, weight gradients
// through a wave-private LDS staging area) run by SYMMETRIC waves: every wave executes the whole
// tile - forward, head stand-in, dgrad, both weight gradients - in one instruction stream, with NO
// workgroup barrier in the loop.  Reports cycles per tile and SIMD at 1 and 2 waves per SIMD, to be
// compared with the pipe-bound sum of issue costs (~16.5 k cycles for this mix).
// build: hipcc -O3 --offload-arch=gfx950 symmetric.hip -o symmetric
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD 68

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float tanh4(float xs) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f), 1.0f); }

template <int WAVES, bool SLIM, int LOCK>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(const float* __restrict__ x, float* out, unsigned long long* cyc, int n_tiles) {
    extern __shared__ __align__(16) float smem[];
    float* w2f = smem;                 // [64][LD]
    float* w2b = SLIM ? w2f : w2f + 64 * LD;        // [64][LD] (SLIM: aliased, to fit two waves per SIMD)
    float* w1t = w2f + (SLIM ? 1 : 2) * 64 * LD;        // [8][64]
    float* w3 = w1t + 8 * 64;          // [64]
    float* stg = w3 + 64;              // per wave: dz [32][LD], h1 [32][LD]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    for (int i = threadIdx.x; i < 2 * 64 * LD + 8 * 64 + 64; i += blockDim.x) smem[i] = 0.01f * ((i * 37) % 101 - 50);
    float* dzs = stg + wave * 2 * 32 * LD;
    float* h1s = dzs + 32 * LD;
    __syncthreads();
    f32x16 gW2[2][2] = {{{0}, {0}}, {{0}, {0}}};
    f32x4 gW1n[4] = {{0}, {0}, {0}, {0}};
    f32x16 gb2[2] = {{0}, {0}}, gW3v[2] = {{0}, {0}};
    float slim_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* xrow = x + ((size_t)(blockIdx.x * WAVES + wave) * 32 + c) * 4;
    float xf[3] = {xrow[h], xrow[2 + h], h ? 0.f : 1.f};
    const int c16 = lane & 15, q4 = lane >> 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n_tiles; ++it) {
        if (LOCK & 8) __syncthreads();
        // ---- layer 1 + tanh
        f32x16 z0 = {0}, z1 = {0};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            z0 = mfma32(w1t[(2 * s + h) * 64 + c], xf[s], z0);
            z1 = mfma32(w1t[(2 * s + h) * 64 + 32 + c], xf[s], z1);
        }
        f32x16 H1[2], H2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { H1[0][r] = tanh4(z0[r]); H1[1][r] = tanh4(z1[r]); }
        if (LOCK & 1) __syncthreads();
        // ---- layer 2 + tanh
        f32x16 y0 = {0}, y1 = {0};
        {
            const float* row0 = w2f + c * LD + 4 * h;
            const float* row1 = row0 + 32 * LD;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(row0 + 32 * st + 8 * q);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(row1 + 32 * st + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y0 = mfma32(a0[j], H1[st][4 * q + j], y0);
                        y1 = mfma32(a1[j], H1[st][4 * q + j], y1);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { H2[0][r] = tanh4(y0[r]); H2[1][r] = tanh4(y1[r]); }
        // ---- output layer (1 output) + head stand-in (~40 dependent VALU incl. 2 transcendental)
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&w3[32 * t + 8 * q + 4 * h]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = fmaf(w[j], H2[t][4 * q + j], acc);
            }
        acc += __shfl_xor(acc, 32);
        float dout = acc;
#pragma unroll
        for (int i = 0; i < 12; ++i) dout = fmaf(dout, 0.999f, 0.001f);
        dout = __builtin_amdgcn_exp2f(dout * 0.01f) - 1.0f;
        // ---- dZ2 = W3^T dout (1 - H2^2); per-lane sums of b2 and W3 gradients; stage dZ2 and H1 as rows
        f32x16 dZ[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&w3[32 * t + 8 * q + 4 * h]);
                f32x4 dz4, h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * q + j;
                    const float hv = H2[t][r];
                    const float dz = (w[j] * dout) * (1.0f - hv * hv);
                    if (SLIM) {
                        slim_acc[r & 3] += dz;
                        slim_acc[4 + (r & 3)] = fmaf(dout, hv, slim_acc[4 + (r & 3)]);
                    } else {
                        gb2[t][r] += dz;
                        gW3v[t][r] = fmaf(dout, hv, gW3v[t][r]);
                    }
                    dZ[t][r] = dz;
                    dz4[j] = dz;
                    h4[j] = H1[t][r];
                }
                *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 32 * t + 8 * q) = dz4;
                *reinterpret_cast<f32x4*>(h1s + c * LD + 4 * h + 32 * t + 8 * q) = h4;
            }
        if (LOCK & 2) __syncthreads();
        // ---- dgrad: dH1 = W2^T dZ2 (B operand straight from registers)
        f32x16 d0 = {0}, d1 = {0};
        {
            const float* row0 = w2b + c * LD + 4 * h;
            const float* row1 = row0 + 32 * LD;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(row0 + 32 * st + 8 * q);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(row1 + 32 * st + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        d0 = mfma32(a0[j], dZ[st][4 * q + j], d0);
                        d1 = mfma32(a1[j], dZ[st][4 * q + j], d1);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d0[r] *= 1.0f - H1[0][r] * H1[0][r];
            d1[r] *= 1.0f - H1[1][r] * H1[1][r];
        }
        if (LOCK & 4) __syncthreads();
        // ---- wgrad 2: dW2 += dZ2^T H1 (column reads of the staged rows)
        {
            const float* zcol = dzs + (16 * h) * LD + c;
            const float* hcol = h1s + (16 * h) * LD + c;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a0 = zcol[s * LD], a1 = zcol[s * LD + 32], b0 = hcol[s * LD], b1 = hcol[s * LD + 32];
                gW2[0][0] = mfma32(a0, b0, gW2[0][0]);
                gW2[0][1] = mfma32(a0, b1, gW2[0][1]);
                gW2[1][0] = mfma32(a1, b0, gW2[1][0]);
                gW2[1][1] = mfma32(a1, b1, gW2[1][1]);
            }
        }
        // ---- dZ1 rows over dZ2's space, wgrad 1 on 16x16x4 tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = d0[4 * q + j]; v1[j] = d1[4 * q + j]; }
            *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 8 * q) = v0;
            *reinterpret_cast<f32x4*>(dzs + c * LD + 4 * h + 32 + 8 * q) = v1;
        }
        {
            const float* zc = dzs + q4 * LD + c16;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    gW1n[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[4 * s * LD + 16 * t4], xf[s % 3], gW1n[t4], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cyc[(blockIdx.x * WAVES + wave) * 2] = t1 - t0; cyc[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0; }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += gW2[0][0][i] + gW2[0][1][i] + gW2[1][0][i] + gW2[1][1][i] + gb2[0][i] + gb2[1][i] + gW3v[0][i] + gW3v[1][i];
    for (int i = 0; i < 8; ++i) s += slim_acc[i];
    for (int i = 0; i < 4; ++i) s += gW1n[0][i] + gW1n[1][i] + gW1n[2][i] + gW1n[3][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, bool SLIM, int LOCK>
void run(int n_tiles) {
    float *x, *out; unsigned long long* cyc;
    (void)hipMalloc(&x, 256 * 8 * 32 * 4 * 4); { static float hx[256 * 8 * 32 * 4]; for (int i = 0; i < 256 * 8 * 32 * 4; ++i) hx[i] = (float)rand() / RAND_MAX - 0.5f; (void)hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice); }
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 16);
    const size_t lds = ((SLIM ? 1 : 2) * 64 * LD + 8 * 64 + 64 + WAVES * 2 * 32 * LD) * 4;
    (void)hipFuncSetAttribute((const void*)k<WAVES, SLIM, LOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<WAVES, SLIM, LOCK>), dim3(256), dim3(WAVES * 64), lds, 0, x, out, cyc, n_tiles);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<WAVES, SLIM, LOCK>), dim3(256), dim3(WAVES * 64), lds, 0, x, out, cyc, n_tiles);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long hc[4096]; (void)hipMemcpy(hc, cyc, 256 * WAVES * 16, hipMemcpyDeviceToHost);
    // per workgroup the longest wave (the SIMD is busy until its last wave ends), mean over workgroups
    double m = 0, mean = 0, ghz = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long mx = 0, rt = 1;
        for (int w = 0; w < WAVES; ++w) { const unsigned long long v = hc[(b * WAVES + w) * 2]; mean += v; if (v > mx) { mx = v; rt = hc[(b * WAVES + w) * 2 + 1]; } }
        m += mx; ghz += (double)mx / rt * 0.1;
    }
    m /= 256; mean /= 256 * WAVES; ghz /= 256;
    const double per_tile_simd = m / n_tiles / (WAVES / 4);
    // useful MFMA cycles per tile: (6 + 64 + 64 + 64) * 64 + 32 * 32 = 13696
    printf("lock %d: %s%d waves per SIMD, LDS %zu B: %8.0f cycles per tile per wave, %8.0f cycles per tile per SIMD; MFMA share %.3f; launch %.3f ms = %.2f us per tile and SIMD; in-kernel clock %.2f GHz; mean wave / longest wave %.3f\n",
           LOCK, SLIM ? "(slim) " : "", WAVES / 4, lds, m / n_tiles, per_tile_simd, 13696.0 / per_tile_simd, ms, ms * 1e3 / n_tiles / (WAVES / 4), ghz, mean / m);
    (void)hipFree(x); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<4, true, 0>(36);
    run<8, true, 0>(18);
    run<8, true, 1>(18);
    run<8, true, 2>(18);
    run<8, true, 3>(18);
    run<8, true, 7>(18);
    run<8, true, 15>(18);
    run<8, true, 10>(18);
    return 0;
}
