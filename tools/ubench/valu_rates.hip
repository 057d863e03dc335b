// Microbenchmark: issue cost (cycles per wave-instruction, per SIMD) of the VALU instruction kinds the
// PPO kernels use next to fp32 MFMA: v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, and
// v_mfma_f32_32x32x2_f32 itself, at 1, 2 and 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, unsigned long long* cyc, int iters) {
    float v[16];
    f32x2 p[8];
    f32x16 acc = {0};
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + threadIdx.x * 1e-4f + i * 1e-2f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{v[i], v[i + 8]};
    const f32x2 m2 = {0.999f, 1.001f}, a2 = {0.25f, 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) v[i] = __builtin_fmaf(v[i], 0.9999f, 0.25f);
                if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
                if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(v[i]);
                if (MODE == 3) p[i & 7] = __builtin_elementwise_fma(p[i & 7], m2, a2);
                if (MODE == 4) { v[i] = __builtin_amdgcn_exp2f(v[i]); v[i] = __builtin_fmaf(v[i], 0.5f, 0.25f); }
                if (MODE == 5) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], v[15 - i], acc, 0, 0, 0);
                if (MODE == 6) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], v[15 - i], acc, 0, 0, 0); v[i] = __builtin_amdgcn_exp2f(v[i]); }
                if (MODE == 7) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0][0], p[1][1], acc, 0, 0, 0); v[i] = __builtin_amdgcn_exp2f(v[i]); v[(i + 8) & 15] = __builtin_amdgcn_rcpf(v[(i + 8) & 15]); }
                if (MODE == 8) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0][0], p[1][1], acc, 0, 0, 0); v[i] = __builtin_fmaf(v[i], 0.9999f, 0.25f); v[(i + 8) & 15] = __builtin_fmaf(v[(i + 8) & 15], 0.9999f, 0.25f); }
                if (MODE == 7 || MODE == 8) asm volatile("" : "+v"(v[(i + 8) & 15]));
                if (MODE != 3 && MODE != 5) asm volatile("" : "+v"(v[i]));
                if (MODE == 3) asm volatile("" : "+v"(p[i & 7]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int threads>
void run(const char* tag) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL((k<MODE, threads>), dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    hipDeviceSynchronize();
    const double groups = (double)iters * 64;                 // loop bodies (1 or 2 instructions each) per wave
    const double waves_per_simd = threads / 256.0;
    // wall-clock cycles per loop body per SIMD, at the kernel's own clock estimate of 2.4 GHz max; s_memtime ticks are reported raw
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%-28s %s: wall %.3f ms; %.2f ns per body per SIMD; %.2f memtime ticks per body per wave\n", tag,
           threads == 256 ? "1w/simd" : threads == 512 ? "2w/simd" : "4w/simd", ms,
           ms * 1e6 / (groups * waves_per_simd), m / groups);
    hipFree(out); hipFree(cyc);
}

template <int MODE> void all(const char* tag) { run<MODE, 256>(tag); run<MODE, 512>(tag); run<MODE, 1024>(tag); }

int main() {
    all<0>("v_fma_f32");
    all<3>("v_pk_fma_f32");
    all<1>("v_exp_f32");
    all<2>("v_rcp_f32");
    all<4>("v_exp_f32 + v_fma_f32");
    all<5>("v_mfma_f32_32x32x2_f32");
    all<6>("mfma + v_exp_f32 (dependent)");
    all<7>("mfma + exp + rcp (indep.)");
    all<8>("mfma + 2 v_fma (indep.)");
    return 0;
}
