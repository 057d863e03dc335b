// Microbenchmark: does VALU work issued between fp32 MFMAs (v_mfma_f32_32x32x2_f32) by the SAME
// wave hide under the MFMA's 64 cycles, and how does that change with 1 vs 2 waves per SIMD?
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FILL, bool TRANS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                if (TRANS && (f % 4 == 3)) v[f % 8] = __builtin_amdgcn_exp2f(v[f % 8]);
                else v[f % 8] = __builtin_fmaf(v[f % 8], 1.0001f, 0.5f);
                asm volatile("" : "+v"(v[f % 8]));
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                if (TRANS && (f % 4 == 3)) v[(f + 4) % 8] = __builtin_amdgcn_exp2f(v[(f + 4) % 8]);
                else v[(f + 4) % 8] = __builtin_fmaf(v[(f + 4) % 8], 0.9999f, 0.25f);
                asm volatile("" : "+v"(v[(f + 4) % 8]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL, bool TRANS, int threads>
void run(const char* tag) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL((k<FILL, TRANS, threads>), dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1, 0); if (hipGetLastError() != hipSuccess) printf("launch failed\n"); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    hipDeviceSynchronize();
    const double flop = 256.0 * (threads / 64) * iters * 16.0 * 4096.0;
    printf("   wall %.3f ms -> %.1f TFLOP/s MFMA; ", ms, flop / (ms * 1e-3) / 1e12);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%s fill=%2d trans=%d threads=%d : %.1f cycles per MFMA (s_memtime ticks)\n", tag, FILL, (int)TRANS, threads, m / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, false, 256>("1w/simd"); run<4, false, 256>("1w/simd"); run<8, false, 256>("1w/simd"); run<16, false, 256>("1w/simd");
    run<16, true, 256>("1w/simd");
    run<0, false, 512>("2w/simd"); run<4, false, 512>("2w/simd"); run<8, false, 512>("2w/simd"); run<12, false, 512>("2w/simd");
    run<16, false, 512>("2w/simd"); run<24, false, 512>("2w/simd"); run<16, true, 512>("2w/simd");
    run<0, false, 1024>("4w/simd"); run<8, false, 1024>("4w/simd"); run<16, false, 1024>("4w/simd"); run<24, false, 1024>("4w/simd");
    return 0;
}
