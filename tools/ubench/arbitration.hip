// Microbenchmark: how fast does a VALU wave progress when the other wave of its SIMD issues a stream
// of independent fp32 MFMAs (64 cycles each)?  Block = 8 waves: waves 0-3 run MFMAs for the whole
// kernel, waves 4-7 run N VALU instructions (a dependent chain, or 16 independent chains, at wave
// priority 0 or 3) and report their own elapsed cycles.
// build: hipcc -O3 --offload-arch=gfx950 arbitration.hip -o arbitration
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool INDEP, int PRIO, bool TRANS, bool VALU_OLD, int NACC>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int n_valu, int n_mfma) {
    const int wave = VALU_OLD ? ((threadIdx.x >> 6) + 4) & 7 : threadIdx.x >> 6;   // VALU_OLD: waves 0-3 are the VALU waves
    float s = 0;
    if (wave < 4) {
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n_mfma / 2; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)      // NACC independent accumulator chains, round-robin
                acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u % NACC], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) cyc[1024 + blockIdx.x * 4 + wave] = t1 - t0;
        for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    } else {
        __builtin_amdgcn_s_setprio(PRIO);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 1.0f + threadIdx.x * 1e-4f + i * 1e-2f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_valu / 16; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float& r = INDEP ? v[i] : v[0];
                r = TRANS ? __builtin_amdgcn_exp2f(r * 0.001f) : __builtin_fmaf(r, 0.9999f, 0.25f);
                asm volatile("" : "+v"(r));
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += v[i];
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave - 4] = t1 - t0;
        __builtin_amdgcn_s_setprio(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool INDEP, int PRIO, bool TRANS, bool VALU_OLD = false, int NACC = 2>
void run(const char* tag, int n_mfma) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2 * 256 * 4 * 8); (void)hipMemset(cyc, 0, 2 * 256 * 4 * 8);
    const int n_valu = 16000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<INDEP, PRIO, TRANS, VALU_OLD, NACC>), dim3(256), dim3(512), 0, 0, out, cyc, n_valu, n_mfma);
    (void)hipDeviceSynchronize();
    static unsigned long long h[2048]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, mm = 0; for (int i = 0; i < 1024; ++i) { m += h[i]; mm += h[1024 + i]; } m /= 1024; mm /= 1024;
    printf("%-52s partner %s: %6.1f cycles per VALU instruction", tag, n_mfma ? "streams MFMAs" : "idle         ", m / n_valu);
    if (n_mfma) printf(";  %5.1f cycles per MFMA over its whole stream", mm / (2.0 * n_mfma));
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<false, 0, false>("dependent v_fma chain, priority 0", 0);
    run<false, 0, false>("dependent v_fma chain, priority 0", 40000);
    run<false, 3, false>("dependent v_fma chain, priority 3", 40000);
    run<true, 0, false>("16 independent v_fma chains, priority 0", 0);
    run<true, 0, false>("16 independent v_fma chains, priority 0", 40000);
    run<true, 3, false>("16 independent v_fma chains, priority 3", 40000);
    run<false, 0, true>("dependent v_exp chain, priority 0", 0);
    run<false, 0, true>("dependent v_exp chain, priority 0", 40000);
    run<true, 3, true>("16 independent v_exp chains, priority 3", 40000);
    run<false, 0, false, true>("VALU waves are the OLDER ones: dependent, prio 0", 40000);
    run<true, 0, false, true>("VALU waves are the OLDER ones: independent, prio 0", 40000);
    run<true, 3, false, true>("VALU waves are the OLDER ones: independent, prio 3", 40000);
    run<true, 0, false, false, 1>("MFMA wave: ONE dependent accumulator chain; indep VALU", 40000);
    run<false, 0, false, false, 1>("MFMA wave: ONE dependent accumulator chain; dep VALU", 40000);
    run<true, 0, false, false, 2>("MFMA wave: 2 chains round-robin; indep VALU", 40000);
    run<true, 0, false, false, 4>("MFMA wave: 4 chains round-robin; indep VALU", 40000);
    run<true, 0, false, false, 2>("half as many MFMAs (total starvation => half)", 20000);
    run<true, 0, false, false, 2>("quarter as many MFMAs", 10000);
    return 0;
}
