// Question: in what order do the fp32 MFMA instructions add their k-slices?  One 16x16 output block
// with K = 64 is computed three ways from the same operands:
//   (a) v_mfma_f32_32x32x2_f32, 32 instructions, k = 2 i + (lane >> 5)      (rows/cols >= 16 zero)
//   (b) v_mfma_f32_16x16x4_f32, 16 instructions, k = 4 i + (lane >> 4)
//   (c) host: acc = fmaf(a[k], b[k], acc) for k = 0..63 (one rounding per k, ascending)
// and the three are compared bit for bit.  Equal (a) and (b) mean a kernel may change its tile
// shape between the two instructions without changing a single result bit.
// build: hipcc -O3 --offload-arch=gfx950 mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* A, const float* B, float* d32, float* d16) {  // A [16][64], B [64][16]
    const int lane = threadIdx.x;
    {
        const int m = lane & 31, h = lane >> 5;
        f32x16 c = {0};
        for (int i = 0; i < 32; ++i) {
            const int kk = 2 * i + h;
            const float a = m < 16 ? A[m * 64 + kk] : 0.f;
            const float b = m < 16 ? B[kk * 16 + m] : 0.f;
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * h + (r & 3);
            if (row < 16 && m < 16) d32[row * 16 + m] = c[r];
        }
    }
    {
        const int m = lane & 15, g = lane >> 4;
        f32x4 c = {0};
        for (int i = 0; i < 16; ++i) {
            const int kk = 4 * i + g;
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 64 + kk], B[kk * 16 + m], c, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) d16[(4 * g + r) * 16 + m] = c[r];
    }
}

int main() {
    float hA[16 * 64], hB[64 * 16], h32[256], h16[256], ref[256];
    srand(12345);
    int bad_ab = 0, bad_ac = 0, trials = 200;
    float *A, *B, *d32, *d16;
    hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&d32, sizeof h32); hipMalloc(&d16, sizeof h16);
    for (int t = 0; t < trials; ++t) {
        for (float& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (float& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (t % 3 == 0 ? 100.f : 1.f);
        hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice);
        hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
        k<<<1, 64>>>(A, B, d32, d16);
        hipMemcpy(h32, d32, sizeof h32, hipMemcpyDeviceToHost);
        hipMemcpy(h16, d16, sizeof h16, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                float acc = 0.f;
                for (int kk = 0; kk < 64; ++kk) acc = fmaf(hA[m * 64 + kk], hB[kk * 16 + n], acc);
                ref[m * 16 + n] = acc;
            }
        bad_ab += memcmp(h32, h16, sizeof h32) != 0;
        bad_ac += memcmp(h32, ref, sizeof h32) != 0;
    }
    printf("trials %d: 32x32x2 != 16x16x4 in %d, 32x32x2 != sequential fmaf in %d\n", trials, bad_ab, bad_ac);
    return 0;
}
