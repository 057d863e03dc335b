// Question: can the CPU store directly into device memory (large BAR), so that a waiting kernel
// polls LOCAL memory instead of reading host memory over PCIe?  Tries fine-grained device memory
// (hipExtMallocWithFlags) and plain hipMalloc memory; a SIGSEGV handler reports "not accessible".
// build: hipcc -O3 --offload-arch=gfx950 host_store_to_vram.hip -o host_store_to_vram
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <ctime>
#include <immintrin.h>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

__global__ void waiter(volatile unsigned* box, unsigned* reply_host, unsigned want, unsigned limit) {
    unsigned it = 0;
    while (__hip_atomic_load((unsigned*)box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want && ++it < limit) {}
    __hip_atomic_store(reply_host, it < limit ? want : 0xdeadu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static void try_kind(const char* name, unsigned* dev, bool flush = false) {
    unsigned* reply;
    hipHostMalloc((void**)&reply, 64, hipHostMallocDefault);
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1)) { printf("%s: CPU store faults (not host-accessible)\n", name); return; }
    *(volatile unsigned*)dev = 0;   // may fault
    double tot = 0; int n = 200, bad = 0;
    for (int i = 1; i <= n; ++i) {
        *reply = 0;
        waiter<<<1, 64>>>(dev, reply, (unsigned)i, 1u << 22);
        timespec ts{0, 200000}; nanosleep(&ts, nullptr);          // let it start polling
        const double t0 = now_us();
        __atomic_store_n(dev, (unsigned)i, __ATOMIC_RELEASE);
        if (flush) _mm_sfence();                                   // a write-combining mapping holds the store back otherwise
        while (__atomic_load_n(reply, __ATOMIC_ACQUIRE) == 0) {}
        tot += now_us() - t0;
        bad += *reply != (unsigned)i;
        hipDeviceSynchronize();
    }
    printf("%s: CPU store -> kernel sees it -> reply visible on host: %.2f us round trip (%d timeouts)\n", name, tot / n, bad);
}

int main() {
    unsigned *fine = nullptr, *coarse = nullptr, *host = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fine, 4096, hipDeviceMallocFinegrained);
    printf("fine-grained alloc: %s\n", hipGetErrorString(e));
    hipMalloc((void**)&coarse, 4096);
    hipHostMalloc((void**)&host, 4096, hipHostMallocDefault);
    try_kind("page-locked host memory (kernel polls over PCIe)", host);
    if (e == hipSuccess) try_kind("fine-grained device memory", fine);
    try_kind("hipMalloc device memory", coarse);
    if (e == hipSuccess) try_kind("fine-grained device memory, sfence after the store", fine, true);
    try_kind("hipMalloc device memory, sfence after the store", coarse, true);
    try_kind("page-locked host memory, sfence after the store", host, true);
    return 0;
}
