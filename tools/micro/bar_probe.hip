// Developer probe: is device memory that the host can write directly (large BAR, fine-grained allocation) available on this box, and
// what does a host -> device -> host round trip cost through it, against pinned host memory (the resident one-line service's mailbox)?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/bar_probe.hip -o /tmp/bar_probe && /tmp/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <atomic>

__global__ void echo(volatile uint32_t* in, volatile uint32_t* out, uint32_t rounds) {
    uint32_t last = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t v;
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
            v = __hip_atomic_load(const_cast<uint32_t*>(in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return;   // 2 s: give up
        } while (v == last);
        last = v;
        __hip_atomic_store(const_cast<uint32_t*>(out), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double run(uint32_t* in_host_view, uint32_t* in_dev, uint32_t* out_host_view, uint32_t* out_dev, const char* what) {
    const uint32_t rounds = 20000;
    *in_host_view = 0; *out_host_view = 0;
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, 0, in_dev, out_dev, rounds);
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 1; r <= rounds; ++r) {
        *(volatile uint32_t*)in_host_view = r;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        while (*(volatile uint32_t*)out_host_view != r) { }
    }
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds;
    printf("%s: %.2f us per round trip\n", what, us);
    return us;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("isLargeBar %d\n", p.isLargeBar);
    uint32_t *h_in, *h_out;
    hipHostMalloc(&h_in, 4096, hipHostMallocMapped);
    hipHostMalloc(&h_out, 4096, hipHostMallocMapped);
    run(h_in, h_in, h_out, h_out, "mailbox and answer in pinned host memory");
    uint32_t* d_in = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&d_in), 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess && p.isLargeBar) {
        hipMemset(d_in, 0, 4096);
        hipDeviceSynchronize();
        run(d_in, d_in, h_out, h_out, "mailbox in device memory (written by the host through the BAR), answer in pinned host memory");
    }
    return 0;
}
