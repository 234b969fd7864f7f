// Developer microbenchmark: how fast can every wave of the chip draw tickets from counters in global memory?
// (a) one counter for the whole grid, (b) one per workgroup; agent scope, value returned; 256 workgroups x 11 waves x K draws.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void __launch_bounds__(704) k_draw(uint32_t* ctr, uint32_t stride_words, int K, int spin, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* c = ctr + static_cast<uint64_t>(blockIdx.x) * stride_words;
    uint32_t acc = 0;
    for (int k = 0; k < K; ++k) {
        uint32_t j = 0;
        if (lane == 0) j = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += __builtin_amdgcn_readfirstlane(j);
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);   // stand-in for a tile's work
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main() {
    uint32_t* ctr; uint32_t* sink;
    hipMalloc(&ctr, 256 * 4096); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 100}) for (uint32_t stride : {0u, 16u, 1024u}) for (int K : {55, 550}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(ctr, 0, 256 * 4096);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_draw, dim3(256), dim3(704), 0, 0, ctr, stride, K, spin, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("spin %3d  stride %4u words  K %3d: %.1f us  -> %.1f ns per draw of the whole chip (%d draws)\n", spin, stride, K, best * 1e3, best * 1e6 / (2816.0 * K), 2816 * K);
    }
    return 0;
}
