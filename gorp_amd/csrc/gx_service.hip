// gx_service.hip -- the resident one-line service: Gorp.extract(String) (core/Gorp.java:145-186) without a kernel launch per call.
//
// A launch and its synchronisation cost 15-16 us on this stack, the walk of a 41-character line 2 us: a caller that keeps
// calling Gorp.extract(String) line by line pays for launches.  With GX_CREATE_RESIDENT_ONE the handle keeps ONE wave resident
// while such calls keep coming: the tables in LDS (the tile kernel's image: dense rows), the wave polls a mailbox in pinned
// host memory, stages the line it finds there, walks it -- the tile kernel's walk<>, the same tables, the same result -- and
// writes the answer back into pinned host memory.  The host's call is a few stores and a spin.
//
// Mailbox (host -> device), 64-byte cache lines, each written payload first and its tag last, so that a line that is read with
// its tag equal to the request's sequence number is the request's:
//   line 0: u32 seq | u16 length in bytes | u16 flags (bit 0: leave) | 56 bytes of text
//   line j: u32 seq | 60 bytes of text                                  (j = 1 .. 16: lines of up to 1 016 bytes)
// Answer (device -> host): i32 match id | i32 caps[2 G] ... | u32 seq last, behind a system-scope fence.
//
// The wave never stays for ever: it leaves when no request has come for `idle_ticks` (100 MHz ticks; the host starts it again
// with the next call that finds it gone: GxService::state), when it has been resident for `life_ticks` (so that a
// hipDeviceSynchronize elsewhere in the process waits milliseconds, not for ever), or when the host says so (gx_destroy).
// Before it leaves it announces it (state := GONE) and looks at the mailbox once more: a request that crossed the announcement is
// still answered.
#include "gx_tile_body.hpp"

namespace gx {

namespace {

__device__ __forceinline__ uint32_t sys_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// MODE 0: the match automaton alone (a definition without capture regexps); 1: the fused automaton, "register := position" programs
template <int MODE>
__global__ void __launch_bounds__(64)
k_one_service(GxLds L, const uint8_t* __restrict__ image, const uint32_t* __restrict__ mailbox, int32_t* __restrict__ answer, uint32_t* __restrict__ state,
              uint32_t last, int max_groups, unsigned long long idle_ticks, unsigned long long life_ticks) {
    {
        extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];
        const uint4* src = reinterpret_cast<const uint4*>(image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    WalkTab W;
    W.at = nullptr;
    W.row_bytes = L.row_bytes;
    W.ops_off = L.ops_off;
    W.ops = L.ops;
    W.acc_tab = L.acc_tab;
    W.ncls = L.ncls;
    W.indexed = L.rec_indexed;
    const uint32_t stage = L.stage;
    const uint32_t regs = L.regs + 128u + lane * 2u;
    const int G = max_groups;
    const unsigned long long born = __builtin_amdgcn_s_memrealtime();
    unsigned long long idle_since = born;
    bool announced = false;
    for (;;) {
        // ---- the mailbox's first cache line: sequence number, length, flags, the first 56 bytes ----
        uint32_t w = lane < 16u ? sys_load(mailbox + lane) : 0u;
        const uint32_t seq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 0));
        const uint32_t head = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 1));
        if (seq != last) {
            const uint32_t len = head & 0xFFFFu;
            if (announced) { if (lane == 0) __hip_atomic_store(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); announced = false; }
            if (lane >= 2u && lane < 16u) lds_st<uint32_t>(stage + 4u * (lane - 2u), w);
            if (len > 56u) {
                // the other cache lines, four per sweep: a line whose tag is not the request's yet is read again
                const uint32_t lines = (len - 56u + 59u) / 60u;   // 1 .. 16
                for (uint32_t c0 = 0; c0 < lines; c0 += 4u) {
                    const uint32_t cl = c0 + (lane >> 4), d = lane & 15u;
                    const bool mine = cl < lines;
                    for (;;) {
                        const uint32_t v = mine ? sys_load(mailbox + 16u * (1u + cl) + d) : seq;
                        const bool tag_ok = d != 0u || v == seq;
                        if (__builtin_amdgcn_ballot_w64(!tag_ok) == 0ull) {
                            if (mine && d != 0u) lds_st<uint32_t>(stage + 56u + 60u * cl + 4u * (d - 1u), v);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- the tile kernel's walk on the one staged line (lane 0's; the other lanes idle along) ----
            const bool on = lane == 0u;
            const uint32_t out = stage + L.stage_bytes - 272u;   // (the answer's words, behind the longest line: one store for all of them below)
            int32_t result;
            if (MODE == 0) {
                const uint32_t mrow = walk<TIER_LDS, false, false>(W, stage, 0u, false, L.m_start, 0u, on ? len : 0u, on, L.m_dead, regs);
                result = state_info<TIER_LDS>(W, mrow);
                if (on) lds_st<uint32_t>(out, static_cast<uint32_t>(result));
            } else {
                const uint32_t urow = walk<TIER_LDS, true, true>(W, stage, 0u, false, L.u_start, 0u, on ? len : 0u, on, L.u_dead, regs);
                const int32_t info = state_info<TIER_LDS>(W, urow);
                result = line_result<TIER_LDS>(info, L.fin_tags, nullptr, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                    if (on) lds_st<u32x2>(out + 4u + 8u * g, u32x2{static_cast<uint32_t>(pb), static_cast<uint32_t>(pe)});
                });
                if (on) lds_st<uint32_t>(out, static_cast<uint32_t>(result));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 1u + 2u * static_cast<uint32_t>(MODE == 0 ? 0 : G)) answer[lane] = static_cast<int32_t>(lds_ld<uint32_t>(out + 4u * lane));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the answer before its sequence number
            if (on) __hip_atomic_store(reinterpret_cast<uint32_t*>(answer) + 1 + 2 * max_groups, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            last = seq;
            idle_since = __builtin_amdgcn_s_memrealtime();
            if (idle_since - born > life_ticks) {   // long enough: the next call starts a fresh wave
                if (lane == 0) __hip_atomic_store(state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                announced = true;
                continue;   // (one more look at the mailbox, below, then out)
            }
            continue;
        }
        if (announced) break;   // announced, looked once more, nothing: gone
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if ((head >> 16) & 1u || now - idle_since > idle_ticks || now - born > life_ticks) {
            if (lane == 0) __hip_atomic_store(state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
            announced = true;
            continue;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

}  // namespace

// lds: the handle's dense-rows image laid out for one wave (gx_api.cpp: plan_service); mode 0 / 1 as above
hipError_t launch_one_service(int mode, const GxLds& lds, const uint8_t* lds_image, const uint32_t* mailbox, int32_t* answer, uint32_t* state,
                              uint32_t last_seq, int max_groups, unsigned long long idle_ticks, unsigned long long life_ticks, hipStream_t stream) {
    if (mode == 0) {
        hipError_t e = allow_full_lds(&k_one_service<0>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_one_service<0>), dim3(1), dim3(64), lds.total_bytes, stream, lds, lds_image, mailbox, answer, state, last_seq, max_groups, idle_ticks, life_ticks);
    } else {
        hipError_t e = allow_full_lds(&k_one_service<1>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_one_service<1>), dim3(1), dim3(64), lds.total_bytes, stream, lds, lds_image, mailbox, answer, state, last_seq, max_groups, idle_ticks, life_ticks);
    }
    return hipGetLastError();
}

}  // namespace gx
