// gx_jsonl.hip -- result materialisation on the device: match ids + capture
// offsets -> one JSON object per matched line, i.e. ExtractionResult.asMap(idAs)
// (core/ExtractionResult.java:65-88) serialised the way a Jackson ObjectMapper writes a LinkedHashMap:
//   {"<idAs>":"<extraction name>","<extractor 1>":"<captured text>",...,<append entries>}\n
// A capture that Matcher.group() would report as null is written as null.  The key order, the
// "later put replaces the value but keeps the position" rule and the append entries are resolved on the host
// into one template per extraction (gx_api.cpp: build_jsonl_templates): a list of segments, each a literal byte
// string followed by an optional capture group.
//
// One wave per line, one lane per byte: pass 1 sums the escaped lengths, an exclusive scan turns the sizes into
// output offsets, pass 2 writes.  Bytes are Latin-1 code units (the batch path's input model) and leave as
// UTF-8; with utf8_passthrough the bytes >= 0x80 are copied as they are (input that was UTF-8 all along).
#include <cstdint>
#include <hip/hip_runtime.h>

#include "gx_device.hpp"

namespace gx {
namespace {

// escaped length of one byte inside a JSON string
__device__ __forceinline__ uint32_t esc_len(uint32_t b, bool passthrough) {
    if (b >= 0x80u) return passthrough ? 1u : 2u;
    if (b >= 0x20u) return (b == 0x22u || b == 0x5Cu) ? 2u : 1u;
    return (b == 0x08u || b == 0x09u || b == 0x0Au || b == 0x0Cu || b == 0x0Du) ? 2u : 6u;
}

__device__ __forceinline__ void esc_write(uint8_t* dst, uint32_t b, bool passthrough) {
    if (b >= 0x80u) {
        if (passthrough) { dst[0] = static_cast<uint8_t>(b); return; }
        dst[0] = static_cast<uint8_t>(0xC0u | (b >> 6));
        dst[1] = static_cast<uint8_t>(0x80u | (b & 0x3Fu));
        return;
    }
    if (b >= 0x20u) {
        if (b == 0x22u || b == 0x5Cu) { dst[0] = '\\'; dst[1] = static_cast<uint8_t>(b); }
        else dst[0] = static_cast<uint8_t>(b);
        return;
    }
    dst[0] = '\\';
    switch (b) {
    case 0x08u: dst[1] = 'b'; return;
    case 0x09u: dst[1] = 't'; return;
    case 0x0Au: dst[1] = 'n'; return;
    case 0x0Cu: dst[1] = 'f'; return;
    case 0x0Du: dst[1] = 'r'; return;
    default: break;
    }
    dst[1] = 'u'; dst[2] = '0'; dst[3] = '0';
    dst[4] = static_cast<uint8_t>('0' + (b >> 4));
    const uint32_t lo = b & 15u;
    dst[5] = static_cast<uint8_t>(lo < 10u ? '0' + lo : 'A' + (lo - 10u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= static_cast<uint32_t>(d)) v += o;
    }
    return v;
}

struct JsonlTemplates {
    const uint32_t* seg_off;    // [n_rules + 1] first segment of each extraction's template
    const uint32_t* lit_off;    // [n_segs] literal bytes of the segment in `lits`
    const uint32_t* lit_len;    // [n_segs]
    const int32_t* group;       // [n_segs] capture group written after the literal, or -1
    const uint32_t* fixed_len;  // [n_rules] sum of the template's literal lengths
    const uint8_t* lits;
};

// WRITE = false: sizes[i] = bytes of line i's JSON text (0 for lines without a match).
// WRITE = true:  the text goes to out + out_off[i].
template <typename OFF, bool WRITE>
__global__ void __launch_bounds__(256) k_jsonl(JsonlTemplates tm, const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n,
                                              const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots,
                                              int passthrough, uint32_t* __restrict__ sizes, const uint64_t* __restrict__ out_off,
                                              uint8_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    const bool pt = passthrough != 0;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const int32_t k = match_id[i];
        if (k < 0) {
            if (!WRITE && lane == 0) sizes[i] = 0;
            continue;
        }
        const uint8_t* line = data + static_cast<uint64_t>(off[i]);
        const int32_t* cp = caps + i * static_cast<uint64_t>(slots);
        uint32_t total = WRITE ? 0u : tm.fixed_len[k];
        uint8_t* dst = WRITE ? out + out_off[i] : nullptr;
        for (uint32_t s = tm.seg_off[k]; s < tm.seg_off[k + 1]; ++s) {
            if (WRITE) {
                const uint8_t* lit = tm.lits + tm.lit_off[s];
                const uint32_t ll = tm.lit_len[s];
                for (uint32_t q = lane; q < ll; q += 64u) dst[q] = lit[q];
                dst += ll;
            }
            const int32_t g = tm.group[s];
            if (g < 0) continue;
            const int32_t b = cp[2 * g], e = cp[2 * g + 1];
            if (b < 0) {
                if (WRITE) {
                    if (lane < 4u) dst[lane] = "null"[lane];
                    dst += 4;
                } else total += 4u;
                continue;
            }
            if (WRITE) {
                if (lane == 0) dst[0] = '"';
                ++dst;
            } else total += 2u;
            for (int32_t c0 = b; c0 < e; c0 += 64) {
                const int32_t p = c0 + static_cast<int32_t>(lane);
                const bool in = p < e;
                const uint32_t v = in ? line[p] : 0u;
                const uint32_t el = in ? esc_len(v, pt) : 0u;
                if (WRITE) {
                    const uint32_t inc = wave_inclusive(el, lane);
                    if (in) esc_write(dst + (inc - el), v, pt);
                    dst += __shfl(inc, 63);
                } else total += wave_sum(el);
            }
            if (WRITE) {
                if (lane == 0) dst[0] = '"';
                ++dst;
            }
        }
        if (!WRITE && lane == 0) sizes[i] = total;
    }
}

// ---- exclusive scan u32[n] -> u64[n + 1] (out[n] = total): block sums, one-workgroup scan, block scans ----
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr uint64_t SCAN_BLOCK = static_cast<uint64_t>(SCAN_THREADS) * SCAN_ITEMS;

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_block_sums(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SCAN_BLOCK;
    uint64_t t = 0;
#pragma unroll
    for (int it = 0; it < SCAN_ITEMS; ++it) {
        const uint64_t i = base + static_cast<uint64_t>(it) * SCAN_THREADS + threadIdx.x;
        if (i < n) t += in[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(static_cast<unsigned long long>(t), d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) s += wsum[w];
        block_sums[blockIdx.x] = s;
    }
}

// in place: block_sums[b] <- sum of the blocks before b; block_sums[nblocks] <- total
__global__ void __launch_bounds__(1024) k_scan_of_sums(uint64_t* __restrict__ block_sums, uint64_t nblocks) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint64_t b = b0 + threadIdx.x;
        const uint64_t v = b < nblocks ? block_sums[b] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(static_cast<unsigned long long>(inc), d);
            if (lane >= static_cast<uint32_t>(d)) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        const uint64_t c = carry;
        if (b < nblocks) block_sums[b] = c + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wbase + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_write(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_sums,
                                                             uint64_t nblocks, uint64_t* __restrict__ out) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    __shared__ uint64_t running;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SCAN_BLOCK;
    if (threadIdx.x == 0) running = block_sums[blockIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nblocks];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int it = 0; it < SCAN_ITEMS; ++it) {
        const uint64_t i = base + static_cast<uint64_t>(it) * SCAN_THREADS + threadIdx.x;
        const uint64_t v = i < n ? in[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(static_cast<unsigned long long>(inc), d);
            if (lane >= static_cast<uint32_t>(d)) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        const uint64_t r0 = running;
        if (i < n) out[i] = r0 + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) running = r0 + wbase + inc;
        __syncthreads();
    }
}

}  // namespace

size_t jsonl_workspace_bytes(uint64_t n) {
    const uint64_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return static_cast<size_t>(n * 4 + (nblocks + 2) * 8 + 64);
}

// Pass 1 + scan: line_out_off[0..n] (device, u64) receives the output offset of every line's text and, in
// [n], the total size.  workspace: jsonl_workspace_bytes(n).
hipError_t launch_jsonl_sizes(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, uint64_t* line_out_off, void* workspace,
                              hipStream_t stream) {
    if (b.n == 0) return hipMemsetAsync(line_out_off, 0, 8, stream);
    const uint64_t nblocks = (b.n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint32_t* sizes = static_cast<uint32_t*>(workspace);
    uint64_t* block_sums = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(workspace) + ((b.n * 4 + 15) & ~static_cast<uint64_t>(15)));
    JsonlTemplates t{tm.seg_off, tm.lit_off, tm.lit_len, tm.group, tm.fixed_len, tm.lits};
    uint64_t blocks = (b.n + 3) / 4;  // 4 waves (lines) per block
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    if (b.offsets64)
        hipLaunchKernelGGL((k_jsonl<uint64_t, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint64_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, sizes, nullptr, nullptr);
    else
        hipLaunchKernelGGL((k_jsonl<uint32_t, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint32_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, sizes, nullptr, nullptr);
    if (nblocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(static_cast<unsigned>(nblocks)), dim3(SCAN_THREADS), 0, stream, sizes, b.n, block_sums);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, stream, block_sums, nblocks);
    hipLaunchKernelGGL(k_scan_write, dim3(static_cast<unsigned>(nblocks)), dim3(SCAN_THREADS), 0, stream, sizes, b.n, block_sums, nblocks, line_out_off);
    return hipGetLastError();
}

// Pass 2: write the text.
hipError_t launch_jsonl_write(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, const uint64_t* line_out_off, uint8_t* out,
                              hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    JsonlTemplates t{tm.seg_off, tm.lit_off, tm.lit_len, tm.group, tm.fixed_len, tm.lits};
    uint64_t blocks = (b.n + 3) / 4;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    if (b.offsets64)
        hipLaunchKernelGGL((k_jsonl<uint64_t, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint64_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, nullptr, line_out_off, out);
    else
        hipLaunchKernelGGL((k_jsonl<uint32_t, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint32_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, nullptr, line_out_off, out);
    return hipGetLastError();
}

}  // namespace gx
