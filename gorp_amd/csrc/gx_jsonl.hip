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
// output offsets, pass 2 writes.  Bytes are Latin-1 code units
// (the batch path's input model) and leave as UTF-8; with utf8_passthrough the bytes >= 0x80 are copied as they are (input that was UTF-8 all along).
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

// (the builtin returns int: every result is cast to uint32_t before it is widened -- an int with bit 31 set would
// sign-extend into the high half of a 64-bit offset)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v)); }
__device__ __forceinline__ int32_t uni(int32_t v) { return static_cast<int32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v))); }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
    return (static_cast<uint64_t>(uni(static_cast<uint32_t>(v >> 32))) << 32) | static_cast<uint64_t>(uni(static_cast<uint32_t>(v)));
}

__device__ __forceinline__ int32_t lane_of(int32_t v, uint32_t l) {  // value held by lane l (l uniform)
    return static_cast<int32_t>(__builtin_amdgcn_readlane(static_cast<uint32_t>(v), static_cast<int>(l)));
}

// A line's template, one segment per lane (lanes >= nseg hold empty segments).  Loaded in three rounds of
// independent loads (extraction id + capture offsets | template bounds | segments) instead of a chain of
// dependent loads per segment: the kernels are latency-bound, not instruction-bound.
struct LineSegs {
    int32_t k;        // extraction, < 0: no text for this line
    uint32_t nseg;    // uniform
    uint32_t fixed;   // sum of literal lengths (uniform)
    uint32_t ll, lo;  // this lane's segment: literal length, literal offset in lits
    int32_t g, b, e;  // its capture group (-1: none) and the group's offsets (b < 0: null)
};

__device__ __forceinline__ LineSegs load_line_segs(const JsonlTemplates& tm, const int32_t* __restrict__ match_id,
                                                   const int32_t* __restrict__ caps, int slots, uint64_t i, uint32_t lane) {
    LineSegs L;
    // round 1: the extraction and (lane j: slots j and 64 + j) the capture offsets of the line
    const int32_t* cp = caps + i * static_cast<uint64_t>(slots);
    const int32_t kv = match_id[i];
    const int32_t cap_a = static_cast<int>(lane) < slots ? cp[lane] : -1;
    const int32_t cap_b = static_cast<int>(lane) + 64 < slots ? cp[lane + 64] : -1;
    L.k = uni(kv);
    L.nseg = 0; L.fixed = 0; L.ll = 0; L.lo = 0; L.g = -1; L.b = -1; L.e = -1;
    if (L.k < 0) return L;
    // round 2: template bounds
    const uint32_t s0v = tm.seg_off[L.k], s1v = tm.seg_off[L.k + 1], fv = tm.fixed_len[L.k];
    const uint32_t s0 = uni(s0v);
    L.nseg = uni(s1v) - s0;
    L.fixed = uni(fv);
    // round 3: one segment per lane
    if (lane < L.nseg && L.nseg <= 64u) {
        L.g = tm.group[s0 + lane];
        L.lo = tm.lit_off[s0 + lane];
        L.ll = tm.lit_len[s0 + lane];
    }
    if (L.nseg <= 64u) {
        const int gi = L.g < 0 ? 0 : 2 * L.g;
        const int32_t b_lo = __shfl(cap_a, gi & 63), b_hi = __shfl(cap_b, gi & 63);
        const int32_t e_lo = __shfl(cap_a, (gi + 1) & 63), e_hi = __shfl(cap_b, (gi + 1) & 63);
        if (L.g >= 0) {
            L.b = gi < 64 ? b_lo : b_hi;
            L.e = gi + 1 < 64 ? e_lo : e_hi;   // (gi is even: gi + 1 < 64 iff gi < 64)
        }
    }
    return L;
}

// One wave per line.  Pass 1: sizes[i] = bytes of line i's JSON text (0 for lines without a match).  One sweep
// over the line: byte p contributes esc_len(p) once per template group that covers it (nested extractors repeat
// their bytes).
template <typename OFF>
__global__ void __launch_bounds__(256) k_jsonl_sizes(JsonlTemplates tm, const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n,
                                                    const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots,
                                                    int passthrough, uint32_t* __restrict__ sizes) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + uni(threadIdx.x >> 6);
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    const bool pt = passthrough != 0;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const uint64_t line_off = static_cast<uint64_t>(off[i]);  // issued with round 1
        const LineSegs L = load_line_segs(tm, match_id, caps, slots, i, lane);
        if (L.k < 0) {
            if (lane == 0) sizes[i] = 0;
            continue;
        }
        const uint8_t* line = data + uni(line_off);
        uint32_t mine = 0;
        if (L.nseg <= 64u) {
            // quotes / "null", and the span of the line that any group touches
            mine = L.g < 0 ? 0u : (L.b < 0 ? 4u : 2u);
            int32_t lo = 0x7FFFFFFF, hi = 0;
            for (uint32_t s = 0; s < L.nseg; ++s) {
                const int32_t b = lane_of(L.b, s), e = lane_of(L.e, s);
                if (b >= 0) { lo = min(lo, b); hi = max(hi, e); }
            }
            for (int32_t c0 = lo; c0 < hi; c0 += 64) {
                const int32_t p = c0 + static_cast<int32_t>(lane);
                const uint32_t el = p < hi ? esc_len(line[p], pt) : 0u;
                uint32_t cover = 0;
                for (uint32_t s = 0; s < L.nseg; ++s) cover += (p >= lane_of(L.b, s) && p < lane_of(L.e, s)) ? 1u : 0u;  // b < 0: e < 0 too
                mine += el * cover;
            }
        } else {
            // (templates with more than 64 segments: the plain loop, one segment after the other)
            const int32_t* cp = caps + i * static_cast<uint64_t>(slots);
            const uint32_t s0 = uni(tm.seg_off[L.k]);
            for (uint32_t s = s0; s < s0 + L.nseg; ++s) {
                const int32_t g = uni(tm.group[s]);
                if (g < 0) continue;
                const int32_t b = uni(cp[2 * g]), e = uni(cp[2 * g + 1]);
                if (b < 0) { if (lane == 0) mine += 4u; continue; }
                if (lane == 0) mine += 2u;
                for (int32_t p = b + static_cast<int32_t>(lane); p < e; p += 64) mine += esc_len(line[p], pt);
            }
        }
        const uint32_t total = L.fixed + wave_sum(mine);
        if (lane == 0) sizes[i] = total;
    }
}

// Pass 2: the text goes to out + out_off[i].  The line's output is a flat sequence of items -- literal bytes,
// quotes, the letters of null, capture bytes -- in output order; the wave takes 64 items at a time: each lane
// finds its item's segment, loads its one source byte (all 64 loads are independent), and a running wave scan
// of the escaped lengths gives every item its place.
template <typename OFF>
__global__ void __launch_bounds__(256) k_jsonl_write(JsonlTemplates tm, const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n,
                                                    const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots,
                                                    int passthrough, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + uni(threadIdx.x >> 6);
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    const bool pt = passthrough != 0;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const uint64_t line_off = static_cast<uint64_t>(off[i]);  // issued with round 1
        const uint64_t o0v = out_off[i];
        const LineSegs L = load_line_segs(tm, match_id, caps, slots, i, lane);
        if (L.k < 0) continue;
        const uint8_t* line = data + uni(line_off);
        uint8_t* dst = out + uni(o0v);
        if (L.nseg <= 64u) {
            const uint32_t cnt = L.ll + (L.g < 0 ? 0u : (L.b < 0 ? 4u : static_cast<uint32_t>(L.e - L.b) + 2u));
            const uint32_t item_end = wave_inclusive(cnt, lane);
            const uint32_t item_start = item_end - cnt;
            const uint32_t items = uni(static_cast<uint32_t>(__shfl(static_cast<int>(item_end), 63)));
            uint32_t running = 0;
            for (uint32_t c0 = 0; c0 < items; c0 += 64u) {
                const uint32_t t = c0 + lane;
                const bool valid = t < items;
                uint32_t seg = 0;
                for (uint32_t s = 0; s + 1 < L.nseg; ++s) seg += t >= static_cast<uint32_t>(lane_of(static_cast<int32_t>(item_end), s)) ? 1u : 0u;
                const uint32_t u = t - static_cast<uint32_t>(__shfl(static_cast<int>(item_start), static_cast<int>(seg)));
                const uint32_t sll = static_cast<uint32_t>(__shfl(static_cast<int>(L.ll), static_cast<int>(seg)));
                const uint32_t slo = static_cast<uint32_t>(__shfl(static_cast<int>(L.lo), static_cast<int>(seg)));
                const int32_t sb = __shfl(L.b, static_cast<int>(seg)), se = __shfl(L.e, static_cast<int>(seg));
                const bool is_lit = u < sll;
                const uint32_t u2 = u - sll;  // position inside the capture's text: quote, bytes, quote -- or n,u,l,l
                const bool is_byte = !is_lit && sb >= 0 && u2 != 0u && u2 != static_cast<uint32_t>(se - sb) + 1u;
                const uint8_t* src = is_byte ? line + (sb + static_cast<int32_t>(u2) - 1) : tm.lits + (is_lit ? slo + u : 0u);
                uint32_t v = valid ? *src : 0u;
                if (!is_lit && !is_byte) v = sb < 0 ? static_cast<uint32_t>("null"[u2 & 3u]) : 0x22u;
                const uint32_t el = !valid ? 0u : (is_byte ? esc_len(v, pt) : 1u);
                const uint32_t inc = wave_inclusive(el, lane);
                uint8_t* at = dst + (running + inc - el);
                if (valid) {
                    if (is_byte) esc_write(at, v, pt);
                    else at[0] = static_cast<uint8_t>(v);
                }
                running += uni(static_cast<uint32_t>(__shfl(static_cast<int>(inc), 63)));
            }
            continue;
        }
        // (templates with more than 64 segments: the plain loop, one segment after the other)
        const int32_t* cp = caps + i * static_cast<uint64_t>(slots);
        const uint32_t s0 = uni(tm.seg_off[L.k]);
        for (uint32_t s = s0; s < s0 + L.nseg; ++s) {
            const uint8_t* lit = tm.lits + uni(tm.lit_off[s]);
            const uint32_t ll = uni(tm.lit_len[s]);
            for (uint32_t q = lane; q < ll; q += 64u) dst[q] = lit[q];
            dst += ll;
            const int32_t g = uni(tm.group[s]);
            if (g < 0) continue;
            const int32_t b = uni(cp[2 * g]), e = uni(cp[2 * g + 1]);
            if (b < 0) {
                if (lane < 4u) dst[lane] = "null"[lane];
                dst += 4;
                continue;
            }
            if (lane == 0) dst[0] = '"';
            ++dst;
            for (int32_t c0 = b; c0 < e; c0 += 64) {
                const int32_t p = c0 + static_cast<int32_t>(lane);
                const bool in = p < e;
                const uint32_t v = in ? line[p] : 0u;
                const uint32_t el = in ? esc_len(v, pt) : 0u;
                const uint32_t inc = wave_inclusive(el, lane);
                if (in) esc_write(dst + (inc - el), v, pt);
                dst += uni(static_cast<uint32_t>(__shfl(static_cast<int>(inc), 63)));
            }
            if (lane == 0) dst[0] = '"';
            ++dst;
        }
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

// ---- compact result rows for transport (the gather payload) ----
// row = [match id as int16][2 * Gmax offsets as uint16, 0xFFFF = unset]: 2 + 4 * Gmax bytes per line instead of
// 4 + 8 * Gmax.  Offsets above 65534 do not fit: such lines are counted and the caller sends the batch wide.
namespace {
__global__ void __launch_bounds__(256) k_pack_results(const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, uint64_t n, int slots,
                                                     uint16_t* __restrict__ packed, unsigned long long* __restrict__ n_overflow) {
    const uint64_t width = static_cast<uint64_t>(slots) + 1;
    const uint64_t total = n * width;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    uint32_t over = 0;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint64_t line = t / width;
        const uint32_t col = static_cast<uint32_t>(t - line * width);
        int32_t v;
        if (col == 0) v = match_id[line];
        else {
            v = caps[line * static_cast<uint64_t>(slots) + (col - 1)];
            if (v > 65534) { over = 1; v = 65534; }
        }
        packed[t] = static_cast<uint16_t>(v);  // -1 -> 0xFFFF; match ids are >= -32768 (at most 32767 extractions)
    }
    if (over) atomicAdd(n_overflow, 1ull);
}
__global__ void __launch_bounds__(256) k_unpack_results(const uint16_t* __restrict__ packed, uint64_t n, int slots, int32_t* __restrict__ match_id,
                                                       int32_t* __restrict__ caps) {
    const uint64_t width = static_cast<uint64_t>(slots) + 1;
    const uint64_t total = n * width;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint64_t line = t / width;
        const uint32_t col = static_cast<uint32_t>(t - line * width);
        const uint16_t v = packed[t];
        if (col == 0) match_id[line] = static_cast<int16_t>(v);
        else caps[line * static_cast<uint64_t>(slots) + (col - 1)] = v == 0xFFFFu ? -1 : static_cast<int32_t>(v);
    }
}
}  // namespace

hipError_t launch_pack_results(const int32_t* match_id, const int32_t* caps, uint64_t n, int slots, uint16_t* packed,
                               unsigned long long* d_overflow, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_overflow, 0, 8, stream);
    if (e != hipSuccess || n == 0) return e;
    uint64_t blocks = (n * (static_cast<uint64_t>(slots) + 1) + 255) / 256;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    hipLaunchKernelGGL(k_pack_results, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, match_id, caps, n, slots, packed, d_overflow);
    return hipGetLastError();
}
hipError_t launch_unpack_results(const uint16_t* packed, uint64_t n, int slots, int32_t* match_id, int32_t* caps, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n * (static_cast<uint64_t>(slots) + 1) + 255) / 256;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    hipLaunchKernelGGL(k_unpack_results, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, packed, n, slots, match_id, caps);
    return hipGetLastError();
}

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
        hipLaunchKernelGGL(k_jsonl_sizes<uint64_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint64_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, sizes);
    else
        hipLaunchKernelGGL(k_jsonl_sizes<uint32_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint32_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, sizes);
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
        hipLaunchKernelGGL(k_jsonl_write<uint64_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint64_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, line_out_off, out);
    else
        hipLaunchKernelGGL(k_jsonl_write<uint32_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, t, static_cast<const uint8_t*>(b.data),
                           static_cast<const uint32_t*>(b.offsets), b.n, b.match_id, b.caps, slots, passthrough, line_out_off, out);
    return hipGetLastError();
}

}  // namespace gx
