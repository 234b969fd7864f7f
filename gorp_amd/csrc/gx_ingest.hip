// gx_ingest.hip -- line ingestion on the device: raw log bytes -> CSR offsets.
//
// The reference has no equivalent: its callers hand in java.lang.Strings "often coming from a line-oriented
// input source" (README.md:26), i.e. BufferedReader.readLine(): a line ends at '\n', at '\r', or at "\r\n"
// (one terminator), and a final line without terminator still counts.  This file produces, for a byte buffer
// in HBM, offsets[0..n] such that line i = bytes[offsets[i], offsets[i+1]) *including* its terminator; the
// extract kernels strip the terminator when gx_batch_opts.strip_eol is set (gx_kernels.hip: trim_eol), so
// match offsets stay relative to the start of the line exactly as for a Java String.
//
// Three bandwidth-bound passes: count line ends per block -> exclusive scan of the block counts -> write offsets (and, optionally,
// a per-line flag "contains a byte >= 0x80": such a line is only Latin-1 if the file is; UTF-8 input needs the UTF-16 route).  Since
// the end of round 4 the text is read ONCE: the counting pass leaves its masks of line ends (16 bits per 16-byte chunk: an eighth of
// a byte per byte of text, stored 16 bytes per thread through LDS; with line flags the masks of the bytes >= 0x80 as well) and the
// writing pass reads those -- 0.57 ms of kernels per 2 GB against 0.78 when pass 3 swept the text again (count 0.44 instead of 0.32,
// write 0.08 instead of 0.41; gx_split_lines 0.55 against 0.79 ms, its workspace kept between calls).
// Two single-sweep variants were built and measured in round 2 (10 M x 201-byte lines; this version: 0.96 ms):
// one pass with decoupled look-back between workgroups (tickets, one 64-bit [tag | count] word per block, agent-scope
// relaxed atomics, wave-wide look-back) took 1.30 ms with 32 KiB blocks and 3.7 ms with 128 KiB blocks -- the polling
// of the predecessors' words goes to the memory side of the XCDs' L2s and competes with the text stream; and pass 1
// spilling its 16-bit chunk masks for pass 3 (text read once, 2-byte stores) took 1.02 ms.  Neither was kept (the second came back in
// round 4 with 16-byte stores: above).
// A third one in round 4: one workgroup of 1 024 threads per CU, a MiB per step, the line-end masks of its 64 chunks per thread kept in 32
// registers, ONE barrier of the grid per step (eight for 2 GB), offsets written out of the masks -- the text read once, no look-back:
// 0.84 ms against 0.79 for the two sweeps: while a step's offsets are written (1 024 LDS-ranked chunks per thread-row) and the grid
// waits at its barrier nothing is loaded, and the masks leave no registers to load the next step's text under them (with the escape
// bits the kernel spilled: 7.5 ms for the pipeline).  Not kept.
// Round 3 (per kernel before: count 328 us = 6.1 TB/s, scan 95 us, write 546 us): the scan takes 4096 counts per step and one barrier
// (52 us); the write pass classifies a thread's eight chunks first -- sixteen loads in flight, as in the counting pass -- and needs
// one barrier per block instead of sixteen; the wave scans run on DPP, not through LDS (491 us).  Measured and not kept: the byte
// behind a chunk taken from the neighbouring lane / wave (shuffle, LDS) instead of loaded: count 404 us, write 565 us.
#include <cstdint>
#include <hip/hip_runtime.h>

#include "gx_device.hpp"

namespace gx {
namespace {

constexpr int SPLIT_THREADS = 256;
constexpr int SPLIT_ITERS = 8;                                   // 16-byte chunks per thread
constexpr uint64_t SPLIT_BLOCK_BYTES = static_cast<uint64_t>(SPLIT_THREADS) * SPLIT_ITERS * 16;  // 32 KiB

// 0x80 in every byte of v that is zero (exact, no cross-byte carries)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {
    const uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | v | 0x7F7F7F7Fu);
}
// bit j of the result = bit 7 of byte j of the four dwords (16 bytes -> 16 bits)
// (v_dot4_u32_u8: four bytes times four weights, summed -- a byte of 0x80 times 2^j is 128 * 2^j; two words per sum, so that
// the weights fit a byte: ten instructions a mask instead of twenty-six, in kernels that count their instructions: the write
// pass with the escape bits was bound by them, 0.55 ms for 2 GB)
__device__ __forceinline__ uint32_t pack_hi_bits(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t lo = __builtin_amdgcn_udot4(b & 0x80808080u, 0x80402010u, __builtin_amdgcn_udot4(a & 0x80808080u, 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(d & 0x80808080u, 0x80402010u, __builtin_amdgcn_udot4(c & 0x80808080u, 0x08040201u, 0u, false), false);
    return (lo >> 7) | ((hi >> 7) << 8);
}

struct Chunk {
    uint32_t ends;    // bit j: byte j ends a line
    uint32_t high;    // bit j: byte j >= 0x80
    uint32_t esc;     // ESC: bit j: byte j takes one more byte inside a JSON string ('"', '\\', tab; a byte >= 0x80 unless it passes as it is)
    uint32_t hard;    // ESC: some byte is a control character other than tab, LF and CR (six bytes in JSON: the sizes pass reads the text then)
};

// Classify the 16 bytes at `pos` (pos is a multiple of 16; bytes at and beyond `size` do not exist).
// next_byte = the byte at pos + 16 (or 0 when there is none): decides whether a '\r' in the last slot stands alone.
template <bool ESC = false>
__device__ __forceinline__ Chunk classify(const uint8_t* __restrict__ data, uint64_t pos, uint64_t size, bool passthrough = false) {
    Chunk c{0, 0, 0, 0};
    if (pos >= size) return c;
    uint32_t w[4];
    uint32_t next_byte = 0;
    if (pos + 16 <= size) {
        const uint4 v = *reinterpret_cast<const uint4*>(data + pos);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (pos + 16 < size) next_byte = data[pos + 16];
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (uint64_t q = pos; q < size; ++q) w[(q - pos) >> 2] |= static_cast<uint32_t>(data[q]) << (((q - pos) & 3) * 8);
    }
    uint32_t valid = 0xFFFFu;
    if (pos + 16 > size) valid = (1u << (size - pos)) - 1u;  // the zero padding above is neither LF nor CR
    uint32_t lf, cr;
    if (ESC) {
        // one sweep over the four words for everything (on the low seven bits of every byte, where v + 0x7F.. sets bit 7 exactly when
        // v != 0 and nothing carries); the >= 0x80 mask is not made here (the JSON Lines pipeline asks for no line flags)
        uint32_t e[4], zl[4], zc[4], hard = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = w[q], v7 = v & 0x7F7F7F7Fu;
            const uint32_t nq = (v7 ^ 0x22222222u) + 0x7F7F7F7Fu, nb = (v7 ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu, nt = (v7 ^ 0x09090909u) + 0x7F7F7F7Fu;
            const uint32_t nl = (v7 ^ 0x0A0A0A0Au) + 0x7F7F7F7Fu, nr = (v7 ^ 0x0D0D0D0Du) + 0x7F7F7F7Fu;
            zl[q] = ~(nl | v);                                               // bit 7 of a byte: LF
            zc[q] = ~(nr | v);                                               // ... CR
            e[q] = (~(nq & nb & nt) & ~v) | (passthrough ? 0u : v);          // ... quote, backslash or tab -- or >= 0x80
            // ... below 0x20 and not tab, LF or CR (LF and CR end lines wherever they stand: never inside a line's text)
            hard |= ~((v7 + 0x60606060u) | v) & nt & nl & nr;
        }
        lf = pack_hi_bits(zl[0], zl[1], zl[2], zl[3]);
        cr = pack_hi_bits(zc[0], zc[1], zc[2], zc[3]);
        c.esc = pack_hi_bits(e[0], e[1], e[2], e[3]) & valid;
        // (the zero padding behind the text's last byte reads as control characters: only whole chunks are asked)
        c.hard = pos + 16 <= size ? (hard & 0x80808080u) : 0u;
        if (pos + 16 > size)
            for (uint64_t q = pos; q < size; ++q) { const uint32_t bt = data[q]; if (bt < 0x20u && bt != 0x09u && bt != 0x0Au && bt != 0x0Du) c.hard = 1u; }
    } else {
        lf = pack_hi_bits(zero_bytes(w[0] ^ 0x0A0A0A0Au), zero_bytes(w[1] ^ 0x0A0A0A0Au), zero_bytes(w[2] ^ 0x0A0A0A0Au), zero_bytes(w[3] ^ 0x0A0A0A0Au));
        cr = pack_hi_bits(zero_bytes(w[0] ^ 0x0D0D0D0Du), zero_bytes(w[1] ^ 0x0D0D0D0Du), zero_bytes(w[2] ^ 0x0D0D0D0Du), zero_bytes(w[3] ^ 0x0D0D0D0Du));
        c.high = pack_hi_bits(w[0], w[1], w[2], w[3]) & valid;
    }
    // a CR ends a line unless the next byte is LF
    const uint32_t lf_after = (lf >> 1) | (next_byte == 0x0Au ? 0x8000u : 0u);
    c.ends = (lf | (cr & ~lf_after)) & valid;
    return c;
}

// inclusive sum over the wave's lanes, on the vector unit alone (DPP: shifts within rows of 16 lanes, then the last lane of a row
// broadcast to the rows after it); `lane` is unused
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, uint32_t) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xF, 0xF, true));   // row_shr:8
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return v;
}

// pass 1: line ends per block AND the blocks' masks of line ends, 16 bits per 16-byte chunk
// in chunk order (end_masks[pos / 16]), so that pass 3 reads an eighth of a byte per byte of text instead of the text.  The masks go
// through LDS and leave as 16-byte stores (a thread's eight masks sit 512 bytes apart; two-byte stores were what made this layout
// lose in round 2: 1.02 against 0.96 ms).  ESC: the escape bits of the text the same way (Chunk::esc: gx_jsonl.hip's sizes pass).
// HIGH (not with ESC): the masks of the bytes >= 0x80 as well (high_masks: the line flags of pass 3).
template <bool ESC, bool HIGH>
__global__ void __launch_bounds__(SPLIT_THREADS) k_split_count_masks(const uint8_t* __restrict__ data, uint64_t size, uint32_t* __restrict__ block_counts,
                                                                      uint16_t* __restrict__ end_masks, uint16_t* __restrict__ esc_bits,
                                                                      uint64_t* __restrict__ hard_any, int passthrough, uint16_t* __restrict__ high_masks) {
    constexpr int CHUNKS = SPLIT_THREADS * SPLIT_ITERS;   // 2 048 per block
    __shared__ uint32_t wsum[SPLIT_THREADS / 64];
    __shared__ __attribute__((aligned(16))) uint16_t tr[((ESC || HIGH) ? 2 : 1) * CHUNKS];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SPLIT_BLOCK_BYTES;
    uint32_t cnt = 0, hard = 0;
#pragma unroll
    for (int it = 0; it < SPLIT_ITERS; ++it) {
        const uint64_t pos = base + (static_cast<uint64_t>(it) * SPLIT_THREADS + threadIdx.x) * 16;
        const Chunk c = classify<ESC>(data, pos, size, passthrough != 0);
        cnt += __popc(c.ends);
        tr[it * SPLIT_THREADS + threadIdx.x] = static_cast<uint16_t>(c.ends);
        if (ESC) { tr[CHUNKS + it * SPLIT_THREADS + threadIdx.x] = static_cast<uint16_t>(c.esc); hard |= c.hard; }
        if (HIGH) tr[CHUNKS + it * SPLIT_THREADS + threadIdx.x] = static_cast<uint16_t>(c.high);
    }
    if (ESC && hard) *hard_any = 1;   // (plain stores of the same value: no atomic)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t inc = wave_inclusive_sum(cnt, lane);
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < SPLIT_THREADS / 64; ++w) t += wsum[w];
        block_counts[blockIdx.x] = t;
    }
    // eight masks = 16 bytes per thread, in chunk order (the arrays are 16-byte aligned and padded to whole blocks)
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * CHUNKS + 8u * threadIdx.x;
    *reinterpret_cast<uint4*>(end_masks + first) = *reinterpret_cast<const uint4*>(tr + 8u * threadIdx.x);
    if (ESC) *reinterpret_cast<uint4*>(esc_bits + first) = *reinterpret_cast<const uint4*>(tr + CHUNKS + 8u * threadIdx.x);
    if (HIGH) *reinterpret_cast<uint4*>(high_masks + first) = *reinterpret_cast<const uint4*>(tr + CHUNKS + 8u * threadIdx.x);
}

// pass 3: offsets[1 + rank(p)] = p + 1 for every line end p, out of the masks: a thread takes eight chunks in a row
// flags (optional, with high_masks): flags[line] = 1 for lines with a byte >= 0x80
template <typename OFF>
__global__ void __launch_bounds__(SPLIT_THREADS) k_split_write_masks(const uint16_t* __restrict__ end_masks, const uint64_t* __restrict__ prefix,
                                                                     OFF* __restrict__ offsets, uint64_t cap_lines, const uint16_t* __restrict__ high_masks,
                                                                     uint8_t* __restrict__ flags) {
    constexpr int CHUNKS = SPLIT_THREADS * SPLIT_ITERS;
    __shared__ uint32_t wsum[SPLIT_THREADS / 64];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SPLIT_BLOCK_BYTES;
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[0] = 0;
    const uint4 v = *reinterpret_cast<const uint4*>(end_masks + static_cast<uint64_t>(blockIdx.x) * CHUNKS + 8u * threadIdx.x);
    const uint32_t mw[4] = {v.x, v.y, v.z, v.w};
    const uint32_t cnt = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_sum(cnt, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (uint32_t q = 0; q < SPLIT_THREADS / 64; ++q) if (q < wave) wbase += wsum[q];
    uint64_t k = prefix[blockIdx.x] + wbase + inc - cnt;   // line ends before this thread's first chunk
    uint4 hv = make_uint4(0u, 0u, 0u, 0u);
    if (flags) hv = *reinterpret_cast<const uint4*>(high_masks + static_cast<uint64_t>(blockIdx.x) * CHUNKS + 8u * threadIdx.x);
    const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t e = (q & 1) ? (mw[q >> 1] >> 16) : (mw[q >> 1] & 0xFFFFu);
        const uint64_t pos = base + (8ull * threadIdx.x + q) * 16;
        uint32_t hb = (q & 1) ? (hw[q >> 1] >> 16) : (hw[q >> 1] & 0xFFFFu);
        while (hb) {   // (k: the line this chunk's first byte belongs to)
            const uint32_t j = __ffs(hb) - 1u;
            hb &= hb - 1u;
            const uint64_t line = k + __popc(e & ((1u << j) - 1u));
            if (line < cap_lines) flags[line] = 1;
        }
        while (e) {
            const uint32_t j = __ffs(e) - 1u;
            e &= e - 1u;
            ++k;  // the line that starts after this end
            if (k <= cap_lines) offsets[k] = static_cast<OFF>(pos + j + 1);
        }
    }
}

// pass 2: exclusive scan of the block counts (one workgroup), total -> *total_ends; 4096 counts per step of the loop
__global__ void __launch_bounds__(1024) k_split_scan(const uint32_t* __restrict__ counts, uint64_t* __restrict__ prefix, uint64_t nblocks,
                                                    uint64_t* __restrict__ total_ends) {
    __shared__ uint64_t wsum[2][16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t carry = 0;   // (every thread keeps it)
    uint32_t flip = 0;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 4096, flip ^= 1u) {
        const uint64_t b = b0 + 4ull * threadIdx.x;
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = b + q < nblocks ? counts[b + q] : 0u;
        const uint64_t mine = static_cast<uint64_t>(v[0]) + v[1] + v[2] + v[3];
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(static_cast<unsigned long long>(inc), d);
            if (lane >= static_cast<uint32_t>(d)) inc += o;
        }
        if (lane == 63) wsum[flip][wave] = inc;
        __syncthreads();   // (one barrier per step: the sums alternate between two rows)
        uint64_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint64_t t = wsum[flip][w];
            if (w < wave) wbase += t;
            total += t;
        }
        uint64_t run = carry + wbase + inc - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (b + q < nblocks) prefix[b + q] = run;
            run += v[q];
        }
        carry += total;
    }
    if (threadIdx.x == 0) *total_ends = carry;
}

// the last line has no terminator: close it with offsets[n] = size
template <typename OFF>
__global__ void k_split_finish(const uint8_t* __restrict__ data, uint64_t size, const uint64_t* __restrict__ total_ends, OFF* __restrict__ offsets,
                               uint64_t cap_lines, uint64_t* __restrict__ n_lines) {
    const uint64_t ends = *total_ends;
    uint64_t n = ends;
    if (size > 0) {
        const uint8_t last = data[size - 1];
        if (!(last == 0x0Au || last == 0x0Du)) {  // a trailing CR always ends a line (nothing follows it)
            n = ends + 1;
            if (n <= cap_lines) offsets[n] = static_cast<OFF>(size);
        }
    }
    *n_lines = n;
}

// the longest line, terminator included (what gx_batch_opts.max_line_bytes wants to hear): one sweep over the offsets just written
// (4 bytes per line: 1 % of the text's bytes)
template <typename OFF>
__global__ void __launch_bounds__(256) k_split_max(const OFF* __restrict__ offsets, const uint64_t* __restrict__ n_lines, uint64_t cap_lines,
                                                   unsigned long long* __restrict__ max_line) {
    const uint64_t n = min(*n_lines, cap_lines);
    unsigned long long m = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        m = max(m, static_cast<unsigned long long>(offsets[i + 1]) - static_cast<unsigned long long>(offsets[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
    // (one atomic per workgroup, and none when the maximum would not move: atomics on one cache line take their turns at 11 ns
    // apiece -- one per wave of 2 048 workgroups made this sweep of 40 MB 105 us)
    __shared__ unsigned long long wmax[4];
    if ((threadIdx.x & 63u) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long b = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (b > __hip_atomic_load(max_line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_line, b);
    }
}

}  // namespace

size_t split_workspace_bytes(uint64_t size, bool with_flags) {
    const uint64_t nblocks = (size + SPLIT_BLOCK_BYTES - 1) / SPLIT_BLOCK_BYTES;
    // (+ the masks of line ends of the text-read-once split: 4 KiB per block = an eighth of the text, 16-byte aligned)
    return static_cast<size_t>((nblocks + 1) * 4 + (nblocks + 3) * 8 + 64 + (with_flags ? 2 : 1) * nblocks * 4096 + 32);
}

// workspace: [total_ends u64][n_lines u64][max_line u64][hard_any u64][prefix u64 * nblocks][counts u32 * nblocks]
hipError_t launch_split_lines(const uint8_t* data, uint64_t size, void* offsets, int offsets64, uint64_t cap_lines, uint8_t* flags,
                              void* workspace, uint64_t** d_n_lines, hipStream_t stream, uint64_t** d_max_line, uint16_t* esc_bits, int passthrough) {
    const uint64_t nblocks = (size + SPLIT_BLOCK_BYTES - 1) / SPLIT_BLOCK_BYTES;
    uint64_t* total_ends = static_cast<uint64_t*>(workspace);
    uint64_t* n_lines = total_ends + 1;
    uint64_t* max_line = total_ends + 2;
    uint64_t* hard_any = total_ends + 3;
    uint64_t* prefix = total_ends + 4;
    uint32_t* counts = reinterpret_cast<uint32_t*>(prefix + nblocks + 1);
    *d_n_lines = n_lines;
    if (d_max_line) *d_max_line = max_line;
    hipError_t e;
    if (flags && cap_lines) {
        e = hipMemsetAsync(flags, 0, cap_lines, stream);
        if (e != hipSuccess) return e;
    }
    if (esc_bits) {
        e = hipMemsetAsync(hard_any, 0, 8, stream);
        if (e != hipSuccess) return e;
    }
    if (nblocks == 0) {
        e = hipMemsetAsync(workspace, 0, 32, stream);  // no ends, no lines
        if (e != hipSuccess) return e;
        return hipMemsetAsync(offsets, 0, offsets64 ? 8 : 4, stream);
    }
    if (nblocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (esc_bits && (offsets64 || flags)) return hipErrorInvalidValue;   // (the escape bits come with 32-bit offsets and without line flags)
    {
        // the text is read once: pass 1 leaves the masks of line ends (and the escape bits, or the masks of the bytes >= 0x80), pass 3 reads those
        uint16_t* end_masks = reinterpret_cast<uint16_t*>((reinterpret_cast<uintptr_t>(counts + nblocks + 1) + 15u) & ~static_cast<uintptr_t>(15u));
        uint16_t* high_masks = flags ? end_masks + nblocks * 2048 : nullptr;
        if (esc_bits) hipLaunchKernelGGL((k_split_count_masks<true, false>), dim3(static_cast<unsigned>(nblocks)), dim3(SPLIT_THREADS), 0, stream, data, size, counts, end_masks,
                                         esc_bits, hard_any, passthrough, nullptr);
        else if (flags) hipLaunchKernelGGL((k_split_count_masks<false, true>), dim3(static_cast<unsigned>(nblocks)), dim3(SPLIT_THREADS), 0, stream, data, size, counts, end_masks,
                                           nullptr, nullptr, 0, high_masks);
        else hipLaunchKernelGGL((k_split_count_masks<false, false>), dim3(static_cast<unsigned>(nblocks)), dim3(SPLIT_THREADS), 0, stream, data, size, counts, end_masks,
                                nullptr, nullptr, 0, nullptr);
        hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, stream, counts, prefix, nblocks, total_ends);
        if (offsets64) {
            hipLaunchKernelGGL(k_split_write_masks<uint64_t>, dim3(static_cast<unsigned>(nblocks)), dim3(SPLIT_THREADS), 0, stream, end_masks, prefix,
                               static_cast<uint64_t*>(offsets), cap_lines, high_masks, flags);
            hipLaunchKernelGGL(k_split_finish<uint64_t>, dim3(1), dim3(1), 0, stream, data, size, total_ends, static_cast<uint64_t*>(offsets), cap_lines, n_lines);
        } else {
            hipLaunchKernelGGL(k_split_write_masks<uint32_t>, dim3(static_cast<unsigned>(nblocks)), dim3(SPLIT_THREADS), 0, stream, end_masks, prefix,
                               static_cast<uint32_t*>(offsets), cap_lines, high_masks, flags);
            hipLaunchKernelGGL(k_split_finish<uint32_t>, dim3(1), dim3(1), 0, stream, data, size, total_ends, static_cast<uint32_t*>(offsets), cap_lines, n_lines);
        }
    }
    if (d_max_line) {
        e = hipMemsetAsync(max_line, 0, 8, stream);
        if (e != hipSuccess) return e;
        const uint64_t want = (size / 64 + 255) / 256 + 1;   // (a thread per line of 64 bytes: more lines than that, and they take turns)
        const dim3 grid(static_cast<unsigned>(want < 1024 ? want : 1024));
        if (offsets64) hipLaunchKernelGGL(k_split_max<uint64_t>, grid, dim3(256), 0, stream, static_cast<const uint64_t*>(offsets), n_lines, cap_lines,
                                          reinterpret_cast<unsigned long long*>(max_line));
        else hipLaunchKernelGGL(k_split_max<uint32_t>, grid, dim3(256), 0, stream, static_cast<const uint32_t*>(offsets), n_lines, cap_lines,
                                reinterpret_cast<unsigned long long*>(max_line));
    }
    return hipGetLastError();
}

}  // namespace gx
