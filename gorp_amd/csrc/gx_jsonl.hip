// gx_jsonl.hip -- result materialisation on the device: match ids + capture
// offsets -> one JSON object per matched line, i.e. ExtractionResult.asMap(idAs)
// (core/ExtractionResult.java:65-88) serialised the way a Jackson ObjectMapper writes a LinkedHashMap:
//   {"<idAs>":"<extraction name>","<extractor 1>":"<captured text>",...,<append entries>}\n
// A capture that Matcher.group() would report as null is written as null.  The key order, the
// "later put replaces the value but keeps the position" rule and the append entries are resolved on the host
// into one template per extraction (gx_api.cpp: build_jsonl_templates): a list of segments, each a literal byte
// string followed by an optional capture group.
//
// Two passes of one kernel (k_jsonl_tile, below: a wave per 64 consecutive lines staged in LDS, a lane per line): the sizes pass
// sums the escaped lengths -- and leaves, per line, the point where the write pass's two waves divide its text, and per tile whether
// anything in it needs an escape at all --, an exclusive scan turns the sizes into output offsets, the write pass assembles each
// tile's text in LDS and flushes it.  A line that does not fit the staging areas is taken by a whole wave, a lane per byte
// (line_size_wave / line_write_wave).  Bytes are Latin-1 code units (the batch path's input model) and leave as UTF-8; with
// utf8_passthrough the bytes >= 0x80 are copied as they are (input that was UTF-8 all along).  DESIGN.md section 9.
#include <algorithm>
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

// One line, the whole wave (the fallback of the tile kernels for a line that does not fit their LDS staging, and
// the path for templates with more than 64 segments).  Returns the bytes of line i's JSON text (0: no match); one
// sweep over the line: byte p contributes esc_len(p) once per template group that covers it (nested extractors
// repeat their bytes).
__device__ uint32_t line_size_wave(const JsonlTemplates& tm, const uint8_t* __restrict__ data, uint64_t line_off, uint64_t i,
                                   const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots, bool pt, uint32_t lane) {
    const LineSegs L = load_line_segs(tm, match_id, caps, slots, i, lane);
    if (L.k < 0) return 0u;
    const uint8_t* line = data + line_off;
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
    return L.fixed + wave_sum(mine);
}

// Pass 2 for one line, the whole wave: the text goes to dst.  The line's output is a flat sequence of items --
// literal bytes, quotes, the letters of null, capture bytes -- in output order; the wave takes 64 items at a
// time: each lane finds its item's segment, loads its one source byte (all 64 loads are independent), and a
// running wave scan of the escaped lengths gives every item its place.
__device__ void line_write_wave(const JsonlTemplates& tm, const uint8_t* __restrict__ data, uint64_t line_off, uint64_t i,
                                const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots, bool pt, uint32_t lane,
                                uint8_t* dst) {
    const LineSegs L = load_line_segs(tm, match_id, caps, slots, i, lane);
    if (L.k < 0) return;
    const uint8_t* line = data + line_off;
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
        return;
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

// ---------------------------------------------------------------------------
// Tile kernels: one wave per 64 consecutive lines, one lane per line
// ---------------------------------------------------------------------------
// With one line per wave nearly all of the ~300-500 vector instructions per line are wave-uniform bookkeeping.
// Here, as in the extraction kernel, a wave stages the contiguous bytes of 64 lines in LDS with coalesced
// 16-byte loads and every lane then works through its own line (template, capture offsets, escapes).  The write
// pass assembles the tile's output -- also one contiguous span -- in LDS and flushes it with 16-byte stores.
// A group of lines that does not fit the staging areas is taken in several rounds of consecutive lanes; a
// single line that does not fit goes through line_size_wave / line_write_wave.
extern __shared__ __attribute__((aligned(16))) uint8_t jx_smem[];

#ifdef GX_DEV
// developer build: cycles per phase of the tile kernels, summed over the waves (tools/jsonl_phases.py)
__device__ unsigned long long jx_phase[16];
#define JX_STAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[slot] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define JX_STAMP(slot) do { } while (0)
#endif

// where the write pass's two waves divide a line's text: at this many 256ths of its estimated size (0.44: the first wave of a pair has
// the many short segments; measured best of 0.38 .. 0.63)
constexpr uint32_t SPLIT_AT_256 = 112;

struct JsonlTileCfg {
    uint32_t lits_lds;    // LDS offset of a copy of tm.lits, or 0xFFFFFFFF: read them from global memory
    uint32_t lits_bytes;
    uint32_t waves;       // per workgroup
    uint32_t in_bytes;    // per-wave staging of line bytes (multiple of 16)
    uint32_t out_bytes;   // per-wave staging of output text (multiple of 16; 0 in the sizes pass)
    uint32_t stage0;      // LDS offset of wave 0's areas
    uint32_t tm_lds;      // LDS offset of a copy of the template arrays (seg_off | fixed_len | lit_off | lit_len | group),
                          // or 0xFFFFFFFF: read them from global memory
    uint32_t n_rules, n_segs;
    uint32_t caps_bytes;  // per-wave staging of the tile's capture rows (64 * slots * 4), 0: read them from global memory
    uint32_t perm_lds;    // LDS offset of the write pass's 16 byte-permute selectors (128 bytes)
};

// four bytes of LDS at any alignment (gfx950 reads LDS unaligned)
struct __attribute__((packed)) UnalignedU32 { uint32_t v; };

// The per-lane code addresses LDS by byte address (address-space-3 pointers made from integers: ds_* instructions, not
// flat ones) and only with ALIGNED 32-bit words: a 32-bit LDS access off its alignment is replayed at 64 cycles, reads
// and writes alike (measured here: SQ_LDS_UNALIGNED_STALL was 88 % of the LDS-active cycles of the write pass when
// it read the captures and wrote the text at byte addresses).  Unaligned text is read as aligned words joined by
// v_alignbyte and written through a three-byte carry (k_jsonl_tile).
#define JX_LDS __attribute__((address_space(3)))
__device__ __forceinline__ void lds_put_u32(uint32_t a, uint32_t v) { ((JX_LDS volatile UnalignedU32*)(uintptr_t)a)->v = v; }
__device__ __forceinline__ uint32_t lds_w(uint32_t a) { return *(JX_LDS const uint32_t*)(uintptr_t)a; }  // a 4-byte aligned word
__device__ __forceinline__ void lds_put_u8(uint32_t a, uint32_t v) { *(JX_LDS uint8_t*)(uintptr_t)a = static_cast<uint8_t>(v); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((JX_LDS const uint8_t*)p)); }

// the bytes of one character inside a JSON string, low byte first, and how many (1..6)
__device__ __forceinline__ uint64_t esc_bytes(uint32_t b, bool passthrough, uint32_t& count) {
    if (b >= 0x80u) {
        if (passthrough) { count = 1u; return b; }
        count = 2u;
        return (0xC0u | (b >> 6)) | (0x80u | (b & 0x3Fu)) << 8;
    }
    if (b >= 0x20u) {
        if (b == 0x22u || b == 0x5Cu) { count = 2u; return 0x5Cu | b << 8; }
        count = 1u;
        return b;
    }
    uint32_t letter = 0u;
    switch (b) {
    case 0x08u: letter = 'b'; break;
    case 0x09u: letter = 't'; break;
    case 0x0Au: letter = 'n'; break;
    case 0x0Cu: letter = 'f'; break;
    case 0x0Du: letter = 'r'; break;
    default: break;
    }
    if (letter) { count = 2u; return 0x5Cu | letter << 8; }
    count = 6u;
    const uint32_t lo = b & 15u;
    const uint64_t hex = static_cast<uint64_t>('0' + (b >> 4)) | static_cast<uint64_t>(lo < 10u ? '0' + lo : 'A' + (lo - 10u)) << 8;
    return 0x5Cu | static_cast<uint64_t>('u') << 8 | static_cast<uint64_t>('0') << 16 | static_cast<uint64_t>('0') << 24 | hex << 32;
}

// global -> LDS copy of the span [lo, hi) of `data`, 16 bytes per lane, skewed so that LDS and global addresses
// agree modulo 16; chunks that stick out of [data, data_end) are read byte by byte.  In two halves: stage_fetch issues all
// of a lane's loads (NQ of them: the staging area is at most 16 KB = 64 lanes x 16 chunks), stage_put stores them -- a
// load-store-load-store chain costs a trip to memory per chunk, and the tile kernels fetch the NEXT tile's bytes while the
// lanes work on this one.  Returns the number of chunks.
template <int NQ>
__device__ __forceinline__ uint32_t stage_fetch(const uint8_t* __restrict__ data, const uint8_t* data_end, uint64_t lo, uint64_t hi, uint32_t lane,
                                                uint32_t threads, uint4 (&v)[NQ]) {  // lane: 0 .. threads - 1; threads * NQ >= 1024
    const uint8_t* g_lo = data + lo;
    const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(g_lo) & 15u);
    const uint8_t* g_al = g_lo - skew;
    const uint32_t nch = static_cast<uint32_t>(((hi - lo) + skew + 15u) >> 4);
    if (g_al >= data && g_al + (static_cast<uint64_t>(nch) << 4) <= data_end) {
        // every tile but the batch's first and last: nothing sticks out, the loads are one straight run (no value of theirs
        // is merged with another path's, so nothing waits for one before the next is issued)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t c = threads * q + lane;
            v[q] = make_uint4(0u, 0u, 0u, 0u);
            if (c < nch) v[q] = *reinterpret_cast<const uint4*>(g_al + (static_cast<uint64_t>(c) << 4));
        }
        return nch;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t c = threads * q + lane;
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (c < nch) {
            const uint8_t* src = g_al + (static_cast<uint64_t>(c) << 4);
            if (src >= data && src + 16 <= data_end) x = *reinterpret_cast<const uint4*>(src);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int r = 0; r < 16; ++r)
                    if (src + r >= data && src + r < data_end) w[r >> 2] |= static_cast<uint32_t>(src[r]) << ((r & 3) * 8);
                x = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        v[q] = x;
    }
    return nch;
}
template <int NQ>
__device__ __forceinline__ void stage_put(uint8_t* stage, uint32_t nch, uint32_t lane, uint32_t threads, const uint4 (&v)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t c = threads * q + lane;
        if (c < nch) *reinterpret_cast<uint4*>(stage + (c << 4)) = v[q];
    }
}

// global -> LDS copy of `words` (<= 2048) int32, every lane's loads issued before its first store
__device__ __forceinline__ void stage_words(const int32_t* __restrict__ src, uint32_t words, int32_t* stage, uint32_t lane, uint32_t threads = 64u) {
    int32_t v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const uint32_t x = threads * q + lane;
        v[q] = x < words ? src[x] : 0;
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const uint32_t x = threads * q + lane;
        if (x < words) stage[x] = v[q];
    }
}

// PAIR (write pass) = 2: the workgroup is TWO waves that share one tile and its staging areas.  Lane l of both is on line l;
// the first writes the text before the line's split point, the second the rest.  The split point (a segment m and a
// character q of its capture, a multiple of 16) is chosen by both passes the same way from the template and the capture
// offsets, near the middle of the line's text; the sizes pass leaves the exact number of bytes before it in split[].
// With a wave per tile the write pass fits 4 waves per CU (13 + 21 KB of staging each), one per SIMD, and a lone wave
// is bound by its own issue rate; pairs make that 8 waves, each with half the text of a line to write.
// FAST: the template arrays, the literals and the tile's capture rows are all in LDS (plan_jsonl_tile: nearly every definition) --
// the lanes read them with ds_read.  (Without it each access is a flat load through a pointer that is LDS or global, which waits
// for every load in flight: the next tile's bytes among them.)
template <typename OFF, bool WRITE, int PAIR, bool FAST>
__global__ void __launch_bounds__(WRITE ? 128 : 768) k_jsonl_tile(JsonlTemplates tm, JsonlTileCfg cfg, const uint8_t* __restrict__ data, const OFF* __restrict__ off,
                                                   uint64_t n, const int32_t* __restrict__ match_id, const int32_t* __restrict__ caps, int slots,
                                                   int passthrough, uint32_t* __restrict__ sizes, const uint64_t* __restrict__ out_off,
                                                   uint8_t* __restrict__ out, uint32_t* __restrict__ split, uint32_t* __restrict__ split_at, uint32_t* __restrict__ tile_flags) {
    const bool pt = passthrough != 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uni(threadIdx.x >> 6);
    const uint32_t wave = wave_in_block / PAIR;   // the tile slot (staging areas) of this wave
    const uint32_t part = wave_in_block % PAIR;   // PAIR == 2: which side of the split point
    const uint32_t ptid = part * 64u + lane;      // lane among the waves that share the tile
    auto pair_barrier = [&]() {  // the waves that share a tile come together (PAIR == 2: they are the whole workgroup)
        if (PAIR == 2) __syncthreads();
        else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // LDS copies of the template arrays, when they are small enough (plan_jsonl_tile)
    // (seg_off[n_rules + 1] | fixed_len[n_rules] | up to the next 16 bytes | one uint4 per segment: lit_off, lit_len, group, 0)
    uint32_t* tl_seg_off = reinterpret_cast<uint32_t*>(jx_smem + (cfg.tm_lds == 0xFFFFFFFFu ? 0u : cfg.tm_lds));
    uint32_t* tl_fixed = tl_seg_off + (cfg.n_rules + 1u);
    const uint32_t tm_seg_rel = ((2u * cfg.n_rules + 1u) * 4u + 15u) & ~15u;
    uint4* tl_seg = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(tl_seg_off) + tm_seg_rel);
    const uint32_t smem0 = lds_addr(jx_smem);
    const uint32_t perm_tab = smem0 + cfg.perm_lds;
    if (WRITE && threadIdx.x < 16u) {
        // byte-permute selectors by escape mask (bit j: character j of the word takes a backslash): selector bytes 0..3 = the
        // characters, 4 = the backslash, 0x0C = a zero byte behind the text
        uint32_t sel[8], at = 0;
        for (uint32_t j = 0; j < 4u; ++j) {
            if ((threadIdx.x >> j) & 1u) sel[at++] = 4u;
            sel[at++] = j;
        }
        while (at < 8u) sel[at++] = 0x0Cu;
        lds_put_u32(perm_tab + threadIdx.x * 8u, sel[0] | sel[1] << 8 | sel[2] << 16 | sel[3] << 24);
        lds_put_u32(perm_tab + threadIdx.x * 8u + 4u, sel[4] | sel[5] << 8 | sel[6] << 16 | sel[7] << 24);
    }
    if (cfg.lits_lds != 0xFFFFFFFFu)
        for (uint32_t q = threadIdx.x; q < cfg.lits_bytes; q += blockDim.x) jx_smem[cfg.lits_lds + q] = tm.lits[q];
    if (cfg.tm_lds != 0xFFFFFFFFu) {
        for (uint32_t q = threadIdx.x; q <= cfg.n_rules; q += blockDim.x) tl_seg_off[q] = tm.seg_off[q];
        for (uint32_t q = threadIdx.x; q < cfg.n_rules; q += blockDim.x) tl_fixed[q] = tm.fixed_len[q];
        for (uint32_t q = threadIdx.x; q < cfg.n_segs; q += blockDim.x) tl_seg[q] = make_uint4(tm.lit_off[q], tm.lit_len[q], static_cast<uint32_t>(tm.group[q]), 0u);
    }
    __syncthreads();
#ifdef GX_DEV
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t per_wave = cfg.in_bytes + cfg.out_bytes + cfg.caps_bytes;
    uint8_t* in_stage = jx_smem + cfg.stage0 + wave * per_wave;
    uint8_t* out_stage = in_stage + cfg.in_bytes;
    uint8_t* caps_stage = out_stage + cfg.out_bytes;
    const uint8_t* data_end = data + static_cast<uint64_t>(off[n]);
    const uint64_t tiles = (n + 63) >> 6;
    const uint64_t wstride = static_cast<uint64_t>(gridDim.x) * (cfg.waves / PAIR);
    // ---- the pipeline: while the lanes work on a tile, the NEXT tile's line bytes and capture rows are already on their way
    // into registers (pf, pfc), its per-lane offsets too (nx_*), and the bounds of the tile after that (bd_*: what the
    // fetch needs to know its span).  Only for a tile that goes in one round (the common case: the staging areas are sized
    // for 64 mean lines and more); any other tile stages inside its rounds as before.  The write pass only: it runs two waves per
    // SIMD; the sizes pass has three and registers for neither the 64 bytes x 16 of a tile nor a fourth wave's worth of latency to hide
    // (measured: 1.17 ms with the fetch, 1.08 without) -- it fetches the per-lane offsets ahead and nothing else. ----
    constexpr int NQ = 16 / PAIR;   // 16-byte chunks of line bytes per lane (in_bytes <= 16 KB)
    constexpr int NC = 4;           // 16-byte chunks of capture rows per lane
    const uint32_t threads = 64u * PAIR;
    uint4 pf[NQ], pfc[NC];
    bool pf_valid = false;          // uniform
    uint32_t pf_nch = 0u, pf_cch = 0u;
    // (what a load fetches ahead keeps the type it is loaded with: widening a 32-bit offset to 64 bits is an instruction on the
    // loaded value, placed right behind the load -- a wait for the load that was meant to fly during the whole tile)
    OFF nx_o0 = 0, nx_o1 = 0, bd_lo = 0, bd_hi = 0;
    uint64_t nx_oo0 = 0, nx_oo1 = 0, bd_olo = 0, bd_ohi = 0;
    int32_t nx_k = -1;
    uint32_t nx_split = 0u;         // write pass, second wave of a pair: the bytes of the line's text before its split point
    uint32_t nx_at = 0xFFFF0000u;   // write pass: the split point (segment - first segment) << 16 | characters / 16; 0xFFFF....: none
    uint32_t bd_flag = 0u, nx_flag = 0u;  // write pass: tile_flags of the tile after next / of the next tile
    auto fetch_lane_offsets = [&](uint64_t t) {
        const uint64_t i = (t << 6) + lane;
        const bool valid = i < n;
        nx_o0 = off[valid ? i : n]; nx_o1 = off[valid ? i + 1 : n];
        nx_k = valid ? match_id[i] : -1;
        if (WRITE) { nx_oo0 = out_off[valid ? i : n]; nx_oo1 = out_off[valid ? i + 1 : n]; }
        if (WRITE && PAIR == 2) {
            nx_split = (valid && part == 1u) ? split[i] : 0u;
            nx_at = valid ? split_at[i] : 0xFFFF0000u;
        }
    };
    // (uniform addresses, but the loads must be VECTOR loads: scalar ones land in SGPRs, of which the kernel has none to spare -- the
    // compiler parks them in VGPR lanes at once, and for that waits for them where they are issued: a trip to memory per tile)
    auto in_vgpr = [&](uint64_t v) -> uint64_t {
        uint32_t lo, hi;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "s"(static_cast<uint32_t>(v)));
        asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "s"(static_cast<uint32_t>(v >> 32)));
        return static_cast<uint64_t>(hi) << 32 | lo;
    };
    auto fetch_bounds = [&](uint64_t t_uniform) {
        if (t_uniform >= tiles) return;
        const uint64_t t = in_vgpr(t_uniform);
        const uint64_t e0 = t << 6, e1 = min(n, e0 + 64u);
        bd_lo = off[e0]; bd_hi = off[e1];
        if (WRITE) { bd_olo = out_off[e0]; bd_ohi = out_off[e1]; bd_flag = tile_flags[t]; }
    };
    const uint64_t tile0 = static_cast<uint64_t>(blockIdx.x) * (cfg.waves / PAIR) + wave;
    if (tile0 < tiles) {
        fetch_lane_offsets(tile0);
        if (WRITE) nx_flag = tile_flags[tile0];
        fetch_bounds(tile0 + wstride);
    }
    for (uint64_t tile = tile0; tile < tiles; tile += wstride) {
        const uint64_t i = (tile << 6) + lane;
        const bool valid = i < n;
        const uint64_t o0 = nx_o0, o1 = nx_o1;
        const int32_t k = nx_k;
        const uint64_t oo0 = nx_oo0, oo1 = nx_oo1;
        const uint32_t split_bytes = nx_split, split_point = nx_at;
        const bool clean = WRITE && uni(nx_flag) != 0u;  // no character of the tile's captures takes an escape: they are copied as they are
        uint32_t tile_dirty = 0u;                           // sizes pass: what becomes tile_flags[tile]
        const uint32_t group_lines = static_cast<uint32_t>(min(static_cast<uint64_t>(64), n - (tile << 6)));
        // this tile's bytes, when they were fetched ahead: registers -> LDS (the staging areas are free: the barrier that ends a round)
        const bool staged = pf_valid;
        if (staged) {
            stage_put<NQ>(in_stage, pf_nch, ptid, threads, pf);
            if (cfg.caps_bytes) stage_put<NC>(caps_stage, pf_cch, ptid, threads, pfc);
        }
        // the next tile
        pf_valid = false;
        {
            const uint64_t t1 = tile + wstride;
            if (t1 < tiles) {
                // (everything that is read here arrived long ago; it is read BEFORE the first new load is issued, or the wait for it
                // -- the counter of loads in flight cannot tell old from new -- would be a wait for the new loads: a trip to memory per tile)
                const uint64_t n_lo = uni(static_cast<uint64_t>(bd_lo)), n_hi = uni(static_cast<uint64_t>(bd_hi)), n_olo = uni(bd_olo), n_ohi = uni(bd_ohi);
                nx_flag = uni(bd_flag);
                __builtin_amdgcn_sched_barrier(0);
                fetch_lane_offsets(t1);
                const uint32_t lines1 = static_cast<uint32_t>(min(static_cast<uint64_t>(64), n - (t1 << 6)));
                const uint32_t words = lines1 * static_cast<uint32_t>(slots);
                bool one_round = WRITE && (n_hi - n_lo) + static_cast<uint32_t>(reinterpret_cast<uintptr_t>(data + n_lo) & 15u) <= cfg.in_bytes;
                if (WRITE) one_round = one_round && (n_ohi - n_olo) + static_cast<uint32_t>(reinterpret_cast<uintptr_t>(out + n_olo) & 15u) <= cfg.out_bytes;
                if (cfg.caps_bytes)
                    one_round = one_round && (words & 3u) == 0u && (words >> 2) <= threads * NC && (reinterpret_cast<uintptr_t>(caps) & 15u) == 0u;
                if (one_round) {
                    pf_nch = stage_fetch<NQ>(data, data_end, n_lo, n_hi, ptid, threads, pf);
                    if (cfg.caps_bytes) {
                        const uint4* rows = reinterpret_cast<const uint4*>(caps + (t1 << 6) * static_cast<uint64_t>(slots));
                        pf_cch = words >> 2;
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const uint32_t c = threads * q + ptid;
                            pfc[q] = make_uint4(0u, 0u, 0u, 0u);
                            if (c < pf_cch) pfc[q] = rows[c];
                        }
                    }
                    pf_valid = true;
                }
                fetch_bounds(t1 + wstride);
            }
        }
        JX_STAMP(0);  // registers -> LDS of this tile (waits for its loads), loads of the next one issued
        uint32_t a = 0;
        while (a < group_lines) {
            // ---- the round: lanes [a, b) whose input (and output) fit the staging areas ----
            const uint64_t lo = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(o0), static_cast<int>(a))));
            const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(data + lo) & 15u);
            bool fits = lane >= a && valid && (o1 - lo) + skew <= cfg.in_bytes;
            uint64_t olo = 0;
            uint32_t oskew = 0;
            if (WRITE) {
                olo = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(oo0), static_cast<int>(a))));
                oskew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(out + olo) & 15u);
                fits = fits && (oo1 - olo) + oskew <= cfg.out_bytes;
            }
            const uint32_t cnt = static_cast<uint32_t>(__popcll(__ballot(fits)));
            if (cnt == 0) {
                // line a alone does not fit: the whole wave takes it
                const uint64_t ia = (tile << 6) + a;
                if (WRITE) { if (part == 0u) line_write_wave(tm, data, lo, ia, match_id, caps, slots, pt, lane, out + olo); }
                else {
                    const uint32_t t = line_size_wave(tm, data, lo, ia, match_id, caps, slots, pt, lane);
                    if (lane == 0) sizes[ia] = t;
                    tile_dirty = 1u;  // (its lines are written by line_write_wave either way; the flag only has to be safe)
                }
                a += 1;
                continue;
            }
            const uint32_t b = a + cnt;
            const uint64_t hi = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(o1), static_cast<int>(b - 1u))));
            const bool in_lds = staged && a == 0u;  // fetched ahead: the whole tile is this one round (b == group_lines)
            if (!in_lds) {
                uint4 v[NQ];
                const uint32_t nch = stage_fetch<NQ>(data, data_end, lo, hi, ptid, threads, v);
                stage_put<NQ>(in_stage, nch, ptid, threads, v);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see below
            }
            if (!cfg.caps_bytes) pair_barrier();
            const bool active = lane >= a && lane < b;
            if (cfg.caps_bytes) {
                // the round's capture rows are contiguous: stage them with coalesced 16-byte loads
                if (!in_lds) {
                    const uint64_t row0 = ((tile << 6) + a) * static_cast<uint64_t>(slots);     // in int32 units
                    const uint32_t words = (b - a) * static_cast<uint32_t>(slots);
                    const int32_t* src = caps + row0;
                    stage_words(src, words, reinterpret_cast<int32_t*>(caps_stage), ptid, 64u * PAIR);
                    // (a load whose store was predicated off is never waited for: without this the compiler has to assume, where
                    // this path and the fetched-ahead one meet, that a register of the work below is still a load's target, and
                    // waits for every load in flight there -- the next tile's among them)
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
                pair_barrier();
            }
            JX_STAMP(1);  // staging inside the round + the barrier that ends staging
            // ---- lane = line ----
            // The write pass: a lane's text leaves through ALIGNED 32-bit stores (a 32-bit LDS store off its alignment is
            // replayed at 64 cycles, measured: SQ_LDS_UNALIGNED_STALL was 88 % of the LDS-active cycles).  Up to three
            // bytes wait in `carry`; the first store of a writer also covers the last bytes of the text before it (the
            // lane before, or the other wave of the pair) with zeros, and those are written by that writer's byte stores
            // after the barrier below, later than every first store of the round.
            uint32_t wp = 0u, wp0 = 0u, head0 = 0u, carry = 0u, pend8 = 0u;  // pend8: BITS waiting in carry (0, 8, 16, 24)
            bool lane_dirty = false;
            uint32_t r_total = 0u, r_split = 0u, r_at = 0xFFFF0000u;  // sizes pass: what the lane stores for its line
            if (active) {
                const uint32_t line = lds_addr(in_stage) + skew + static_cast<uint32_t>(o0 - lo);  // LDS byte addresses
                uint32_t total = 0, unescaped = 0;  // sizes pass: the text's bytes, and what they would be without escapes
                auto put_4 = [&](uint32_t e) {  // four bytes
                    const uint64_t t = static_cast<uint64_t>(e) << pend8;
                    *(JX_LDS uint32_t*)(uintptr_t)wp = carry | static_cast<uint32_t>(t);
                    wp += 4u;
                    carry = static_cast<uint32_t>(t >> 32);
                };
                auto put_n = [&](uint32_t e, uint32_t count) {  // the low `count` (0..4) bytes of e; its other bytes are zero
                    const uint64_t t = static_cast<uint64_t>(e) << pend8;
                    const uint32_t acc = carry | static_cast<uint32_t>(t);
                    const uint32_t np8 = pend8 + 8u * count;
                    if (np8 >= 32u) {
                        *(JX_LDS uint32_t*)(uintptr_t)wp = acc;
                        wp += 4u;
                        carry = static_cast<uint32_t>(t >> 32);
                        pend8 = np8 - 32u;
                    } else {
                        carry = acc;
                        pend8 = np8;
                    }
                };
                auto put_e = [&](uint32_t e_lo, uint32_t e_hi, uint32_t extra) {  // the low 4 + extra (0..4) bytes of e_hi:e_lo; the others are zero
                    const uint64_t t = (static_cast<uint64_t>(e_hi) << 32 | e_lo) << pend8;
                    const uint32_t t2 = (e_hi >> 8) >> (24u - pend8);  // what the shift pushed out of the 64 bits
                    *(JX_LDS uint32_t*)(uintptr_t)wp = carry | static_cast<uint32_t>(t);
                    const uint32_t np8 = pend8 + 32u + 8u * extra;
                    if (np8 >= 64u) {
                        *(JX_LDS uint32_t*)(uintptr_t)(wp + 4u) = static_cast<uint32_t>(t >> 32);
                        wp += 8u;
                        carry = t2;
                    } else {
                        wp += 4u;
                        carry = static_cast<uint32_t>(t >> 32);
                    }
                    pend8 = np8 & 24u;
                };
                // The same without a branch, for text that more text of the same writer follows (at least four bytes that it stores
                // itself): the second dword is stored whether it is complete or not -- an incomplete one is what waits in `carry`, zeros
                // above it, and the writer's next store, at that very address, puts the whole dword there.  (Never the last store of a
                // writer: the zeros would land on the first bytes of the text behind it, whose writer stored them long before.)
                // Straight-line code: the four words of sixteen characters make one basic block, and their four table reads one
                // LDS round trip instead of four.
                auto put_e_fast = [&](uint32_t e_lo, uint32_t e_hi, uint32_t extra) {
                    const uint64_t t = (static_cast<uint64_t>(e_hi) << 32 | e_lo) << pend8;
                    const uint32_t t2 = (e_hi >> 8) >> (24u - pend8);
                    *(JX_LDS uint32_t*)(uintptr_t)wp = carry | static_cast<uint32_t>(t);
                    *(JX_LDS uint32_t*)(uintptr_t)(wp + 4u) = static_cast<uint32_t>(t >> 32);
                    const uint32_t np8 = pend8 + 32u + 8u * extra;   // 32 .. 88
                    carry = np8 >= 64u ? t2 : static_cast<uint32_t>(t >> 32);
                    wp += 4u + ((np8 >> 4) & 4u);
                    pend8 = np8 & 24u;
                };
                if (k >= 0) {
                    const int32_t* cp_g = caps + i * static_cast<uint64_t>(slots);
                    const int32_t* cp_l = reinterpret_cast<const int32_t*>(caps_stage) + (lane - a) * static_cast<uint32_t>(slots);
                    const uint32_t caps_row = lds_addr(caps_stage) + (lane - a) * static_cast<uint32_t>(slots) * 4u;
                    const uint32_t tmb = smem0 + cfg.tm_lds;   // FAST: the template arrays' LDS copy (see tl_seg)
                    const uint32_t tm_fixed = tmb + (cfg.n_rules + 1u) * 4u, tm_seg = tmb + tm_seg_rel;
                    auto cap = [&](int idx) -> int32_t {
                        if (FAST) return static_cast<int32_t>(lds_w(caps_row + static_cast<uint32_t>(idx) * 4u));
                        return cfg.caps_bytes ? cp_l[idx] : cp_g[idx];
                    };
                    auto t_seg_off = [&](uint32_t x) -> uint32_t { if (FAST) return lds_w(tmb + x * 4u); return cfg.tm_lds != 0xFFFFFFFFu ? tl_seg_off[x] : tm.seg_off[x]; };
                    auto t_fixed = [&](uint32_t x) -> uint32_t { if (FAST) return lds_w(tm_fixed + x * 4u); return cfg.tm_lds != 0xFFFFFFFFu ? tl_fixed[x] : tm.fixed_len[x]; };
                    auto seg_rec = [&](uint32_t x) -> uint4 {  // lit_off, lit_len, group of segment x
                        if (FAST) {  // one ds_read_b128
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 r = *(JX_LDS const u32x4*)(uintptr_t)(tm_seg + x * 16u);
                            return make_uint4(r.x, r.y, r.z, r.w);
                        }
                        if (cfg.tm_lds != 0xFFFFFFFFu) return tl_seg[x];
                        return make_uint4(tm.lit_off[x], tm.lit_len[x], static_cast<uint32_t>(tm.group[x]), 0u);
                    };
                    const uint32_t s0 = t_seg_off(k), s1 = t_seg_off(k + 1), fixed_k = t_fixed(k);
                    // ---- the split point (m, q): part 0 writes the segments before m, m's literal and -- m has a capture that
                    // is not null -- the opening quote and q characters of it (q a multiple of 16); part 1 the rest.  From the
                    // unescaped sizes: the first place at or past the middle of the text.  m == s1: no split. ----
                    uint32_t sp_m = s1, sp_q = 0u;
                    if (WRITE) {
                        // (the sizes pass left it in split_at)
                        if (PAIR == 2 && (split_point >> 16) != 0xFFFFu) { sp_m = s0 + (split_point >> 16); sp_q = (split_point & 0xFFFFu) << 4; }
                    }
                    // (the sizes pass chooses it inside its one loop over the segments, below: the first place at or past `half` of the
                    // unescaped text -- half of an ESTIMATE of its size, literals + line, so that no loop has to run ahead to add the
                    // captures up; the split only balances the two waves, any point is a correct one)
                    const uint32_t half = (WRITE || split == nullptr) ? 0u : ((fixed_k + static_cast<uint32_t>(o1 - o0)) * SPLIT_AT_256) >> 8;
                    const bool splittable = !WRITE && split != nullptr && s1 - s0 < 0xFFFFu;
                    uint32_t cum = 0u;  // sizes pass: the unescaped bytes of the segments so far
                    uint32_t s_from = s0, s_to = s1;
                    if (WRITE && PAIR == 2) {
                        if (part == 0u) s_to = sp_m < s1 ? sp_m + 1u : s1;
                        else s_from = sp_m;   // (sp_m == s1: nothing)
                    }
                    const uint32_t dst0 = WRITE ? lds_addr(out_stage) + oskew + static_cast<uint32_t>(oo0 - olo) + ((PAIR == 2 && part == 1u) ? split_bytes : 0u) : 0u;
                    wp = wp0 = dst0 & ~3u;
                    head0 = dst0 & 3u;
                    pend8 = head0 * 8u;
                    uint32_t lit_cum = 0u, before = 0u;  // sizes pass: literal bytes so far, bytes before the split point
                    if (!WRITE) total = fixed_k;
                    // The loop over the segments looks nothing up that it then waits for: a segment's record arrives while the segment
                    // two before it is written, its capture offsets (which need the record's group) during the one before it.
                    uint4 rec_n = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u), rec_nn = rec_n;
                    int32_t cb_n = -1, ce_n = -1;
                    if (s_from < s_to) {
                        rec_n = seg_rec(s_from);
                        if (s_from + 1u < s_to) rec_nn = seg_rec(s_from + 1u);
                        if (static_cast<int32_t>(rec_n.z) >= 0) { cb_n = cap(2 * static_cast<int32_t>(rec_n.z)); ce_n = cap(2 * static_cast<int32_t>(rec_n.z) + 1); }
                    }
                    for (uint32_t s = s_from; s < s_to; ++s) {
                        const uint4 rec = rec_n;
                        const int32_t cb = cb_n, ce = ce_n;
                        rec_n = rec_nn;
                        cb_n = ce_n = -1;
                        if (s + 1u < s_to && static_cast<int32_t>(rec_n.z) >= 0) { cb_n = cap(2 * static_cast<int32_t>(rec_n.z)); ce_n = cap(2 * static_cast<int32_t>(rec_n.z) + 1); }
                        if (s + 2u < s_to) rec_nn = seg_rec(s + 2u);
                        const bool at_split = WRITE && PAIR == 2 && s == sp_m;
                        const bool skip_literal = at_split && part == 1u;   // part 1 enters segment m behind its literal and quote
                        if (WRITE && !skip_literal) {
                            const uint32_t ll = rec.y, lo_l = rec.x;
                            if (FAST || cfg.lits_lds != 0xFFFFFFFFu) {
                                const uint32_t lit = smem0 + cfg.lits_lds + lo_l;
                                uint32_t q = 0;
                                for (; q + 16u <= ll; q += 16u) {  // four words per LDS round trip (the host aligns every literal to 4 bytes)
                                    const uint32_t w0 = lds_w(lit + q), w1 = lds_w(lit + q + 4u), w2 = lds_w(lit + q + 8u), w3 = lds_w(lit + q + 12u);
                                    put_4(w0); put_4(w1); put_4(w2); put_4(w3);
                                }
                                for (; q + 4u <= ll; q += 4u) put_4(lds_w(lit + q));
                                if (q < ll) put_n(lds_w(lit + q) & ((1u << (8u * (ll - q))) - 1u), ll - q);  // (the word's other bytes: the host's padding)
                            } else {
                                const uint8_t* lit = tm.lits + lo_l;
                                for (uint32_t q = 0; q < ll; ++q) put_n(lit[q], 1u);
                            }
                        }
                        const int32_t g = static_cast<int32_t>(rec.z);
                        if (!WRITE && split != nullptr) {
                            const uint32_t ll = rec.y;
                            lit_cum += ll;
                            const bool text = g >= 0 && cb >= 0;
                            const uint32_t len = text ? static_cast<uint32_t>(ce - cb) : 0u;
                            const uint32_t e = ll + (g < 0 ? 0u : (text ? len + 2u : 4u));
                            if (splittable && sp_m == s1 && cum + e > half) {
                                sp_m = s;
                                const uint32_t at = cum + ll + 1u;  // where the capture's characters begin
                                sp_q = (text && half > at) ? ((half - at) & ~15u) : 0u;
                                if (sp_q > (len & ~15u)) sp_q = len & ~15u;
                            }
                            cum += e;
                        }
                        if (g < 0) { if (!WRITE && s == sp_m) before = lit_cum + (total - fixed_k); continue; }
                        if (cb < 0) {
                            if (WRITE) { if (!skip_literal) put_4(0x6C6C756Eu); }  // null (part 0's, with the literal)
                            else { total += 4u; if (s == sp_m) before = lit_cum + (total - fixed_k); }
                            continue;
                        }
                        // the characters [c_from, c_to) of the capture
                        const int32_t c_from = skip_literal ? cb + static_cast<int32_t>(sp_q) : cb;
                        const int32_t c_to = (at_split && part == 0u) ? cb + static_cast<int32_t>(sp_q) : ce;
                        if (WRITE) {
                            if (!skip_literal) put_n(0x22u, 1u);
                            // Lanes run in lock step, so whatever one lane needs every lane pays for.  Four characters that
                            // are plain or only take a backslash -- nearly all of them -- are expanded without a branch: the
                            // mask of the characters to escape picks one of 16 byte-permute selectors (a 128-byte table
                            // in LDS) that lay the characters and their backslashes out in 4..8 bytes.  Control characters
                            // and bytes >= 0x80 take the byte-by-byte path.
                            auto put1 = [&](uint32_t v) {
                                uint32_t cnt;
                                const uint64_t e = esc_bytes(v, pt, cnt);
                                put_n(static_cast<uint32_t>(e), cnt < 4u ? cnt : 4u);
                                if (cnt > 4u) put_n(static_cast<uint32_t>(e >> 32), cnt - 4u);
                            };
                            auto put4_plain = [&](uint32_t w) {  // every byte in 0x20..0x7F
                                // (v + 0x7F sets bit 7 of such a byte exactly when v != 0, and nothing carries)
                                const uint32_t nq = (w ^ 0x22222222u) + 0x7F7F7F7Fu, nb = (w ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu;
                                const uint32_t m = (~(nq & nb) & 0x80808080u) >> 7;  // bit 8 j: character j takes a backslash
                                const uint32_t t = m | (m >> 7);
                                const uint32_t idx = (t | (t >> 14)) & 15u;  // bit j: character j
                                const uint32_t sel = perm_tab + idx * 8u;
                                const uint32_t s_lo = *(JX_LDS const uint32_t*)(uintptr_t)sel, s_hi = *(JX_LDS const uint32_t*)(uintptr_t)(sel + 4u);
                                put_e(__builtin_amdgcn_perm(0x5C5C5C5Cu, w, s_lo), __builtin_amdgcn_perm(0x5C5C5C5Cu, w, s_hi), __popc(idx));
                            };
                            auto odd = [&](uint32_t w) {  // bit 7 of a byte: a control character or a byte >= 0x80
                                return ~((w & 0x7F7F7F7Fu) + 0x60606060u) | w;
                            };
                            auto put4 = [&](uint32_t w) {
                                if (odd(w) & 0x80808080u) {
                                    for (int q = 0; q < 4; ++q) put1((w >> (8 * q)) & 0xFFu);
                                    return;
                                }
                                put4_plain(w);
                            };
                            auto put16 = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {  // one test for the sixteen
                                if ((odd(w0) | odd(w1) | odd(w2) | odd(w3)) & 0x80808080u) { put4(w0); put4(w1); put4(w2); put4(w3); }
                                else { put4_plain(w0); put4_plain(w1); put4_plain(w2); put4_plain(w3); }
                            };
                            auto esc_index = [&](uint32_t w) {  // bit j: character j of the word (all in 0x20..0x7F) takes a backslash
                                const uint32_t nq = (w ^ 0x22222222u) + 0x7F7F7F7Fu, nb = (w ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu;
                                const uint32_t m = (~(nq & nb) & 0x80808080u) >> 7;
                                const uint32_t t = m | (m >> 7);
                                return (t | (t >> 14)) & 15u;
                            };
                            auto put16_fast = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {  // sixteen characters that more follow
                                if ((odd(w0) | odd(w1) | odd(w2) | odd(w3)) & 0x80808080u) { put4(w0); put4(w1); put4(w2); put4(w3); return; }
                                const uint32_t i0 = esc_index(w0), i1 = esc_index(w1), i2 = esc_index(w2), i3 = esc_index(w3);
                                const uint32_t a0 = perm_tab + i0 * 8u, a1 = perm_tab + i1 * 8u, a2 = perm_tab + i2 * 8u, a3 = perm_tab + i3 * 8u;
                                const uint32_t l0 = lds_w(a0), h0 = lds_w(a0 + 4u), l1 = lds_w(a1), h1 = lds_w(a1 + 4u), l2 = lds_w(a2), h2 = lds_w(a2 + 4u),
                                               l3 = lds_w(a3), h3 = lds_w(a3 + 4u);
                                put_e_fast(__builtin_amdgcn_perm(0x5C5C5C5Cu, w0, l0), __builtin_amdgcn_perm(0x5C5C5C5Cu, w0, h0), __popc(i0));
                                put_e_fast(__builtin_amdgcn_perm(0x5C5C5C5Cu, w1, l1), __builtin_amdgcn_perm(0x5C5C5C5Cu, w1, h1), __popc(i1));
                                put_e_fast(__builtin_amdgcn_perm(0x5C5C5C5Cu, w2, l2), __builtin_amdgcn_perm(0x5C5C5C5Cu, w2, h2), __popc(i2));
                                put_e_fast(__builtin_amdgcn_perm(0x5C5C5C5Cu, w3, l3), __builtin_amdgcn_perm(0x5C5C5C5Cu, w3, h3), __popc(i3));
                            };
                            // the last one to three characters of a capture: the same expansion over a word filled up with plain
                            // characters, cut to the bytes that count
                            auto put_tail = [&](uint32_t w, uint32_t rem) {
                                const uint32_t keep = (1u << (8u * rem)) - 1u;
                                const uint32_t wf = (w & keep) | (0x41414141u & ~keep);
                                const uint32_t t7 = (wf & 0x7F7F7F7Fu) + 0x60606060u;
                                if ((~t7 | wf) & 0x80808080u) {
                                    for (uint32_t q = 0; q < rem; ++q) put1((w >> (8u * q)) & 0xFFu);
                                    return;
                                }
                                const uint32_t nq = (wf ^ 0x22222222u) + 0x7F7F7F7Fu, nb = (wf ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu;
                                const uint32_t m = (~(nq & nb) & 0x80808080u) >> 7;
                                const uint32_t t = m | (m >> 7);
                                const uint32_t idx = (t | (t >> 14)) & 15u;
                                const uint32_t sel = perm_tab + idx * 8u;
                                const uint32_t s_lo = *(JX_LDS const uint32_t*)(uintptr_t)sel, s_hi = *(JX_LDS const uint32_t*)(uintptr_t)(sel + 4u);
                                const uint32_t cnt = rem + __popc(idx);  // 1..6 bytes
                                const uint64_t e = (static_cast<uint64_t>(__builtin_amdgcn_perm(0x5C5C5C5Cu, wf, s_hi)) << 32 | __builtin_amdgcn_perm(0x5C5C5C5Cu, wf, s_lo)) &
                                                   ((1ull << (8u * cnt)) - 1ull);
                                put_n(static_cast<uint32_t>(e), cnt < 4u ? cnt : 4u);
                                put_n(static_cast<uint32_t>(e >> 32), cnt > 4u ? cnt - 4u : 0u);
                            };
                            // the capture's bytes: aligned 32-bit reads (one off its alignment is replayed too) joined by v_alignbyte
                            auto sweep = [&](auto&& on_chunk_more, auto&& on_chunk, auto&& on_word, auto&& on_tail) {
                                int32_t p = c_from;
                                uint32_t ap = (line + static_cast<uint32_t>(c_from)) & ~3u;
                                const uint32_t mis = (line + static_cast<uint32_t>(c_from)) & 3u;
                                uint32_t prev = lds_w(ap);
                                for (; p + 32 <= c_to; p += 16, ap += 16u) {  // sixteen characters that sixteen more follow
                                    const uint32_t d0 = lds_w(ap + 4u), d1 = lds_w(ap + 8u), d2 = lds_w(ap + 12u), d3 = lds_w(ap + 16u);
                                    on_chunk_more(__builtin_amdgcn_alignbyte(d0, prev, mis), __builtin_amdgcn_alignbyte(d1, d0, mis), __builtin_amdgcn_alignbyte(d2, d1, mis),
                                                  __builtin_amdgcn_alignbyte(d3, d2, mis));
                                    prev = d3;
                                }
                                for (; p + 16 <= c_to; p += 16, ap += 16u) {  // four words per LDS round trip
                                    const uint32_t d0 = lds_w(ap + 4u), d1 = lds_w(ap + 8u), d2 = lds_w(ap + 12u), d3 = lds_w(ap + 16u);
                                    on_chunk(__builtin_amdgcn_alignbyte(d0, prev, mis), __builtin_amdgcn_alignbyte(d1, d0, mis), __builtin_amdgcn_alignbyte(d2, d1, mis),
                                             __builtin_amdgcn_alignbyte(d3, d2, mis));
                                    prev = d3;
                                }
                                for (; p + 4 <= c_to; p += 4, ap += 4u) {
                                    const uint32_t d = lds_w(ap + 4u);
                                    on_word(__builtin_amdgcn_alignbyte(d, prev, mis));
                                    prev = d;
                                }
                                if (p < c_to) on_tail(__builtin_amdgcn_alignbyte(lds_w(ap + 4u), prev, mis), static_cast<uint32_t>(c_to - p));  // (may read past the line: LDS)
                            };
                            if (clean)   // the sizes pass found nothing to escape in this tile: the characters as they are
                            {
                                auto copy16 = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { put_4(w0); put_4(w1); put_4(w2); put_4(w3); };
                                sweep(copy16, copy16, [&](uint32_t w) { put_4(w); }, [&](uint32_t w, uint32_t rem) { put_n(w & ((1u << (8u * rem)) - 1u), rem); });
                            } else
                                sweep(put16_fast, put16, put4, put_tail);
                            if (!(at_split && part == 0u)) put_n(0x22u, 1u);
                        } else {
                            // four characters at a time: 4 + one per quote / backslash (+ one per byte >= 0x80 that becomes
                            // two bytes of UTF-8); a word with a control character is counted byte by byte
                            uint32_t t = 2u;
                            int32_t p = cb;
                            uint32_t ap = (line + static_cast<uint32_t>(cb)) & ~3u;
                            const uint32_t mis = (line + static_cast<uint32_t>(cb)) & 3u;
                            uint32_t prev = lds_w(ap);
                            const bool is_m = s == sp_m;
                            const int32_t p_split = cb + static_cast<int32_t>(sp_q);
                            auto count4 = [&](uint32_t w) {
                                const uint32_t tt = (w & 0x7F7F7F7Fu) + 0x60606060u;
                                if ((~(tt | w)) & 0x80808080u) {  // some byte < 0x20
                                    for (int q = 0; q < 4; ++q) t += esc_len((w >> (8 * q)) & 0xFFu, pt);
                                } else {
                                    // (on the low seven bits of every byte, where v + 0x7F sets bit 7 exactly when v != 0 and nothing
                                    // carries; a byte >= 0x80 is no quote whatever its low bits say)
                                    const uint32_t w7 = w & 0x7F7F7F7Fu;
                                    const uint32_t nq = (w7 ^ 0x22222222u) + 0x7F7F7F7Fu, nb = (w7 ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu;
                                    t += 4u + __popc(~(nq & nb) & ~w & 0x80808080u) + (pt ? 0u : __popc(w & 0x80808080u));
                                }
                            };
                            // bytes before the split point: the literals so far, the captures before this one, the opening quote and
                            // the characters before p_split (t counts both quotes: t - 1)
                            auto mark = [&]() { before = lit_cum + (total - fixed_k) + t - 1u; };
                            for (; p + 16 <= ce; p += 16, ap += 16u) {  // four aligned words per LDS round trip
                                if (is_m && p == p_split) mark();
                                const uint32_t d0 = lds_w(ap + 4u), d1 = lds_w(ap + 8u), d2 = lds_w(ap + 12u), d3 = lds_w(ap + 16u);
                                count4(__builtin_amdgcn_alignbyte(d0, prev, mis));
                                count4(__builtin_amdgcn_alignbyte(d1, d0, mis));
                                count4(__builtin_amdgcn_alignbyte(d2, d1, mis));
                                count4(__builtin_amdgcn_alignbyte(d3, d2, mis));
                                prev = d3;
                            }
                            if (is_m && p == p_split) mark();  // (the split point is the end of the 16-byte steps)
                            for (; p + 4 <= ce; p += 4, ap += 4u) {
                                const uint32_t d = lds_w(ap + 4u);
                                count4(__builtin_amdgcn_alignbyte(d, prev, mis));
                                prev = d;
                            }
                            if (p < ce) {  // the last one to three characters: a word filled up with plain ones
                                const uint32_t rem = static_cast<uint32_t>(ce - p), keep = (1u << (8u * rem)) - 1u;
                                count4((__builtin_amdgcn_alignbyte(lds_w(ap + 4u), prev, mis) & keep) | (0x41414141u & ~keep));
                                t -= 4u - rem;
                            }
                            total += t;
                        }
                    }
                    if (!WRITE && split != nullptr) {
                        r_split = sp_m < s1 ? before : total;
                        r_at = sp_m < s1 ? ((sp_m - s0) << 16 | (sp_q >> 4)) : 0xFFFF0000u;
                        unescaped = cum;
                    }
                }
                r_total = total;
                lane_dirty = total != unescaped;
            }
            if (!WRITE) {
                // The next tile's offsets (loaded while this one was staged) have long arrived.  Said HERE, before this tile's stores
                // are issued: the counter of memory operations in flight is one for loads and stores, in order, so at the top of the
                // next tile "the offsets are there" would read "nothing is in flight" -- a wait for these stores to be acknowledged.
                asm volatile("" : : "v"(nx_o0), "v"(nx_o1), "v"(nx_k));
                if (active) {
                    sizes[i] = r_total;
                    if (split != nullptr && k >= 0) { split[i] = r_split; split_at[i] = r_at; }
                }
                if (__ballot(lane_dirty) != 0ull) tile_dirty = 1u;
            }
            JX_STAMP(2);  // the lanes' work
            if (WRITE) {
                // the bytes still waiting in the carries: byte stores (the rest of such a dword is the next writer's), once every
                // writer of the round has done its dword stores.  (A writer that wrote no dword leaves the bytes before its text alone.)
                pair_barrier();
                JX_STAMP(3);  // waiting for the other wave of the pair
                if (active && k >= 0)
                    for (uint32_t q = (wp == wp0 ? head0 : 0u); q < (pend8 >> 3); ++q) lds_put_u8(wp + q, (carry >> (8u * q)) & 0xFFu);
            }
            if (WRITE) {
                // ---- flush: the round's text is the contiguous span [olo, ohi) of the output ----
                pair_barrier();
                JX_STAMP(4);  // carries + barrier
                const uint64_t ohi = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(oo1), static_cast<int>(b - 1u))));
                const uint32_t span = static_cast<uint32_t>(ohi - olo);
                uint8_t* g_al = out + olo - oskew;  // 16-byte aligned
                const uint32_t nch = (span + oskew + 15u) >> 4;
                // four chunks per lane and LDS round trip; the two chunks at the ends of the span (the neighbouring bytes are another
                // tile's) leave as dwords and bytes out of the registers -- no load-store chain per byte
                const uint32_t lds_out = lds_addr(out_stage);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                for (uint32_t c0 = 0; c0 < nch; c0 += 4u * threads) {
                    u32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t c = c0 + threads * q + ptid;
                        if (c < nch) v[q] = *(JX_LDS const u32x4*)(uintptr_t)(lds_out + (c << 4));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t c = c0 + threads * q + ptid;
                        if (c >= nch) continue;
                        const uint32_t first = c << 4;
                        if (first >= oskew && first + 16u <= oskew + span) {
                            *reinterpret_cast<uint4*>(g_al + first) = make_uint4(v[q].x, v[q].y, v[q].z, v[q].w);
                        } else {
                            const uint32_t from = max(first, oskew), to = min(first + 16u, oskew + span);   // the bytes of the chunk that are text
                            const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const uint32_t b0 = first + 4u * d;
                                if (from <= b0 && b0 + 4u <= to) *reinterpret_cast<uint32_t*>(g_al + b0) = w[d];
                                else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        if (b0 + r >= from && b0 + r < to) g_al[b0 + r] = static_cast<uint8_t>(w[d] >> (8 * r));
                                }
                            }
                        }
                    }
                }
            }
            // the staging areas are reused by the next round
            pair_barrier();
            JX_STAMP(5);  // flush + barrier
            a = b;
        }
        if (!WRITE && tile_flags != nullptr && lane == 0u) tile_flags[tile] = tile_dirty ? 0u : 1u;
    }
#ifdef GX_DEV
    if (lane == 0u)
        for (int q = 0; q < 6; ++q) atomicAdd(&jx_phase[(WRITE ? 8 : 0) + q], ph[q]);
#endif
}

// ---- the sizes pass without the text (gx_text_to_jsonl): the split pass left a bit per byte of the text that takes one more byte
// inside a JSON string (gx_ingest.hip: Chunk::esc), and the text holds no byte that takes five more -- a capture's escaped length is
// its length plus the bits of its range.  One lane per line, no staging: a line's offsets, its capture row and the few words of bits
// its captures cover come through the caches (neighbouring lanes, neighbouring lines).  What it leaves is what k_jsonl_tile<.., false,
// ..> leaves, value for value: sizes[i], split[i] / split_at[i] (where the write pass's two waves divide the line's text, chosen by
// the same rule) and tile_flags[tile] (1: nothing in the tile's captures takes an escape). ----
// Everything a lane looks up sits in LDS: the template arrays (when they fit), per wave the tile's capture rows and the bit words of
// its span of the text, all fetched coalesced -- a segment is three lookups, each behind the one before (its record, its capture's
// offsets, the bits of that range): out of global memory that was three trips to L1 / L2 per segment and lane.
constexpr uint32_t SB_BIT_WORDS = 1024;   // per wave: 32 KiB of text per 64 lines; a tile beyond that reads its bits from global memory
template <typename OFF>
__global__ void __launch_bounds__(256) k_jsonl_sizes_bits(JsonlTemplates tm, const OFF* __restrict__ off, uint64_t n, const int32_t* __restrict__ match_id,
                                                         const int32_t* __restrict__ caps, int slots, const uint32_t* __restrict__ bits,
                                                         uint32_t* __restrict__ sizes, uint32_t* __restrict__ split, uint32_t* __restrict__ split_at,
                                                         uint32_t* __restrict__ tile_flags, uint32_t n_rules, uint32_t n_segs, uint32_t tm_lds, uint32_t caps_lds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sb_smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* l_seg_off = reinterpret_cast<uint32_t*>(sb_smem);
    uint32_t* l_fixed = l_seg_off + (n_rules + 1u);
    uint32_t* l_seg = l_fixed + n_rules;            // {lit_len, group} per segment
    if (tm_lds) {
        for (uint32_t q = threadIdx.x; q <= n_rules; q += blockDim.x) l_seg_off[q] = tm.seg_off[q];
        for (uint32_t q = threadIdx.x; q < n_rules; q += blockDim.x) l_fixed[q] = tm.fixed_len[q];
        for (uint32_t q = threadIdx.x; q < n_segs; q += blockDim.x) { l_seg[2u * q] = tm.lit_len[q]; l_seg[2u * q + 1u] = static_cast<uint32_t>(tm.group[q]); }
    }
    __syncthreads();
    uint32_t* l_bits = reinterpret_cast<uint32_t*>(sb_smem + tm_lds) + wave * SB_BIT_WORDS;
    int32_t* l_caps = reinterpret_cast<int32_t*>(sb_smem + tm_lds + 4u * SB_BIT_WORDS * 4u) + wave * 64u * static_cast<uint32_t>(slots);
    const uint64_t tiles = (n + 63) >> 6;
    const uint64_t wstride = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    for (uint64_t tile = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave; tile < tiles; tile += wstride) {
        const uint64_t i = (tile << 6) + lane;
        const uint32_t lines = static_cast<uint32_t>(min(static_cast<uint64_t>(64), n - (tile << 6)));
        bool dirty = false;
        int32_t k = -1;
        uint64_t o0 = 0, o1 = 0;
        if (i < n) { k = match_id[i]; o0 = off[i]; o1 = off[i + 1]; }
        // the tile's span of the text and its bit words
        const uint64_t lo = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(o0), 0)));
        const uint64_t hi = uni(static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(o1), static_cast<int>(lines - 1u))));
        const uint64_t w0 = lo >> 5;
        const uint64_t nw = hi > lo ? ((hi + 31) >> 5) - w0 : 0;
        const bool bits_lds = nw <= SB_BIT_WORDS;   // uniform
        if (bits_lds) for (uint32_t q = lane; q < static_cast<uint32_t>(nw); q += 64u) l_bits[q] = bits[w0 + q];
        if (caps_lds) {
            const uint64_t row0 = (tile << 6) * static_cast<uint64_t>(slots);
            const uint32_t words = lines * static_cast<uint32_t>(slots);
            for (uint32_t q = lane; q < words; q += 64u) l_caps[q] = caps[row0 + q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto word = [&](uint32_t w) -> uint32_t { return bits_lds ? l_bits[w] : bits[w0 + w]; };   // (word w of the tile's span)
        // the bits of the text's bytes [x, y), positions relative to the span's first word (32-bit arithmetic: a line is < 4 GiB from it)
        auto count = [&](uint32_t x, uint32_t y) -> uint32_t {
            if (x >= y) return 0u;
            const uint32_t wa = x >> 5, wb = (y - 1u) >> 5;
            const uint32_t first = 0xFFFFFFFFu << (x & 31u), last = 0xFFFFFFFFu >> (31u - ((y - 1u) & 31u));
            if (wa == wb) return __popc(word(wa) & first & last);
            uint32_t c = __popc(word(wa) & first) + __popc(word(wb) & last);
            for (uint32_t w = wa + 1u; w < wb; ++w) c += __popc(word(w));
            return c;
        };
        const uint32_t line_rel = static_cast<uint32_t>(o0 - (w0 << 5));   // this lane's line, in bytes from the span's first word
        if (i < n) {
            uint32_t total = 0u;
            if (k >= 0) {
                const int32_t* cp = caps_lds ? l_caps + lane * static_cast<uint32_t>(slots) : caps + i * static_cast<uint64_t>(slots);
                const uint32_t s0 = tm_lds ? l_seg_off[k] : tm.seg_off[k], s1 = tm_lds ? l_seg_off[k + 1] : tm.seg_off[k + 1];
                const uint32_t fixed_k = tm_lds ? l_fixed[k] : tm.fixed_len[k];
                const uint32_t half = ((fixed_k + static_cast<uint32_t>(o1 - o0)) * SPLIT_AT_256) >> 8;
                const bool splittable = s1 - s0 < 0xFFFFu;
                uint32_t sp_m = s1, sp_q = 0u, cum = 0u, lit_cum = 0u, before = 0u;
                total = fixed_k;
                for (uint32_t s = s0; s < s1; ++s) {
                    const uint32_t ll = tm_lds ? l_seg[2u * s] : tm.lit_len[s];
                    const int32_t g = tm_lds ? static_cast<int32_t>(l_seg[2u * s + 1u]) : tm.group[s];
                    int32_t cb = -1, ce = -1;
                    if (g >= 0) { cb = cp[2 * g]; ce = cp[2 * g + 1]; }
                    lit_cum += ll;
                    const bool text = g >= 0 && cb >= 0;
                    const uint32_t len = text ? static_cast<uint32_t>(ce - cb) : 0u;
                    const uint32_t e = ll + (g < 0 ? 0u : (text ? len + 2u : 4u));
                    if (splittable && sp_m == s1 && cum + e > half) {
                        sp_m = s;
                        const uint32_t at = cum + ll + 1u;
                        sp_q = (text && half > at) ? ((half - at) & ~15u) : 0u;
                        if (sp_q > (len & ~15u)) sp_q = len & ~15u;
                    }
                    cum += e;
                    if (g < 0) { if (s == sp_m) before = lit_cum + (total - fixed_k); continue; }
                    if (cb < 0) { total += 4u; if (s == sp_m) before = lit_cum + (total - fixed_k); continue; }
                    const uint32_t x = line_rel + static_cast<uint32_t>(cb);
                    if (s == sp_m) before = lit_cum + (total - fixed_k) + 1u + sp_q + count(x, x + sp_q);
                    total += 2u + len + count(x, x + len);
                }
                split[i] = sp_m < s1 ? before : total;
                split_at[i] = sp_m < s1 ? ((sp_m - s0) << 16 | (sp_q >> 4)) : 0xFFFF0000u;
                dirty = total != cum;
            }
            sizes[i] = total;
        }
        const bool any_dirty = __ballot(dirty) != 0ull;
        if (lane == 0u) tile_flags[tile] = any_dirty ? 0u : 1u;
        // (the wave's LDS is rewritten by its next tile)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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
// the same from u8 rows (gx_batch_opts.compact_results = 2: int8 id, offsets with 0xFF = unset)
__global__ void __launch_bounds__(256) k_unpack_results8(const uint8_t* __restrict__ rows, uint64_t n, int slots, int32_t* __restrict__ match_id,
                                                        int32_t* __restrict__ caps) {
    const uint64_t width = static_cast<uint64_t>(slots) + 1;
    const uint64_t total = n * width;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint64_t line = t / width;
        const uint32_t col = static_cast<uint32_t>(t - line * width);
        const uint8_t v = rows[t];
        if (col == 0) match_id[line] = static_cast<int8_t>(v);
        else caps[line * static_cast<uint64_t>(slots) + (col - 1)] = v == 0xFFu ? -1 : static_cast<int32_t>(v);
    }
}
}  // namespace

hipError_t launch_unpack_results8(const uint8_t* rows, uint64_t n, int slots, int32_t* match_id, int32_t* caps, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n * (static_cast<uint64_t>(slots) + 1) + 255) / 256;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    hipLaunchKernelGGL(k_unpack_results8, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, rows, n, slots, match_id, caps);
    return hipGetLastError();
}
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

// counts[0] = lines with match_id >= 0, counts[1] = lines with match_id <= -2 (capture regexp rejected the line)
namespace {
__global__ void __launch_bounds__(256) k_count_outcomes(const int32_t* __restrict__ match_id, uint64_t n, unsigned long long* __restrict__ counts) {
    // (one pair of atomics per workgroup: atomics on one cache line take their turns at 11 ns apiece -- one pair per WAVE of 2 048
    // workgroups made this kernel 108 us for 10 M lines, the read of 40 MB a tenth of that)
    __shared__ uint32_t sums[2][4];
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x * 4u;
    uint32_t matched = 0, rejected = 0;
    for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4u; i < n; i += stride) {
        int32_t k[4] = {-1, -1, -1, -1};
        if (i + 4 <= n && (reinterpret_cast<uintptr_t>(match_id) & 15u) == 0u) {
            const int4 v = *reinterpret_cast<const int4*>(match_id + i);
            k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
        } else {
            for (int q = 0; q < 4; ++q) if (i + q < n) k[q] = match_id[i + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            matched += k[q] >= 0 ? 1u : 0u;
            rejected += k[q] <= -2 ? 1u : 0u;
        }
    }
    matched = wave_sum(matched);
    rejected = wave_sum(rejected);
    if ((threadIdx.x & 63u) == 0) { sums[0][threadIdx.x >> 6] = matched; sums[1][threadIdx.x >> 6] = rejected; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = sums[0][0] + sums[0][1] + sums[0][2] + sums[0][3], r = sums[1][0] + sums[1][1] + sums[1][2] + sums[1][3];
        if (m) atomicAdd(counts, static_cast<unsigned long long>(m));
        if (r) atomicAdd(counts + 1, static_cast<unsigned long long>(r));
    }
}
}  // namespace

hipError_t launch_count_outcomes(const int32_t* match_id, uint64_t n, unsigned long long* d_counts, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_counts, 0, 16, stream);
    if (e != hipSuccess || n == 0) return e;
    uint64_t blocks = std::min<uint64_t>((n + 1023) / 1024, 256u * 2u);
    hipLaunchKernelGGL(k_count_outcomes, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, match_id, n, d_counts);
    return hipGetLastError();
}

// workspace: u32 sizes[n] | u64 block_sums[nblocks + 2] | u32 split[n] (bytes of a line's text before its split point) |
// u32 split_at[n] (the split point itself) | u32 tile_flags[tiles] (1: no captured character of the tile takes an escape)
static uint64_t jsonl_ws_sums(uint64_t n) { return (n * 4 + 15) & ~static_cast<uint64_t>(15); }
static uint64_t jsonl_ws_split(uint64_t n) {
    const uint64_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return jsonl_ws_sums(n) + (((nblocks + 2) * 8 + 15) & ~static_cast<uint64_t>(15));
}
size_t jsonl_workspace_bytes(uint64_t n) { return static_cast<size_t>(jsonl_ws_split(n) + n * 8 + ((n + 63) >> 6) * 4 + 64); }

namespace {
// LDS plan of the tile kernels for lines of mean_in bytes producing mean_out bytes of text (0: sizes pass).
bool plan_jsonl_tile(const GxJsonl& tm, int slots, uint32_t mean_in, uint32_t mean_out, JsonlTileCfg* cfg) {
    const uint32_t LDS = 163840;
    JsonlTileCfg c{};
    uint32_t used = 0;
    c.lits_bytes = tm.lits_bytes;
    c.lits_lds = 0xFFFFFFFFu;
    c.tm_lds = 0xFFFFFFFFu;
    c.n_rules = tm.n_rules;
    c.n_segs = tm.n_segs;
    if (tm.lits_bytes <= 24u * 1024u) { c.lits_lds = 0; used = (tm.lits_bytes + 15u) & ~15u; }
    const uint32_t tm_bytes = (((2u * tm.n_rules + 1u) * 4u + 15u) & ~15u) + 16u * tm.n_segs;  // (k_jsonl_tile: tl_seg)
    if (tm_bytes <= 20u * 1024u) { c.tm_lds = used; used += (tm_bytes + 15u) & ~15u; }
    c.perm_lds = used;
    used += 128u;
    c.caps_bytes = slots > 0 && slots <= 32 ? 64u * static_cast<uint32_t>(slots) * 4u : 0u;
    c.in_bytes = std::min<uint32_t>((64u * std::max<uint32_t>(mean_in, 1u) + 64u + 15u) & ~15u, 16384u);
    c.out_bytes = mean_out ? std::min<uint32_t>((64u * (mean_out + mean_out / 8u) + 64u + 15u) & ~15u, 49152u) : 0u;
    c.stage0 = used;
    const uint32_t per_wave = c.in_bytes + c.out_bytes + c.caps_bytes;
    uint32_t w = (LDS - used) / per_wave;
    if (w < 1) return false;
    c.waves = std::min<uint32_t>(w, 12u);   // (the sizes pass: launch bounds of k_jsonl_tile; the write pass runs pairs)
    *cfg = c;
    return true;
}

// one instantiation of the tile kernel: offsets width, pass, and whether everything the lanes look up sits in LDS
template <bool WRITE, int PAIR>
hipError_t launch_jsonl_tile(const JsonlTemplates& t, const JsonlTileCfg& cfg, const GxBatch& b, int slots, int passthrough, unsigned blocks, unsigned threads,
                             uint32_t lds, uint32_t* sizes, const uint64_t* line_out_off, uint8_t* out, uint32_t* split, uint32_t* split_at, uint32_t* tile_flags,
                             hipStream_t stream) {
    const bool fast = cfg.tm_lds != 0xFFFFFFFFu && cfg.lits_lds != 0xFFFFFFFFu && cfg.caps_bytes != 0u;
    auto go = [&](auto kernel, auto offsets) -> hipError_t {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, stream, t, cfg, static_cast<const uint8_t*>(b.data), offsets, b.n, b.match_id, b.caps, slots,
                           passthrough, sizes, line_out_off, out, split, split_at, tile_flags);
        return hipGetLastError();
    };
    if (b.offsets64) {
        const uint64_t* o = static_cast<const uint64_t*>(b.offsets);
        return fast ? go(&k_jsonl_tile<uint64_t, WRITE, PAIR, true>, o) : go(&k_jsonl_tile<uint64_t, WRITE, PAIR, false>, o);
    }
    const uint32_t* o = static_cast<const uint32_t*>(b.offsets);
    return fast ? go(&k_jsonl_tile<uint32_t, WRITE, PAIR, true>, o) : go(&k_jsonl_tile<uint32_t, WRITE, PAIR, false>, o);
}
}  // namespace

// Pass 1 + scan: line_out_off[0..n] (device, u64) receives the output offset of every line's text and, in
// [n], the total size.  workspace: jsonl_workspace_bytes(n).
hipError_t launch_jsonl_sizes(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, uint32_t mean_in, uint64_t* line_out_off,
                              void* workspace, hipStream_t stream, const uint32_t* esc_bits) {
    if (b.n == 0) return hipMemsetAsync(line_out_off, 0, 8, stream);
    const uint64_t nblocks = (b.n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint32_t* sizes = static_cast<uint32_t*>(workspace);
    uint64_t* block_sums = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(workspace) + jsonl_ws_sums(b.n));
    uint32_t* split = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(workspace) + jsonl_ws_split(b.n));
    uint32_t* split_at = split + b.n;
    uint32_t* tile_flags = split_at + b.n;
    JsonlTemplates t{tm.seg_off, tm.lit_off, tm.lit_len, tm.group, tm.fixed_len, tm.lits};
    JsonlTileCfg cfg;
    if (!plan_jsonl_tile(tm, slots, mean_in, 0, &cfg)) return hipErrorInvalidValue;
    if (esc_bits && !b.offsets64) {
        // (the text's escape bits are there and say that no byte takes more than one: no look at the text)
        const uint64_t tiles = (b.n + 63) >> 6;
        const uint64_t blocks = std::min<uint64_t>((tiles + 3) / 4, 256u * 16u);
        // LDS: the template arrays when they are small, per wave 4 KiB of bit words and, when they fit beside all that, the tile's capture rows
        const uint32_t n_rules = static_cast<uint32_t>(tm.n_rules), n_segs = static_cast<uint32_t>(tm.n_segs);
        uint32_t tm_lds = ((2u * n_rules + 1u + 2u * n_segs) * 4u + 15u) & ~15u;
        if (tm_lds > 24576u) tm_lds = 0u;
        const uint32_t bits_bytes = 4u * SB_BIT_WORDS * 4u, caps_bytes = 4u * 64u * static_cast<uint32_t>(slots) * 4u;
        const uint32_t caps_lds = (slots > 0 && tm_lds + bits_bytes + caps_bytes <= 65536u) ? 1u : 0u;
        hipLaunchKernelGGL(k_jsonl_sizes_bits<uint32_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), tm_lds + bits_bytes + (caps_lds ? caps_bytes : 0u), stream, t,
                           static_cast<const uint32_t*>(b.offsets), b.n, b.match_id, b.caps, slots, esc_bits, sizes, split, split_at, tile_flags, n_rules, n_segs, tm_lds,
                           caps_lds);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else {
        const uint32_t lds = cfg.stage0 + cfg.waves * (cfg.in_bytes + cfg.out_bytes + cfg.caps_bytes);
        const uint64_t tiles = (b.n + 63) >> 6;
        uint64_t blocks = std::min<uint64_t>((tiles + cfg.waves - 1) / cfg.waves, 256u * 4u);
        const hipError_t e = launch_jsonl_tile<false, 1>(t, cfg, b, slots, passthrough, static_cast<unsigned>(blocks), cfg.waves * 64u, lds, sizes, nullptr, nullptr,
                                                         split, split_at, tile_flags, stream);
        if (e != hipSuccess) return e;
    }
    if (nblocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(static_cast<unsigned>(nblocks)), dim3(SCAN_THREADS), 0, stream, sizes, b.n, block_sums);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, stream, block_sums, nblocks);
    hipLaunchKernelGGL(k_scan_write, dim3(static_cast<unsigned>(nblocks)), dim3(SCAN_THREADS), 0, stream, sizes, b.n, block_sums, nblocks, line_out_off);
    return hipGetLastError();
}

// Pass 2: write the text.
hipError_t launch_jsonl_write(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, uint32_t mean_in, uint32_t mean_out,
                              const uint64_t* line_out_off, uint8_t* out, void* workspace, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    JsonlTemplates t{tm.seg_off, tm.lit_off, tm.lit_len, tm.group, tm.fixed_len, tm.lits};
    JsonlTileCfg cfg;
    if (!plan_jsonl_tile(tm, slots, mean_in, std::max<uint32_t>(mean_out, 1u), &cfg)) return hipErrorInvalidValue;
    // two waves per tile (k_jsonl_tile, PAIR): a workgroup is one pair with one set of staging areas; the split points are
    // in the workspace the sizes pass of this batch used
    cfg.waves = 2;
    uint32_t* split = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(workspace) + jsonl_ws_split(b.n));
    uint32_t* split_at = split + b.n;
    uint32_t* tile_flags = split_at + b.n;
    const uint32_t lds = cfg.stage0 + cfg.in_bytes + cfg.out_bytes + cfg.caps_bytes;
    const uint64_t tiles = (b.n + 63) >> 6;
    const uint64_t blocks = std::min<uint64_t>(tiles, 256u * 16u);
    return launch_jsonl_tile<true, 2>(t, cfg, b, slots, passthrough, static_cast<unsigned>(blocks), 128u, lds, nullptr, line_out_off, out, split, split_at, tile_flags, stream);
}

#ifdef GX_DEV
hipError_t jsonl_dev_phases(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(jx_phase), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(jx_phase), z, sizeof(z));
    }
    return e;
}
#endif

}  // namespace gx
