// gx_tile_body.hpp -- the batch kernel of the Gorp match-and-extract hot path on gfx950: one wave per tile of 64 lines.
//
// Replaces, per line (lane-per-line, no cross-line state):
//   hot loop #1  PolyMatcher.match      core/autom/PolyMatcher.java:123-133
//                Automata.step/accept   core/autom/Automata.java:133-139
//   hot loop #2  JDKRegexpCookedExtraction.match/_constructMatch
//                                       core/jdkre/JDKRegexpCookedExtraction.java:36-59
//   driver       Gorp.extract           core/Gorp.java:159-186
//
// One wave owns one tile of 64 consecutive lines at a time.  The tile's bytes are one contiguous span of the CSR
// buffer: the wave fetches it with 16-byte-per-lane coalesced loads (every HBM byte is read exactly once, in whole
// cache lines) into registers one tile AHEAD, copies it into its LDS staging area, and then each lane walks its own
// line out of LDS.  All automaton tables (byte->class map, class-compressed rows, capture programs) live in LDS
// (TIER_LDS) or, for large definitions, the rows stay in global memory / L2 (TIER_L2).
//
// Self-loop acceleration, two levels:
//  * per state the host precomputes the longest run [lo,hi] of ASCII byte values on which the state loops to itself
//    with no capture operation.  While a lane sits in such a state it tests 16 staged bytes at once with SWAR
//    arithmetic and skips them if all lie in [lo,hi]; any other byte falls through to the exact one-byte steps.
//  * the widest such interval of the definition is the "hot" interval.  While the tile is still in registers, every
//    lane tests the chunks it holds against the hot interval (all 64 lanes busy on distinct data, no LDS traffic) and
//    the wave leaves one bit per 16-byte chunk in LDS.  A lane whose state loops on the whole hot interval then
//    jumps over a run of such chunks with one bitmap lookup instead of testing window after window.
//  Exactness never depends on either: a set bit / passed test is a fact about bytes on which the state provably
//  stays put; everything else takes the exact steps.
#pragma once
#include "gx_walk.hpp"
#include "gx_hop_dev.hpp"

namespace gx {

namespace {

// What the kernel reads and writes besides the table layout (GxLds).
struct TileIO {
    const uint8_t* image;        // the LDS table image, in global memory
    const uint8_t* at_global;    // TIER_L2: automaton rows in global memory
    const uint8_t* data;
    const void* off;
    uint64_t n;
    int32_t* match_id;
    int32_t* caps;
    uint16_t* packed;            // compact rows instead of match_id / caps
    unsigned long long* overflow;
    int32_t narrow;              // ... as u8 rows (gx_device.hpp: GxBatch::narrow)
    uint32_t* oversize_flag;
    uint32_t seq;
    int32_t max_groups;
    int32_t strip_eol;
    uint32_t* steal;             // tile counters of the launch's workgroups, u32[2][GX_STEAL_MAX * GX_STEAL_STRIDE] (gx_device.hpp: GxBatch::steal)
    uint32_t steal_parity;       // which of the two rows this launch draws from (it zeroes the other one for the next launch)
    uint32_t share64;            // 64ths of a workgroup's tiles that are handed out through its global counter (other workgroups may take them)
    uint8_t* wide_flags;         // WIDE (UTF-16 code units in `data`): [n], 1 for a line that holds a unit above 0xFF -- the per-line walk takes it again
    uint32_t* wide_any;          // ... and the launch's sequence number here when there is any such line
    int32_t* state_out;          // MODE 0 on dense rows (gx_match_batch): [n], the product-DFA state a line ends in (-1: the dead state)
#ifdef GX_DEV
    unsigned long long* stamps;  // developer build: [8] per wave -- per-phase cycle totals [0..3], begin / end on the chip's 100 MHz clock, tiles, XCC id
    uint32_t dev_flags;          // developer build: experiments (bit 1: nontemporal result stores;
                                 // bit 2: no result stores)
#endif
};

// Walk one automaton over the staged line [start, end) (offsets into the wave's staging area at LDS address
// `stage`), all lanes in lock step over 16-byte windows of their own line.  In every window a lane either proves
// with one SWAR test that all its bytes stay inside the current state's self-loop interval (state unchanged, 16
// bytes skipped -- and, in a state that loops on the hot interval, every following chunk whose bit is set in
// `bitmap` as well), or takes 16 exact steps (the partial last window of a line goes through the identity column
// or, when it passes the range test dword by dword, is skipped too).  Returns the row of the final state.
template <int TIER, bool CAPTURE, bool SIMPLE>
__device__ __forceinline__ uint32_t walk(const WalkTab& W, uint32_t stage, uint32_t bitmap, bool use_map, uint32_t row, uint32_t start,
                                         uint32_t end, bool on, uint32_t dead_row, uint32_t regs) {
    uint32_t wb = start;  // windows are relative to the line, not to the staging area: see lds_window
    const uint32_t len = end - start;
    const uint32_t full_lim = len >= 16u ? len - 15u : 0u;  // window at line offset rel is full iff rel < full_lim
    uint32_t acc = state_acc<TIER>(W, row);  // self-loop interval: lo | (0x7F - hi) << 8 | hot << 16
    uint32_t lo4 = splat_byte0(acc), k4 = splat_byte1(acc);
    bool more = on && start < end;
    while (__any(more)) {
        // a finished lane keeps its last window: the address stays inside the staged tile
        const uint4 w0 = lds_window_any(stage + wb);
        const uint32_t rel = wb - start;
        const bool full0 = rel < full_lim;
        bool ok0 = false;
        bool step = more;
        uint32_t mask = 0xFFFFu;
        if (__any(more && k4 != HI_BITS)) {  // (no lane in a state with a self-loop interval: straight to the steps)
            const uint32_t bx = outside_bits(w0.x, lo4, k4), by = outside_bits(w0.y, lo4, k4);
            const uint32_t bz = outside_bits(w0.z, lo4, k4), bw = outside_bits(w0.w, lo4, k4);
            ok0 = full0 & (((or3(bx, by, bz) | bw) & HI_BITS) == 0u);
            step = more && !ok0;
            if (step && !full0) {
                mask = window_mask(start, end, wb);
                step = !partial_window_ok(bx, by, bz, bw, mask);
            }
        } else if (!full0) mask = window_mask(start, end, wb);
        uint32_t nwb = wb + 16u;
        // The window just tested is verified and the state loops on the whole hot interval: the chunk that begins
        // inside the window, and every chunk after it, is skipped as long as its bit says "all bytes hot".
        const bool jump = use_map && more && ok0 && (acc & 0x10000u) != 0u;
        if (__any(jump)) {
            const uint32_t c0 = (wb + 15u) >> 4;
            const uint32_t wa = bitmap + ((c0 >> 5) << 2);
            const uint32_t m_lo = lds_ld<uint32_t>(wa), m_hi = lds_ld<uint32_t>(wa + 4u);
            const uint32_t bits = __builtin_amdgcn_alignbit(m_hi, m_lo, c0 & 31u);  // bits of chunks c0 .. c0 + 31
            const uint32_t run = min(static_cast<uint32_t>(__builtin_ffs(static_cast<int>(~bits)) - 1), 32u);
            if (jump && run) nwb = min((c0 + run) << 4, end);
        }
        // one variant per wave: if any stepping lane has a partial window, every stepping lane takes the masked
        // steps (its mask is all ones) instead of the wave running both variants one after the other
        const bool masked = __any(step && !full0);
        if (step) {
            if (!masked) row = steps16<TIER, CAPTURE, false, SIMPLE>(w0, mask, W, row, rel, regs);
            else row = steps16<TIER, CAPTURE, true, SIMPLE>(w0, mask, W, row, rel, regs);
            acc = state_acc<TIER>(W, row);
            lo4 = splat_byte0(acc);
            k4 = splat_byte1(acc);
        }
        if (more) wb = nwb;
        more = more && wb < end && row != dead_row;
    }
    return row;
}

// Staging copy for a span that touches the first or last bytes of the buffer: never reads outside [data, data_end).
template <bool WIDE>  // WIDE: the buffer holds 16-bit code units; their low bytes are staged (16 of them per chunk, out of 32 bytes)
__device__ __attribute__((unused)) void stage_span_guarded(const uint8_t* __restrict__ g_al, uint32_t nch, uint32_t stage, uint32_t lane,
                                   const uint8_t* data, const uint8_t* data_end) {
    for (uint32_t c = lane; c < nch; c += 64) {
        const uint8_t* src = g_al + (static_cast<uint64_t>(c) << (WIDE ? 5 : 4));
        uint32_t w[4] = {0, 0, 0, 0};
        for (int q = 0; q < 16; ++q) {
            const uint8_t* at = src + (WIDE ? 2 * q : q);
            if (at >= data && at < data_end) w[q >> 2] |= static_cast<uint32_t>(*at) << ((q & 3) * 8);
        }
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        lds_st<u32x4>(stage + (c << 4), v);
    }
}

// What one wave needs to know about a round: the lines of a 64-line group (lane = line) that are staged and
// walked together.  Normally one round covers the whole group; when the group's bytes do not fit the staging
// area (lines longer than the hint, mixed lengths) the group is taken in several rounds of consecutive lanes.
// (32-bit flags and no padding: a struct with padding bytes is copied through scratch memory.)
struct TileInfo {
    uint64_t i;            // this lane's line index
    uint64_t o0, o1;       // its byte range in the CSR buffer
    const uint8_t* g_al;   // 16-byte aligned start of the round's span in global memory
    uint32_t nch;          // 16-byte chunks in the span
    uint32_t start, end;   // this lane's line inside the staging area (0, 0 for a lane outside the round)
    uint32_t mode;         // 0: prefetched into registers; 1: touches the buffer edge (guarded copy);
                           // 2: one line that does not fit the staging area (left to the per-line kernel); 3: nothing
    uint32_t active;       // this lane's line belongs to the round
    uint32_t a, b;         // the round covers lanes [a, b) of the group (wave-uniform)
    uint32_t pad_;
};

// Clamped, unconditional accesses on both sides: a lane beyond the span re-reads / rewrites the last chunk
// with identical data.  (Per-lane conditions make the compiler spill the array and serialise the batch.)
// Unconditional as well: a round that is not prefetched (mode != 0, or no next round) loads one dummy chunk instead.
// A branch around the loads would make the compiler lose count of them at the join and wait for all of them --
// i.e. for the whole prefetch -- at the next vector-memory dependency, long before the walk.
// WIDE: a staged chunk is the low bytes of 16 code units = 32 bytes of the buffer; the lane loads both halves (two instructions
// whose lanes sit 32 bytes apart: together they sweep the span once).
template <int KCH, bool WIDE>
__device__ __forceinline__ void tile_issue_loads(const TileInfo& t, uint32_t lane, u32x4 (&pre)[WIDE ? 2 * KCH : KCH], const uint8_t* __restrict__ dummy) {
    // wave-uniform: scalar base + 32-bit lane offset (readfirstlane keeps the selects on the scalar unit; folded into
    // the per-lane offsets they cost two vector instructions per load)
    const uint64_t src_u = reinterpret_cast<uint64_t>(t.mode == 0 ? t.g_al : dummy);
    const uint32_t src_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(src_u));  // (the builtin returns int: no sign extension)
    const uint32_t src_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(src_u >> 32));
    const uint64_t src = (static_cast<uint64_t>(src_hi) << 32) | src_lo;
    const uint32_t last16 = __builtin_amdgcn_readfirstlane(t.mode == 0 ? (t.nch - 1u) << (WIDE ? 5 : 4) : 0u);
    const uint32_t lane16 = lane << (WIDE ? 5 : 4);
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const uint32_t o = min(lane16 + (WIDE ? 2048u : 1024u) * k, last16);
        if (WIDE) {
            pre[2 * k] = __builtin_nontemporal_load((__attribute__((address_space(1))) const u32x4*)(src + o));
            // (the dummy chunk of a round that is not prefetched is 16 bytes of the table image: its second half is the same again)
            pre[2 * k + 1] = __builtin_nontemporal_load((__attribute__((address_space(1))) const u32x4*)(src + o + (t.mode == 0 ? 16u : 0u)));
        } else pre[k] = __builtin_nontemporal_load((__attribute__((address_space(1))) const u32x4*)(src + o));  // a global_load, not a flat one
    }
}
// the low bytes of 16 code units
__device__ __forceinline__ u32x4 narrow16(const u32x4& a, const u32x4& b) {
    return u32x4{__builtin_amdgcn_perm(a.y, a.x, 0x06040200u), __builtin_amdgcn_perm(a.w, a.z, 0x06040200u),
                 __builtin_amdgcn_perm(b.y, b.x, 0x06040200u), __builtin_amdgcn_perm(b.w, b.z, 0x06040200u)};
}
// Registers -> staging area, and the hot-interval bit of every chunk -> the wave's bitmap (bit 64 k + lane of the
// map belongs to the chunk lane `lane` holds in pre[k]; a clamped lane describes a chunk beyond the span, which no
// line of the round reaches).
// WIDE: returns true for a lane whose line holds a code unit above 0xFF (its staged low bytes are not the line).
template <int KCH, int MAP, bool WIDE>  // MAP 0: no bitmap; 1: general hot interval; 2: hot interval ends at 0x7F; 3: no bitmap (TIER_HOP)
__device__ __forceinline__ bool commit_chunks(const TileInfo& t, uint32_t lane, const u32x4 (&pre)[WIDE ? 2 * KCH : KCH], uint32_t stage, uint32_t bitmap,
                                              uint32_t hot_lo4, uint32_t hot_k4) {
    const uint32_t last = stage + ((t.nch - 1u) << 4), mine = stage + (lane << 4);
    bool wide_line = false;
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const u32x4 v = WIDE ? narrow16(pre[2 * k], pre[2 * k + 1]) : pre[k];
        lds_st<u32x4>(min(mine + 1024u * k, last), v);
        if (MAP == 1 || MAP == 2) {
            const unsigned long long m = __ballot(chunk_inside<MAP == 2>(v, hot_lo4, hot_k4));
            if (lane == 0) lds_st<u32x2>(bitmap + 8u * k, u32x2{static_cast<uint32_t>(m), static_cast<uint32_t>(m >> 32)});
        }
        if (WIDE) {
            const u32x4 &a = pre[2 * k], &b = pre[2 * k + 1];
            const uint32_t high = (or3(a.x, a.y, a.z) | or3(a.w, b.x, b.y) | b.z | b.w) & 0xFF00FF00u;
            unsigned long long wm = __builtin_amdgcn_ballot_w64(high != 0u);
            while (wm != 0ull) {   // (rare: log text is Latin-1) the lines that reach into a chunk with such a unit
                const uint32_t src = static_cast<uint32_t>(__builtin_ctzll(wm));
                wm &= wm - 1ull;
                const uint32_t c16 = min((src << 4) + 1024u * k, (t.nch - 1u) << 4);   // the chunk's first staged byte (a clamped lane's: the last chunk again)
                if (t.active && t.start < c16 + 16u && t.end > c16) wide_line = true;
            }
        }
    }
    return wide_line;
}
template <int KCH, bool HOP, bool WIDE>
__device__ __forceinline__ bool tile_commit(const TileInfo& t, uint32_t lane, const u32x4 (&pre)[WIDE ? 2 * KCH : KCH], uint32_t stage, uint32_t bitmap,
                                            bool use_map, uint32_t hot_lo4, uint32_t hot_k4, const uint8_t* data, const uint8_t* data_end) {
    bool wide_line = false;
    if (t.mode == 0) {
        if (HOP) wide_line = commit_chunks<KCH, 3, WIDE>(t, lane, pre, stage, bitmap, hot_lo4, hot_k4);
        else if (!use_map) wide_line = commit_chunks<KCH, 0, WIDE>(t, lane, pre, stage, bitmap, hot_lo4, hot_k4);
        else if (hot_k4 == 0u) wide_line = commit_chunks<KCH, 2, WIDE>(t, lane, pre, stage, bitmap, hot_lo4, hot_k4);
        else wide_line = commit_chunks<KCH, 1, WIDE>(t, lane, pre, stage, bitmap, hot_lo4, hot_k4);
    } else if (t.mode == 1) {
        stage_span_guarded<WIDE>(t.g_al, t.nch, stage, lane, data, data_end);
        if (use_map && lane < 2u * KCH) lds_st<uint32_t>(bitmap + 4u * lane, 0u);  // no chunk of a guarded round is skipped
        if (WIDE && t.active) {   // (the first and the last round of a buffer: every lane looks at its own line's units)
            const uint16_t* u = reinterpret_cast<const uint16_t*>(data) + t.o0;
            for (uint64_t q = 0; q < t.o1 - t.o0; ++q) wide_line = wide_line || u[q] > 0xFFu;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return wide_line;
}

#ifdef GX_DEV
#define GX_STAMP(slot)                                                        \
    do {                                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
        phase_cycles[slot] += now_ - stamp_;                                  \
        stamp_ = now_;                                                        \
    } while (0)
#else
#define GX_STAMP(slot) do { } while (0)
#endif

// MODE 0: PolyMatcher.match alone (match automaton, first accepting extraction).
// MODE 1: the fused automaton with branch-free capture steps (every program is "one register := position").
// MODE 2: everything else: fused automaton with general programs, or match automaton then the winning
//         extraction's capture automaton (definitions too large to fuse).
// KCH: 16-byte chunks per lane that cover the staging area (stage_bytes <= KCH * 1024).  A tile's span is
// fetched into KCH*4 VGPRs per lane one tile AHEAD: the loads are issued before the current tile is walked
// and land while the wave computes out of LDS, so HBM latency is hidden without a second LDS buffer.
// PACKED: 1 = results leave as compact u16 rows, 2 = as u8 rows (gx_batch_opts.compact_results) -- kernels of their own, so that profiles tell
// the two result formats apart.
// WIDE: `data` holds UTF-16 code units (gx_batch_opts.utf16), offsets count units; the units' low bytes are staged (26 prefetch
// registers per 13 staged chunks: two waves per SIMD) and the lines that hold a unit above 0xFF are flagged for the per-line walk.
template <typename OFF, int KCH, int TIER, int MODE, int PACKED, bool WIDE = false>
__global__ void __launch_bounds__((KCH > 13 || WIDE) ? 512 : 768) __attribute__((amdgpu_waves_per_eu(1, (KCH > 13 || WIDE) ? 2 : 3)))
k_extract_tile(GxLds L, TileIO io) {
    // ---- prologue: table image -> LDS (the only access through the __shared__ symbol; its address is 0) ----
    {
        extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];
        const uint4* src = reinterpret_cast<const uint4*>(io.image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
        // the tile counters of the NEXT launch on this stream (the other row): zero when it begins
        uint32_t* other = io.steal + (io.steal_parity ^ 1u) * (GX_STEAL_MAX * GX_STEAL_STRIDE);
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < GX_STEAL_MAX; q += gridDim.x * blockDim.x) other[q * GX_STEAL_STRIDE] = 0u;
        if (threadIdx.x == 0) lds_st<uint32_t>(L.counter, 0u);   // the workgroup's tile counter (below)
    }
    __syncthreads();

    constexpr bool GT = TIER == TIER_L2 || TIER == TIER_RECG || TIER == TIER_HOP;  // automaton tables and final records in global memory
    constexpr bool HOP = TIER == TIER_HOP;
    HopTab H;
    H.rows = io.at_global;
    H.hops = io.at_global + L.c_base;
    H.row_bytes = L.row_bytes;
    H.info_off = L.ncls * 4u;
    H.n_hot = L.rec_indexed;
    H.lrow_cols = (2u * L.ncls + 3u) & ~3u;
    H.sets = L.hop_sets;
    const uint8_t* __restrict__ data = io.data;
    const OFF* __restrict__ off = static_cast<const OFF*>(io.off);
    const uint64_t n = io.n;
    // match automaton / fused (or per-extraction capture) automaton
    WalkTab Wm, Wc;
    Wm.at = GT ? io.at_global + (TIER == TIER_RECG ? L.c_base : 0u) : nullptr;  // (records: one index space per image)
    Wm.row_bytes = L.row_bytes;
    Wm.ops_off = L.ops_off;
    Wm.ops = L.ops;
    Wm.acc_tab = L.acc_tab;
    Wm.ncls = L.ncls;
    Wm.indexed = L.rec_indexed;
    Wc = Wm;
    if (GT) Wc.at = io.at_global + L.c_base;  // (two-pass layout: set per lane below)

    const uint32_t lane = threadIdx.x & 63u;
    // (the wave index through readfirstlane: everything derived from it -- tile numbers, LDS areas, the round's span --
    // is then wave-uniform for the compiler too, i.e. scalar registers and scalar arithmetic)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stage = L.stage + wave * L.stage_bytes;
    const uint32_t bitmap = L.bitmap + wave * GX_BITMAP_WAVE_BYTES;
    const bool use_map = !HOP && L.hot_k4 != HI_BITS;
    // register r of this lane = u16 at regs + r * 128; the column before register 0 is a write-only dummy
    const uint32_t regs = L.regs + wave * L.regs_wave_bytes + 128u + lane * 2u;

    const int G = io.max_groups;
    const uint32_t slots = 2u * static_cast<uint32_t>(G);
    const uint64_t tiles = (n + 63) >> 6;
    const uint8_t* data_end = data + (static_cast<uint64_t>(off[n]) << (WIDE ? 1 : 0));

    auto load_offsets = [&](uint64_t tile, uint64_t& o0, uint64_t& o1) {
        const uint64_t i = (tile << 6) + lane;
        const bool valid = i < n;
        o0 = off[valid ? i : n];
        o1 = off[valid ? i + 1 : n];
    };
    // the value lane `src` (wave-uniform) holds, as a wave-uniform value
    auto lane_value = [&](uint64_t v, uint32_t src) -> uint64_t {
        const uint32_t l = __builtin_amdgcn_readfirstlane(src);
        const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), l);
        const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), l);
        return (static_cast<uint64_t>(hi) << 32) | lo;
    };
    // Round of group `tile` starting at lane a: as many consecutive lines as fit the staging area.
    auto make_round = [&](uint64_t tile, uint32_t a, uint64_t o0, uint64_t o1) {
        TileInfo t;
        t.i = (tile << 6) + lane;
        const bool valid = t.i < n;
        t.pad_ = 0;
        t.o0 = o0; t.o1 = o1;
        t.a = a;
        const uint64_t lo = lane_value(o0, a);  // lane a holds a valid line
        const uint8_t* g_lo = data + (lo << (WIDE ? 1 : 0));
        const uint32_t skew_b = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(g_lo) & (WIDE ? 31u : 15u));   // (WIDE: a staged chunk is 32 bytes of the buffer)
        const uint32_t skew = skew_b >> (WIDE ? 1 : 0);   // ... in staged bytes
        t.g_al = g_lo - skew_b;  // 16-byte aligned; still a global-address-space pointer for the compiler
        // offsets ascend, so the lines that fit are a run of lanes starting at a (+48: the walk looks ahead of the line)
        const bool fits = lane >= a && valid && (o1 - lo) + skew + 48u <= L.stage_bytes;
        const uint32_t cnt = static_cast<uint32_t>(__popcll(__ballot(fits)));
        if (cnt == 0) {  // line a alone is longer than the staging area
            t.b = a + 1u;
            t.mode = 2;
            t.nch = 1;
        } else {
            t.b = a + cnt;
            const uint64_t hi = lane_value(o1, t.b - 1u);
            const uint64_t span = (hi - lo) + skew;
            t.nch = max(static_cast<uint32_t>((span + 15) >> 4), 1u);
            t.mode = (t.g_al >= data && t.g_al + (static_cast<uint64_t>(t.nch) << (WIDE ? 5 : 4)) <= data_end) ? 0u : 1u;
        }
        t.active = (lane >= a && lane < t.b) ? 1u : 0u;
        t.start = t.active ? skew + static_cast<uint32_t>(o0 - lo) : 0u;
        t.end = t.active ? skew + static_cast<uint32_t>(o1 - lo) : 0u;
        return t;
    };

    // Tiles are handed out on demand.  Workgroup b owns tiles b, b + grid, b + 2 grid, ... (so the tiles being read at any
    // moment are one dense window of the buffer), and a wave that has finished a tile takes the workgroup's next one from a
    // counter.  A static split would leave every wave the same number of tiles, but the waves of a workgroup do not run at
    // the same speed -- 11 waves sit 3, 3, 3 and 2 to a SIMD -- and the kernel would end with the slowest.  Nor do the
    // workgroups: the XCDs differ (on some boxes of this pool the odd ones finish an equal share 3 % later with u8 result
    // rows, 10 % later with dense results: profiles/r04_tile_tail.txt).  So a workgroup's tiles come in two parts: the
    // first (1 - share) from a counter in LDS, as cheap as a draw gets; the last `share` from a counter in GLOBAL memory,
    // one cache line per workgroup, which the waves of OTHER workgroups draw from as well once their own tiles are gone --
    // they pick a workgroup that has tiles left at random.  (All tiles from global counters: 8 % slower -- a returning
    // atomic per tile and wave in a kernel that saturates the memory system; counters that share cache lines: 29 %, atomics
    // on one line take their turns at 11 ns apiece; a single counter for the grid: 1.8 ms per 10 M lines.
    // tools/micro/atomic_rate.hip, tools/ab_flags.sh.)  Each wave looks three tiles ahead (walking / bytes in flight /
    // offsets in flight) and draws a whole walk before it needs the answer, so no draw is waited for until the wave steals,
    // at the end of the launch.  Exit: counters only grow; a wave leaves when no workgroup has a tile left.
    const uint32_t grid = gridDim.x;
    const uint32_t j_base = 2u * L.nwaves;   // (the first two tiles of every wave are fixed: draw j is the workgroup's tile j_base + j)
    uint32_t* const ctr = io.steal + io.steal_parity * (GX_STEAL_MAX * GX_STEAL_STRIDE);
    const uint32_t tiles32 = static_cast<uint32_t>(tiles);   // (n < 2^32 lines: at most 2^26 tiles)
#ifndef GX_TILE_CHUNK
#define GX_TILE_CHUNK 1u   // consecutive tiles a workgroup takes before the next workgroup's begin (1: tiles b, b + grid, ...)
#endif
    constexpr uint32_t CH = GX_TILE_CHUNK;
    // workgroup wg's k-th tile: chunks of CH consecutive tiles, dealt round robin
    auto own_of = [&](uint32_t v) -> uint32_t {
        if (CH == 1u) return v < tiles32 ? (tiles32 - v + grid - 1u) / grid : 0u;   // tiles v, v + grid, ...
        const uint32_t round = grid * CH, full = tiles32 / round, rem = tiles32 % round;
        return full * CH + (rem > v * CH ? min(rem - v * CH, CH) : 0u);
    };
    auto tile_at = [&](uint32_t wg, uint32_t k) -> uint64_t {
        if (CH == 1u) return min(static_cast<uint64_t>(wg) + static_cast<uint64_t>(k) * grid, tiles);
        return min((static_cast<uint64_t>(k / CH) * grid + wg) * CH + k % CH, tiles);
    };
    // workgroup v's draws: `local` from its LDS counter, then `shared` from its global one
    auto shared_of = [&](uint32_t v, uint32_t& local) -> uint32_t {
        const uint32_t own = own_of(v);
        const uint32_t avail = own > j_base ? own - j_base : 0u;
        const uint32_t shared = (avail * io.share64 + 63u) >> 6;
        local = avail - shared;
        return shared;
    };
    uint32_t local_own;
    const uint32_t shared_own = shared_of(blockIdx.x, local_own);
    // The next draw of this workgroup, in two steps: the LDS ticket (inc 0: none wanted -- the instruction is issued all the same;
    // lane 0's value counts) at the top of a round, and -- behind the round's prefetch, when the ticket has long answered -- for a
    // ticket beyond the local part a draw of the global counter, lane 0's `g`.  (Both in one place behind the prefetch: the wave
    // waited out an LDS round trip per tile there, 1 % of the kernel on boxes whose XCDs are even.)
    auto draw_ticket = [&](uint32_t inc) -> uint32_t {
        uint32_t j = 0;
        if (lane == 0) j = __hip_atomic_fetch_add((GX_LDS uint32_t*)(uintptr_t)L.counter, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return j;
    };
    auto draw_shared = [&](uint32_t j, uint32_t inc, uint32_t& g) -> bool {
        const bool from_shared = inc != 0u && io.share64 != 0u && __builtin_amdgcn_readfirstlane(j) >= local_own;
        g = 0;
        if (from_shared) {   // (wave-uniform; the last `share` of the launch only)
            if (lane == 0) g = __hip_atomic_fetch_add(ctr + blockIdx.x * GX_STEAL_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return from_shared;
    };
    // a tile from another workgroup's shared part, or `tiles` when nobody has one left (wave-uniform; waits for its loads)
    uint32_t steal_salt = (blockIdx.x * L.nwaves + wave) * 2654435761u;
    auto steal = [&]() -> uint64_t {
        if (io.share64 == 0u) return tiles;   // (a launch that shares nothing: gx_tile.hip)
        for (;;) {
            // workgroups with shared draws left, and how many there are
            uint32_t cands = 0;
            for (uint32_t v0 = 0; v0 < grid; v0 += 64u) {
                const uint32_t v = v0 + lane;
                bool left = false;
                if (v < grid) {
                    uint32_t lv;
                    left = __hip_atomic_load(ctr + v * GX_STEAL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < shared_of(v, lv);
                }
                cands += static_cast<uint32_t>(__popcll(__ballot(left)));
            }
            if (cands == 0u) return tiles;
            steal_salt = steal_salt * 1664525u + 1013904223u;
            uint32_t pick = (steal_salt >> 8) % cands, victim = 0xFFFFFFFFu, victim_local = 0;
            for (uint32_t v0 = 0; v0 < grid && victim == 0xFFFFFFFFu; v0 += 64u) {   // (the counters again: they are close by now)
                const uint32_t v = v0 + lane;
                bool left = false;
                uint32_t lv = 0;
                if (v < grid) left = __hip_atomic_load(ctr + v * GX_STEAL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < shared_of(v, lv);
                const unsigned long long m = __ballot(left);
                const uint32_t c = static_cast<uint32_t>(__popcll(m));
                if (pick < c) {
                    unsigned long long mm = m;
                    for (uint32_t q = 0; q < pick; ++q) mm &= mm - 1ull;
                    const uint32_t src = static_cast<uint32_t>(__ffsll(static_cast<long long>(mm)) - 1);
                    victim = v0 + src;
                    victim_local = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lv), static_cast<int>(src)));
                } else pick -= c;
            }
            if (victim == 0xFFFFFFFFu) continue;   // (taken in between: look again)
            uint32_t g = 0;
            if (lane == 0) g = __hip_atomic_fetch_add(ctr + victim * GX_STEAL_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = __builtin_amdgcn_readfirstlane(g);
            const uint64_t t = tile_at(victim, j_base + victim_local + g);   // (beyond the victim's tiles: `tiles`)
            if (t < tiles) return t;
        }
    };
    uint64_t tile = tile_at(blockIdx.x, wave);
    uint64_t t1 = tile_at(blockIdx.x, L.nwaves + wave);   // the group after `tile`: its offsets are loaded one iteration ahead
    uint64_t t2 = tiles;                        // and the one after that: from the draw (j3, g3), once that has answered (need_t2)
    uint32_t g3 = 0, jn = 0u, gn = 0u;
    bool sh3 = false, shn = false;
    uint32_t j3 = draw_ticket(1u);
    sh3 = draw_shared(j3, 1u, g3);
    bool need_t2 = true, stealing = false, drew_before = false;   // (wave-uniform)
    if (tile >= tiles) return;
#ifdef GX_DEV
    unsigned long long phase_cycles[4] = {0, 0, 0, 0};
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    const unsigned long long dev_begin = __builtin_amdgcn_s_memrealtime();   // (100 MHz, one clock for the whole chip)
    unsigned long long dev_tiles = 0;
#endif
    bool hop_second = false;   // hop tier: this tile's walk carries the second chances (the tile before it met text that wants them)
    u32x4 pre[WIDE ? 2 * KCH : KCH];  // the next round's bytes, in flight or landed
    uint64_t no0 = 0, no1 = 0;
    TileInfo cur;
    {
        uint64_t o0, o1;
        load_offsets(tile, o0, o1);
        cur = make_round(tile, 0, o0, o1);
    }
    load_offsets(min(t1, tiles - 1), no0, no1);
    tile_issue_loads<KCH, WIDE>(cur, lane, pre, io.image);

    for (;;) {
        const bool wide_line = tile_commit<KCH, HOP, WIDE>(cur, lane, pre, stage, bitmap, use_map, L.hot_lo4, L.hot_k4, data, data_end);
        GX_STAMP(0);
        // ---- software pipeline: start fetching the next round (and the offsets of the group after it) ----
        const uint32_t group_lines = static_cast<uint32_t>(min(static_cast<uint64_t>(64), n - (tile << 6)));
        const bool same_group = cur.b < group_lines;
        const uint64_t ntile = same_group ? tile : t1;
        const bool has_next = ntile < tiles;
        if (drew_before) { j3 = jn; g3 = gn; sh3 = shn; }   // (the draw of the round before: it went out behind that round's prefetch, which has landed)
        if (need_t2) {   // (wave-uniform) the draw that was issued a whole walk ago
            if (!stealing) {
                if (!sh3) t2 = tile_at(blockIdx.x, j_base + __builtin_amdgcn_readfirstlane(j3));
                else {
                    const uint32_t g = __builtin_amdgcn_readfirstlane(g3);
                    t2 = g < shared_own ? tile_at(blockIdx.x, j_base + local_own + g) : tiles;
                }
                if (t2 >= tiles) stealing = true;   // this workgroup's tiles are all taken: from now on, other workgroups'
            }
            if (stealing) t2 = steal();
        }
        const bool drew = need_t2 && !stealing;     // ... and the next one: its ticket now, its global draw (if any) behind the prefetch, below
        need_t2 = false;
        jn = draw_ticket(drew ? 1u : 0u);
        const uint64_t after = same_group ? t1 : t2;  // the group after `ntile`
        // Offsets of the group after the next one first, then the prefetch, and nothing in between that depends on
        // vector memory: both are unconditional (index clamped -- the same values again while the group is
        // unchanged; a dummy chunk when there is no next round), because a conditional load needs a register copy
        // at the join, and that copy would wait for every load issued before it.
        uint64_t nno0, nno1;
        load_offsets(min(after, tiles - 1), nno0, nno1);
        TileInfo nxt = make_round(has_next ? ntile : tile, has_next ? (same_group ? cur.b : 0u) : cur.a,
                                  has_next && !same_group ? no0 : cur.o0, has_next && !same_group ? no1 : cur.o1);
        if (!has_next) nxt.mode = 3;  // nothing to fetch: the loop ends after this round
        tile_issue_loads<KCH, WIDE>(nxt, lane, pre, io.image);
        shn = draw_shared(jn, drew ? 1u : 0u, gn);
        drew_before = drew;
        GX_STAMP(1);

        const uint64_t i = cur.i;
        const bool valid = cur.active != 0u;
        const uint32_t start = cur.start;
        uint32_t end = cur.end;
        if (io.strip_eol && cur.mode != 2) {  // the terminator is staged with the line (trim_eol, from LDS)
            if (end > start && lds_ld<uint8_t>(stage + end - 1u) == 0x0Au) --end;
            if (end > start && lds_ld<uint8_t>(stage + end - 1u) == 0x0Du) --end;
        }
        if (cur.mode == 2) {
            // one line that does not fit the staging area: the per-line kernel takes it in a follow-up launch
            if (lane == cur.a) __hip_atomic_store(io.oversize_flag, io.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (MODE == 0 && HOP) {
            // ---- hot loop #1 alone on the match automaton's hop records: a state's info word is its first accepting extraction ----
            const uint32_t mrow = walk_hop<false>(H, L.rec_indexed >= L.sort_chunk, stage, L.m_start, start, end, true, L.m_dead, regs, hop_second);
            int32_t first = static_cast<int16_t>(lds_ld<uint16_t>(L.acc_tab + 2u * min(mrow, H.n_hot - 1u)));
            if (wave_any(mrow >= H.n_hot)) {
                if (mrow >= H.n_hot) first = *reinterpret_cast<const int32_t*>(H.rows + (static_cast<uint64_t>(mrow) * H.row_bytes + H.info_off));
            }
            if (valid) io.match_id[i] = first;
            GX_STAMP(2);
        } else if (MODE == 0) {
            // ---- hot loop #1 alone: PolyMatcher.match ----
            const uint32_t mrow = walk<TIER, false, false>(Wm, stage, bitmap, use_map, L.m_start, start, end, true, L.m_dead, regs);
            if (valid) io.match_id[i] = state_info<TIER>(Wm, mrow);
            if ((TIER == TIER_LDS || TIER == TIER_L2) && io.state_out && valid)   // (a row IS a state of the match automaton: gx_state_accepts reads the rest)
                io.state_out[i] = mrow == L.m_dead ? -1 : static_cast<int32_t>(TIER == TIER_LDS ? (mrow - L.m_start) / L.row_bytes : mrow - L.m_start);
            GX_STAMP(2);
        } else {
            int32_t info;  // of the state the line's walk ended in: -1 null, -2-k ExtractionException, else its final record
            if (HOP) {
                // ---- fused pass on the hop records: a run and a chain per iteration (gx_hop_dev.hpp) ----
                const uint32_t urow = walk_hop<true>(H, L.rec_indexed >= L.sort_chunk, stage, L.u_start, start, end, true, L.u_dead, regs, hop_second);
                // the final state's info word: int16 in LDS for the hot states (offset / 16, or -1 / -2-k), else its dense row's last column
                const int32_t hot_info = static_cast<int16_t>(lds_ld<uint16_t>(L.acc_tab + 2u * min(urow, H.n_hot - 1u)));
                info = hot_info >= 0 ? hot_info * 16 : hot_info;
                if (wave_any(urow >= H.n_hot)) {
                    if (urow >= H.n_hot) info = *reinterpret_cast<const int32_t*>(H.rows + (static_cast<uint64_t>(urow) * H.row_bytes + H.info_off));
                }
            } else if (MODE == 1 || L.u_start != 0xFFFFFFFFu) {
                // ---- fused pass: match automaton x joined capture automata, one walk ----
                uint32_t urow;
                if (MODE == 1 || L.simple_ops) urow = walk<TIER, true, true>(Wc, stage, bitmap, use_map, L.u_start, start, end, true, L.u_dead, regs);
                else urow = walk<TIER, true, false>(Wc, stage, bitmap, use_map, L.u_start, start, end, true, L.u_dead, regs);
                info = state_info<TIER>(Wc, urow);
            } else {
                // ---- hot loop #1, then hot loop #2 on extraction k's tagged automaton ----
                const uint32_t mrow = walk<TIER, false, false>(Wm, stage, bitmap, use_map, L.m_start, start, end, true, L.m_dead, regs);
                const int32_t k = state_info<TIER>(Wm, mrow);
                info = k;
                uint32_t crow = GT ? 0u : L.m_dead;  // any valid row: the walk below is off for lanes without a match
                if (k >= 0) crow = lds_ld<uint32_t>(L.c_rule + 8u * k);
                if (L.simple_ops) crow = walk<TIER, true, true>(Wc, stage, bitmap, use_map, crow, start, end, k >= 0, 0xFFFFFFFFu, regs);
                else crow = walk<TIER, true, false>(Wc, stage, bitmap, use_map, crow, start, end, k >= 0, 0xFFFFFFFFu, regs);
                if (k >= 0) {
                    info = state_info<TIER>(Wc, crow);
                    if (info < 0) info = -2 - k;  // DFA said yes, capture regex says no -> ExtractionException
                }
            }
            GX_STAMP(2);
            const uint32_t len = end - start;
            if (HOP) lds_st<uint16_t>(regs - 128u, static_cast<uint16_t>(len));   // (the dummy column, free now: the tag "the line's length" names it -- gx_hop.cpp)
            const uint8_t* fin_g = GT && !(HOP && L.at != 0u) ? io.at_global + L.fin_tags : nullptr;
            const uint32_t fin_lds = HOP ? L.at : L.fin_tags;
            // The tile's 64 result rows are one contiguous block of the output.  Transpose them through the staging
            // area (free now: every lane has finished its walk) so that each store instruction writes 1 KiB of
            // consecutive bytes, instead of every lane writing pieces of its own row.
            const bool full_tile = cur.a == 0u && cur.b == 64u;
            if (PACKED) {
                // u16 rows (int16 id + u16 offsets), or -- io.narrow, wave-uniform -- u8 rows (int8 id + u8 offsets, an offset
                // above 254 stored as 254 and counted).  A staged line is shorter than 65 535 bytes, so u16 rows never clamp here.
                constexpr bool narrow = PACKED == 2;
                const uint32_t row_b = narrow ? 1u + slots : 2u + 2u * slots;
                uint8_t* out_rows = reinterpret_cast<uint8_t*>(io.packed) + (cur.i - lane) * static_cast<uint64_t>(row_b);
                const bool rows_aligned = (reinterpret_cast<uintptr_t>(io.packed) & 15u) == 0u;  // (64 rows are a multiple of 16 bytes)
                uint32_t clamped = 0;
                if (full_tile && rows_aligned && 64u * row_b + 16u <= L.stage_bytes) {
                    const uint32_t my_row = stage + lane * row_b;
                    int32_t result;
                    if (narrow && !__any(len > 254u)) {
                        // (no offset of this tile's lines is above 254: nothing to clip or to count)
                        result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            lds_st<uint8_t>(my_row + 1u + 2u * g, static_cast<uint8_t>(pb));
                            lds_st<uint8_t>(my_row + 2u + 2u * g, static_cast<uint8_t>(pe));
                        }, L.fin_unset);
                        lds_st<uint8_t>(my_row, static_cast<uint8_t>(result));
                    } else if (narrow) {
                        result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            clamped += (pb > 254 ? 1u : 0u) + (pe > 254 ? 1u : 0u);
                            lds_st<uint8_t>(my_row + 1u + 2u * g, static_cast<uint8_t>(pb > 254 ? 254 : pb));
                            lds_st<uint8_t>(my_row + 2u + 2u * g, static_cast<uint8_t>(pe > 254 ? 254 : pe));
                        }, L.fin_unset);
                        lds_st<uint8_t>(my_row, static_cast<uint8_t>(result));
                    } else {
                        result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            lds_st<uint16_t>(my_row + 2u + 4u * g, static_cast<uint16_t>(pb));
                            lds_st<uint16_t>(my_row + 4u + 4u * g, static_cast<uint16_t>(pe));
                        }, L.fin_unset);
                        lds_st<uint16_t>(my_row, static_cast<uint16_t>(result));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t c = lane; c < 4u * row_b; c += 64u) {  // 64 * row_b / 16 chunks
#ifdef GX_DEV
                        if (io.dev_flags & 2u) { __builtin_nontemporal_store(lds_ld<u32x4>(stage + (c << 4)), reinterpret_cast<u32x4*>(out_rows + (c << 4))); continue; }
                        if ((io.dev_flags & 4u) && c != 0u) continue;  // experiment: (almost) no result stores
#endif
                        *reinterpret_cast<u32x4*>(out_rows + (c << 4)) = lds_ld<u32x4>(stage + (c << 4));
                    }
                } else if (valid) {
                    if (narrow) {
                        uint8_t* rp = reinterpret_cast<uint8_t*>(io.packed) + i * static_cast<uint64_t>(row_b);
                        const int32_t result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            clamped += (pb > 254 ? 1u : 0u) + (pe > 254 ? 1u : 0u);
                            rp[1 + 2 * g] = static_cast<uint8_t>(pb > 254 ? 254 : pb);
                            rp[2 + 2 * g] = static_cast<uint8_t>(pe > 254 ? 254 : pe);
                        }, L.fin_unset);
                        rp[0] = static_cast<uint8_t>(result);
                    } else {
                        uint16_t* rp = io.packed + i * static_cast<uint64_t>(1u + slots);
                        const int32_t result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            rp[1 + 2 * g] = static_cast<uint16_t>(pb);
                            rp[2 + 2 * g] = static_cast<uint16_t>(pe);
                        }, L.fin_unset);
                        rp[0] = static_cast<uint16_t>(result);
                    }
                }
                if (narrow && io.overflow && __any(clamped != 0u)) {   // (one atomic per wave: this file is built without the compiler's atomic optimizer)
                    uint32_t sum = clamped;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) sum += static_cast<uint32_t>(__shfl_xor(static_cast<int>(sum), d));
                    if (lane == 0) atomicAdd(io.overflow, static_cast<unsigned long long>(sum));
                }
            } else {
                const uint32_t row_b = slots * 4u;
                const bool caps_aligned = ((reinterpret_cast<uintptr_t>(io.caps) | reinterpret_cast<uintptr_t>(io.match_id)) & 15u) == 0u;
                if (full_tile && caps_aligned && 64u * row_b + 256u <= L.stage_bytes) {
                    const uint32_t my_row = stage + lane * row_b;
                    const int32_t result = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                        lds_st<u32x2>(my_row + 8u * g, u32x2{static_cast<uint32_t>(pb), static_cast<uint32_t>(pe)});
                    }, L.fin_unset);
                    const uint32_t ids = stage + 64u * row_b;  // the tile's 64 match ids = 256 bytes
                    lds_st<uint32_t>(ids + 4u * lane, static_cast<uint32_t>(result));
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint8_t* out = reinterpret_cast<uint8_t*>(io.caps + (cur.i - lane) * static_cast<uint64_t>(slots));
                    for (uint32_t c = lane; c < 4u * row_b; c += 64u)  // 64 * row_b / 16 chunks
                        *reinterpret_cast<u32x4*>(out + (c << 4)) = lds_ld<u32x4>(stage + (c << 4));  // (nontemporal: measured slower)
                    if (lane < 16u)
                        *reinterpret_cast<u32x4*>(io.match_id + (cur.i - lane) + 4u * lane) = lds_ld<u32x4>(ids + 16u * lane);
                } else if (valid) {
                    int32_t* cp = io.caps + i * static_cast<uint64_t>(slots);
                    io.match_id[i] = line_result<TIER>(info, fin_lds, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                        cp[2 * g] = pb;
                        cp[2 * g + 1] = pe;
                    }, L.fin_unset);
                }
            }
        }
        if (WIDE) {
            // the per-line walk on the code units takes the flagged lines again (gx_kernels.hip: k_extract_flagged); a line left to
            // the follow-up launch (mode 2) is walked on its units there anyway
            if (valid || (cur.mode == 2 && lane == cur.a)) io.wide_flags[i] = (wide_line && cur.mode != 2) ? 1 : 0;
            if (__builtin_amdgcn_ballot_w64(wide_line && valid && cur.mode != 2) != 0ull && lane == 0)
                __hip_atomic_store(io.wide_any, io.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the staging area is reused by the next tile: all lanes must be done reading it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GX_STAMP(3);
#ifdef GX_DEV
        ++dev_tiles;
#endif
        if (!has_next) break;
        if (!same_group) {  // (wave-uniform)
            t1 = t2;
            need_t2 = true;
        }
        cur = nxt;
        tile = ntile;
        no0 = nno0;
        no1 = nno1;
    }
#ifdef GX_DEV
    if (io.stamps && lane == 0) {
        unsigned long long* s = io.stamps + 8ull * (static_cast<uint64_t>(blockIdx.x) * L.nwaves + wave);
        for (int q = 0; q < 4; ++q) s[q] = phase_cycles[q];
        s[4] = dev_begin;
        s[5] = __builtin_amdgcn_s_memrealtime();
        s[6] = dev_tiles;
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        s[7] = xcc;
    }
#endif
}

template <typename OFF, int KCH, int TIER, int MODE, int PACKED, bool WIDE = false>
hipError_t launch_tile_p(const GxLds& lds, const TileIO& io, dim3 grid, dim3 block, hipStream_t stream) {
    hipError_t e = allow_full_lds(&k_extract_tile<OFF, KCH, TIER, MODE, PACKED, WIDE>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_extract_tile<OFF, KCH, TIER, MODE, PACKED, WIDE>), grid, block, lds.total_bytes, stream, lds, io);
    return hipGetLastError();
}
template <typename OFF, int KCH, int TIER, int MODE, bool WIDE = false>
hipError_t launch_tile_t(const GxLds& lds, const TileIO& io, dim3 grid, dim3 block, hipStream_t stream) {
    if (MODE != 0 && io.packed && io.narrow) return launch_tile_p<OFF, KCH, TIER, MODE, MODE != 0 ? 2 : 0, WIDE>(lds, io, grid, block, stream);
    if (MODE != 0 && io.packed) return launch_tile_p<OFF, KCH, TIER, MODE, MODE != 0 ? 1 : 0, WIDE>(lds, io, grid, block, stream);
    return launch_tile_p<OFF, KCH, TIER, MODE, 0, WIDE>(lds, io, grid, block, stream);
}
// UTF-16 batches: the one variant with 13 staged chunks per lane (lds.stage_bytes <= 13 KB, at most 8 waves: plan_tile_layout)
template <typename OFF, int TIER>
hipError_t launch_tile_wide(int mode, const GxLds& lds, const TileIO& io, dim3 grid, dim3 block, hipStream_t stream) {
    if (lds.stage_bytes > 13u * 1024u || lds.nwaves > 8u) return hipErrorInvalidValue;
    if (mode == 0) return launch_tile_t<OFF, 13, TIER, 0, true>(lds, io, grid, block, stream);
    if (mode == 1) return launch_tile_t<OFF, 13, TIER, 1, true>(lds, io, grid, block, stream);
    if (TIER == TIER_HOP) return hipErrorInvalidValue;
    return launch_tile_t<OFF, 13, TIER, TIER == TIER_HOP ? 1 : 2, true>(lds, io, grid, block, stream);
}
template <typename OFF, int TIER, int MODE>
hipError_t launch_tile_k(const GxLds& lds, const TileIO& io, dim3 grid, dim3 block, hipStream_t stream) {
    const uint32_t kch = (lds.stage_bytes + 1023u) / 1024u;
    if (kch <= 4) return launch_tile_t<OFF, 4, TIER, MODE>(lds, io, grid, block, stream);
    if (kch <= 8) return launch_tile_t<OFF, 8, TIER, MODE>(lds, io, grid, block, stream);
    if (kch <= 13) return launch_tile_t<OFF, 13, TIER, MODE>(lds, io, grid, block, stream);
    if (kch <= 16) return launch_tile_t<OFF, 16, TIER, MODE>(lds, io, grid, block, stream);
    return hipErrorInvalidValue;
}
template <typename OFF, int TIER>
hipError_t launch_tile_m(int mode, const GxLds& lds, const TileIO& io, dim3 grid, dim3 block, hipStream_t stream) {
    if (mode == 0) return launch_tile_k<OFF, TIER, 0>(lds, io, grid, block, stream);
    if (mode == 1) return launch_tile_k<OFF, TIER, 1>(lds, io, grid, block, stream);
    return launch_tile_k<OFF, TIER, 2>(lds, io, grid, block, stream);
}


}  // namespace

// one translation unit per table tier instantiates its kernels (gx_tile_lds.hip, gx_tile_l2.hip, gx_tile_rec.hip,
// gx_tile_recg.hip: they compile in parallel); gx_tile.hip dispatches
#define GX_TILE_TIER_ENTRY(NAME, TIER)                                                                                        \
    hipError_t NAME(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream) {       \
        const TileIO& t = *static_cast<const TileIO*>(io);                                                                    \
        if (off64) return launch_tile_m<uint64_t, TIER>(mode, lds, t, grid, block, stream);                                   \
        return launch_tile_m<uint32_t, TIER>(mode, lds, t, grid, block, stream);                                              \
    }

}  // namespace gx
