// gx_kernels.hip -- gfx950 kernels for the Gorp match-and-extract hot path.
//
// Replaces, per line (lane-per-line, no cross-line state):
//   hot loop #1  PolyMatcher.match      core/autom/PolyMatcher.java:123-133
//                Automata.step/accept   core/autom/Automata.java:133-139
//   hot loop #2  JDKRegexpCookedExtraction.match/_constructMatch
//                                       core/jdkre/JDKRegexpCookedExtraction.java:36-59
//   driver       Gorp.extract           core/Gorp.java:159-186
#include "gx_device.hpp"

namespace gx {

namespace {

constexpr int GENERIC_MAX_REGS = 96;  // must cover GxDev::max_regs (checked on the host)
constexpr uint16_t SRC_POS = 0xFFFF;
constexpr uint16_t SRC_NIL = 0xFFFE;

template <typename CH>
__device__ __forceinline__ int class_of(const GxDev& T, CH ch) {
    uint32_t c = static_cast<uint32_t>(ch);
    if (sizeof(CH) == 1 || c < 256u) return T.cls256[c];
    // code units >= 256: last breakpoint <= c
    int lo = 0, hi = T.n_hi;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (T.hi_lo[mid] <= c) lo = mid; else hi = mid;
    }
    return T.hi_cls[lo];
}

// ---------------------------------------------------------------------------
// One line, tables read through L1/L2 (any table size, any line length).
// ---------------------------------------------------------------------------
template <typename CH, typename MS>
__device__ void extract_line_global(const GxDev& T, const MS* __restrict__ m_next, const CH* __restrict__ s, int64_t len,
                                    uint64_t i, int32_t* __restrict__ match_id, int32_t* __restrict__ caps,
                                    int32_t* __restrict__ state_out, int match_only) {
    const int ncls = T.ncls;
    const int slots = 2 * T.max_groups;
    // match_only < 0: CookedExtraction.match(String) alone (core/jdkre/JDKRegexpCookedExtraction.java:36-39) for
    // extraction -match_only - 1 -- no matcher stage, and a regexp that does not match means null, not an exception
    const bool capture_only = match_only < 0;
    int32_t k = -match_only - 1;
    if (!capture_only) {
        // ---- hot loop #1: walk the match automaton ----
        uint32_t st = 0;
        const uint32_t dead = static_cast<uint32_t>(T.m_dead);
        for (int64_t p = 0; p < len; ++p) {
            st = m_next[static_cast<size_t>(st) * ncls + class_of(T, s[p])];
            if (st == dead) break;  // the reference's early return on -1
        }
        k = T.m_accept_first[st];
        if (state_out) state_out[i] = (st == dead) ? -1 : static_cast<int32_t>(st);
        if (match_only || !T.has_capture) { match_id[i] = k; return; }
    }

    int32_t* cp = caps + i * static_cast<uint64_t>(slots);
    if (k < 0) {
        match_id[i] = -1;
        for (int t = 0; t < slots; ++t) cp[t] = -1;
        return;
    }
    // ---- hot loop #2: walk extraction k's tagged automaton ----
    const uint32_t* tr = T.c_trans + T.c_trans_off[k];
    int32_t regs[GENERIC_MAX_REGS];
    uint32_t ts = 0;
    for (int64_t p = 0; p < len; ++p) {
        const uint32_t w = tr[static_cast<size_t>(ts) * ncls + class_of(T, s[p])];
        ts = w & 0xFFFFu;
        const uint32_t op = w >> 16;
        if (op) {
            for (uint32_t j = T.ops_off[op]; j < T.ops_off[op + 1]; ++j) {
                const uint16_t dst = T.ops[2 * j], src = T.ops[2 * j + 1];
                regs[dst] = (src == SRC_POS) ? static_cast<int32_t>(p) : regs[src];
            }
        }
    }
    const int32_t f = (T.c_fin + T.c_fin_off[k])[ts];
    if (f < 0) {  // DFA said yes, capture regex says no -> ExtractionException (capture alone: just no match)
        match_id[i] = capture_only ? -1 : -2 - k;
        for (int t = 0; t < slots; ++t) cp[t] = -1;
        return;
    }
    const int ng = T.c_ngroups[k];
    for (int g = 0; g < ng; ++g) {
        const uint16_t vb = T.fin_tags[f + 2 * g], ve = T.fin_tags[f + 2 * g + 1];
        int32_t pb = (vb == SRC_POS) ? static_cast<int32_t>(len) : (vb == SRC_NIL ? -1 : regs[vb]);
        int32_t pe = (ve == SRC_POS) ? static_cast<int32_t>(len) : (ve == SRC_NIL ? -1 : regs[ve]);
        if (pb < 0 || pe < 0) { pb = -1; pe = -1; }
        cp[2 * g] = pb;
        cp[2 * g + 1] = pe;
    }
    for (int t = 2 * ng; t < slots; ++t) cp[t] = -1;
    match_id[i] = k;
}

// gx_batch_opts.strip_eol: a line handed over with its terminator (gx_split_lines) loses one trailing '\n' and
// then one trailing '\r' -- "\n", "\r\n" and a lone "\r" are BufferedReader.readLine's three terminators.
template <typename CH>
__device__ __forceinline__ int64_t trim_eol(const CH* __restrict__ s, int64_t len) {
    if (len > 0 && s[len - 1] == 0x0A) --len;
    if (len > 0 && s[len - 1] == 0x0D) --len;
    return len;
}

// Generic kernel: one lane per line.  The correctness backstop for tables that
// do not fit LDS, UTF-16 input and 32-bit match states.
template <typename CH, typename OFF, typename MS>
__global__ void __launch_bounds__(256)
k_extract_generic(GxDev T, const CH* __restrict__ data, const OFF* __restrict__ off, uint64_t n,
                  int32_t* __restrict__ match_id, int32_t* __restrict__ caps, int32_t* __restrict__ state_out,
                  int match_only, const MS* __restrict__ m_next, int strip_eol) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = off[i], e = off[i + 1];
        int64_t len = static_cast<int64_t>(e - b);
        if (strip_eol) len = trim_eol(data + b, len);
        extract_line_global<CH, MS>(T, m_next, data + b, len, i, match_id, caps, state_out, match_only);
    }
}

template <typename CH, typename OFF>
hipError_t launch_generic_t(const GxDev& dev, const GxBatch& b, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    const int block = 256;
    uint64_t blocks = (b.n + block - 1) / block;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    dim3 grid(static_cast<unsigned>(blocks));
    if (dev.m_next16)
        hipLaunchKernelGGL((k_extract_generic<CH, OFF, uint16_t>), grid, dim3(block), 0, stream, dev,
                           static_cast<const CH*>(b.data), static_cast<const OFF*>(b.offsets), b.n, b.match_id, b.caps,
                           b.state_out, b.match_only, dev.m_next16, b.strip_eol);
    else
        hipLaunchKernelGGL((k_extract_generic<CH, OFF, uint32_t>), grid, dim3(block), 0, stream, dev,
                           static_cast<const CH*>(b.data), static_cast<const OFF*>(b.offsets), b.n, b.match_id, b.caps,
                           b.state_out, b.match_only, dev.m_next32, b.strip_eol);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Tile kernel (LDS tier)
// ---------------------------------------------------------------------------
// One wave owns one tile of 64 consecutive lines at a time.  The tile's bytes
// are one contiguous span of the CSR buffer: the wave copies it into its own
// LDS staging area with 16-byte-per-lane coalesced loads (every HBM byte is
// fetched exactly once, in full cache lines), then each lane walks its line
// out of LDS.  All automaton tables (byte->class map, class-compressed match
// table, per-extraction tagged tables, capture programs) live in LDS.
//
// Self-loop acceleration: for every automaton state the host precomputes the
// longest run [lo,hi] of ASCII byte values on which the state loops to itself
// (with no capture operation).  While a lane sits in such a state it tests 16
// (or 4) staged bytes at once with SWAR arithmetic and skips them if all lie in
// [lo,hi]; any other byte falls through to the exact one-byte step.  Log
// templates are dominated by \S+ / \d+ / .* runs, so most bytes take this path.

extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];

constexpr uint32_t HI_BITS = 0x80808080u;

// native vector type: a plain 128-bit value the optimiser keeps in registers (HIP's uint4 struct is copied with
// memcpy, which pins the prefetch array in scratch memory)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t splat_byte0(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x00000000u); }
__device__ __forceinline__ uint32_t splat_byte1(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x01010101u); }

// Bit 7 of some byte of the result is set iff some byte of x lies outside [lo, hi] (lo, hi < 0x80), where
// lo4 = lo in every byte and k4 = 0x7F - hi in every byte.  An "any byte" test only: a borrow or carry can
// only leave a byte that is itself out of range, so the lowest offending byte is always reported.
// k = 0x80 encodes "no interval" (every byte fails).
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t outside_bits(uint32_t x, uint32_t lo4, uint32_t k4) {
    return or3(x - lo4, x + k4, x);
}
// (a << 2) + b in one instruction: the address of column a in the row at b
__device__ __forceinline__ uint32_t lshl2_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Partial window (first / last of a line), given the per-dword outside bits: every dword must lie wholly
// outside the line, or wholly inside and in range; a dword that the line boundary cuts through fails (the
// exact steps handle it).
__device__ __forceinline__ bool partial_window_ok(uint32_t bx, uint32_t by, uint32_t bz, uint32_t bw, uint32_t mask) {
    // spread each nibble of the byte mask over a dword: 0x80 per byte that belongs to the line
    auto spread = [](uint32_t nib) { return ((nib * 0x00204081u) & 0x01010101u) << 7; };
    const uint32_t mx = spread(mask & 15u), my = spread((mask >> 4) & 15u), mz = spread((mask >> 8) & 15u), mw = spread(mask >> 12);
    const bool ox = (mx == 0u) | ((mx == HI_BITS) & ((bx & HI_BITS) == 0u));
    const bool oy = (my == 0u) | ((my == HI_BITS) & ((by & HI_BITS) == 0u));
    const bool oz = (mz == 0u) | ((mz == HI_BITS) & ((bz & HI_BITS) == 0u));
    const bool ow = (mw == 0u) | ((mw == HI_BITS) & ((bw & HI_BITS) == 0u));
    return ox & oy & oz & ow;
}

// bit j set iff byte j of the 16-byte window at `wb` lies inside the line [start, end)
__device__ __forceinline__ uint32_t window_mask(uint32_t start, uint32_t end, uint32_t wb) {
    const uint32_t lo = start > wb ? start - wb : 0u;                 // < 16 for a window that overlaps the line
    const uint32_t hi = end - wb < 16u ? end - wb : 16u;              // 1..16
    return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
}

// General capture program (anything but a single "tag := position"): executed from the LDS copy of the
// op lists (rare).  Inlined on purpose: a real call would force the prefetch registers, which are live
// across the whole walk, to be spilled around it.  regs_b = byte offset in LDS of this lane's register column.
__device__ __forceinline__ void run_op_list(uint32_t ops_off_b, uint32_t ops_b, uint32_t regs_b, uint32_t op, uint16_t pos) {
    const uint32_t* ops_off = reinterpret_cast<const uint32_t*>(gx_smem + ops_off_b);
    const uint16_t* ops = reinterpret_cast<const uint16_t*>(gx_smem + ops_b);
    uint16_t* regs = reinterpret_cast<uint16_t*>(gx_smem + regs_b);
    for (uint32_t q = ops_off[op]; q < ops_off[op + 1]; ++q) {
        const uint32_t dst = ops[2 * q], src = ops[2 * q + 1];
        regs[dst * 64] = (src == SRC_POS) ? pos : regs[src * 64];
    }
}

// One automaton-table word.  LDS tier (GT = false): `row` is the LDS byte address of the state's row (the host
// bakes the table's position into every successor field, so `at` is the start of LDS).  L2 tier (GT = true):
// the table lives in global memory (it stays resident in the XCD's L2), `row` is the state index and the row
// address is state * row_bytes.
template <bool GT>
__device__ __forceinline__ uint32_t tab_read(const uint8_t* at, uint32_t row, uint32_t off, uint32_t rs) {
    if (GT) return *reinterpret_cast<const uint32_t*>(at + (static_cast<uint64_t>(row) * rs + off));
    return *reinterpret_cast<const uint32_t*>(gx_smem + row + off);
}

// 16 exact automaton steps over one staged window.  An entry is (successor | capture program << 16).  In the
// LDS tier the two halves are read as two 16-bit LDS loads, so the dependent chain per byte is one LDS read
// plus one shift-add (successor address + class * 4) and nothing has to be unpacked.  The byte->class lookups
// do not depend on the state and are issued up front.  MASKED windows (first/last of a line) send out-of-line
// bytes through the identity column, which maps every state to itself with no capture program.
// SIMPLE: every program of the definition is "one register := position"; the program field is then the byte
// offset of that register's column in the wave's register block (0 = a write-only dummy column), so a step is
// branch-free.  regs = this lane's slot in register column 0 (one column = 64 lanes x u16, after the dummy).
template <bool CAPTURE, bool MASKED, bool SIMPLE, bool GT>
__device__ __forceinline__ uint32_t steps16(const uint4& win, uint32_t mask, const uint8_t* at, uint32_t row, uint32_t rel,
                                            uint16_t* regs, const GxLds& L) {
    const uint8_t* cmap = gx_smem;  // GxLds: the byte->class map sits at LDS offset 0; cmap[256] = identity column
    const uint32_t d[4] = {win.x, win.y, win.z, win.w};
    uint32_t cls[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t b = (d[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        if (MASKED) {
            // out-of-line bytes look up entry 256; selecting the index (not the class) keeps the 16 lookups
            // independent of each other, so they are issued back to back
            // (written as two instructions; the compiler's own choice is and + compare + select)
            uint32_t in_line;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(in_line) : "v"(mask), "n"(j));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(b) : "v"(in_line), "v"(b), "v"(256u));
        }
        cls[j] = cmap[b];
    }
    uint8_t* dummy_col = reinterpret_cast<uint8_t*>(regs) - 128;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t op = 0;
        if (GT) {
            const uint32_t e = *reinterpret_cast<const uint32_t*>(at + (static_cast<uint64_t>(row) * L.row_bytes + cls[j] * 4u));
            row = e & 0xFFFFu;
            op = e >> 16;
        } else {
            const uint16_t* p = reinterpret_cast<const uint16_t*>(gx_smem + lshl2_add(cls[j], row));
            row = p[0];
            if (CAPTURE) op = p[1];
        }
        if (CAPTURE) {
            const uint16_t pos = static_cast<uint16_t>(rel + j);
            if (SIMPLE) {
                *reinterpret_cast<uint16_t*>(dummy_col + op) = pos;
            } else if (op) {
                if (op & 0x8000u) regs[(op & 0x7FFFu) * 64] = pos;  // the common program: one register := position
                else run_op_list(L.ops_off, L.ops, static_cast<uint32_t>(reinterpret_cast<uint8_t*>(regs) - gx_smem), op, pos);
            }
        }
    }
    return row;
}

// 16 bytes of a staged line at any byte address.  gfx950 reads LDS unaligned (ds_read_b128 at an arbitrary
// address), which lets every line start its windows at its own first byte: no line has a partial first window,
// whatever its alignment in the buffer, and only its last window can be partial.
struct __attribute__((packed)) UnalignedWindow { u32x4 v; };
__device__ __forceinline__ uint4 load_window(const uint8_t* p) {
    const u32x4 v = reinterpret_cast<const UnalignedWindow*>(p)->v;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Walk one automaton over the staged line [start, end), all lanes in lock step over 16-byte windows of their own
// line.  In every window a lane either proves with one SWAR test that all its bytes stay inside the current
// state's self-loop interval (state unchanged, 16 bytes skipped), or takes 16 exact steps (partial windows at
// the ends of a line go through the identity column or, when they pass the range test dword by dword, are
// skipped too).  Returns the row of the final state.
// (A 32-byte look-ahead variant was measured slower: the extra test is paid in the header windows too.)
template <bool CAPTURE, bool GT>
__device__ __forceinline__ uint32_t walk(const uint8_t* stage, const uint8_t* at, uint32_t row, uint32_t start, uint32_t end,
                                         bool on, uint32_t dead_row, uint16_t* regs, const GxLds& L) {
    const uint32_t acc_off = L.row_bytes - 8u; // self-loop interval column: lo | (0x7F - hi) << 8
    uint32_t wb = start;  // windows are relative to the line, not to the staging area: see load_window
    const uint32_t len = end - start;
    const uint32_t full_lim = len >= 16u ? len - 15u : 0u;  // window at line offset rel is full iff rel < full_lim (unsigned)
    uint32_t acc = tab_read<GT>(at, row, acc_off, L.row_bytes);
    uint32_t lo4 = splat_byte0(acc), k4 = splat_byte1(acc);
    bool more = on && start < end;
    while (__any(more)) {
        // a finished lane keeps its last window: the address stays inside the staged tile
        const uint4 w0 = load_window(stage + wb);
        const uint32_t rel = wb - start;  // "negative" (huge) for the first window of a line that starts inside it
        const bool full0 = rel < full_lim;
        const uint32_t bx = outside_bits(w0.x, lo4, k4), by = outside_bits(w0.y, lo4, k4);
        const uint32_t bz = outside_bits(w0.z, lo4, k4), bw = outside_bits(w0.w, lo4, k4);
        bool ok0 = full0 & (((or3(bx, by, bz) | bw) & HI_BITS) == 0u);
        if (L.debug_ablate == 4) ok0 = true;  // timing ablation: no exact steps at all
        bool step = more && !ok0;
        uint32_t mask = 0xFFFFu;
        if (step && !full0) {
            mask = window_mask(start, end, wb);
            step = !partial_window_ok(bx, by, bz, bw, mask);
        }
        // one variant per wave: if any stepping lane has a partial window, every stepping lane takes the masked
        // steps (its mask is all ones) instead of the wave running both variants one after the other
        const bool masked = __any(step && !full0);
        if (step) {
            if (CAPTURE && L.simple_ops) {
                if (!masked) row = steps16<CAPTURE, false, true, GT>(w0, mask, at, row, rel, regs, L);
                else row = steps16<CAPTURE, true, true, GT>(w0, mask, at, row, rel, regs, L);
            } else {
                if (!masked) row = steps16<CAPTURE, false, false, GT>(w0, mask, at, row, rel, regs, L);
                else row = steps16<CAPTURE, true, false, GT>(w0, mask, at, row, rel, regs, L);
            }
            acc = tab_read<GT>(at, row, acc_off, L.row_bytes);
            lo4 = splat_byte0(acc);
            k4 = splat_byte1(acc);
        }
        if (more) wb += 16u;
        more = more && wb < end && row != dead_row;
    }
    return row;
}

// Staging copy for a span that touches the first or last bytes of the buffer: never reads outside [data, data_end).
__device__ void stage_span_guarded(const uint8_t* __restrict__ g_al, uint32_t nch, uint8_t* stage, uint32_t lane,
                                   const uint8_t* data, const uint8_t* data_end) {
    for (uint32_t c = lane; c < nch; c += 64) {
        const uint8_t* src = g_al + (static_cast<uint64_t>(c) << 4);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int q = 0; q < 16; ++q)
            if (src + q >= data && src + q < data_end) w[q >> 2] |= static_cast<uint32_t>(src[q]) << ((q & 3) * 8);
        *reinterpret_cast<uint4*>(stage + (c << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// What one wave needs to know about a round: the lines of a 64-line group (lane = line) that are staged and
// walked together.  Normally one round covers the whole group; when the group's bytes do not fit the staging
// area (lines longer than the hint, mixed lengths) the group is taken in several rounds of consecutive lanes.
// (32-bit flags and no padding: a struct with padding bytes is copied through scratch memory, which costs a
// store + load per round and stalls on the prefetch loads.)
struct TileInfo {
    uint64_t i;            // this lane's line index
    uint64_t o0, o1;       // its byte range in the CSR buffer
    const uint8_t* g_al;   // 16-byte aligned start of the round's span in global memory
    uint32_t nch;          // 16-byte chunks in the span
    uint32_t start, end;   // this lane's line inside the staging area (0, 0 for a lane outside the round)
    uint32_t mode;         // 0: prefetched into registers; 1: touches the buffer edge (guarded copy);
                           // 2: one line that does not fit the staging area (per-lane global path)
    uint32_t active;       // this lane's line belongs to the round
    uint32_t a, b;         // the round covers lanes [a, b) of the group (wave-uniform)
    uint32_t pad_;
};

// Clamped, unconditional accesses on both sides: a lane beyond the span re-reads / rewrites the last chunk
// with identical data.  (Per-lane conditions make the compiler spill the array and serialise the batch.)

// Unconditional as well: a round that is not prefetched (mode != 0, or no next round) loads one dummy chunk instead.
// A branch around the loads would make the compiler lose count of them at the join and wait for all of them --
// i.e. for the whole prefetch -- at the next vector-memory dependency, long before the walk.
template <int KCH>
__device__ __forceinline__ void tile_issue_loads(const TileInfo& t, uint32_t lane, u32x4 (&pre)[KCH], const uint8_t* __restrict__ dummy) {
    const uint8_t* src = t.mode == 0 ? t.g_al : dummy;
    const uint32_t last = t.mode == 0 ? t.nch - 1u : 0u;
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const uint32_t c = min(lane + 64u * k, last);
        pre[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (static_cast<uint64_t>(c) << 4)));
    }
}
template <int KCH>
__device__ __forceinline__ void tile_commit(const TileInfo& t, uint32_t lane, const u32x4 (&pre)[KCH], uint8_t* stage,
                                            const uint8_t* data, const uint8_t* data_end, bool copy) {
    if (t.mode == 0) {
        if (copy) {
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const uint32_t c = min(lane + 64u * k, t.nch - 1u);
                *reinterpret_cast<u32x4*>(stage + (c << 4)) = pre[k];
            }
        }
    } else if (t.mode == 1) {
        stage_span_guarded(t.g_al, t.nch, stage, lane, data, data_end);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// KCH: 16-byte chunks per lane that cover the staging area (stage_bytes <= KCH * 1024).  A tile's span is
// fetched into KCH*4 VGPRs per lane one tile AHEAD: the loads are issued before the current tile is walked
// and land while the wave computes out of LDS, so HBM latency is hidden without a second LDS buffer.
template <typename OFF, int KCH, bool GT>
__global__ void __launch_bounds__(KCH > 13 ? 512 : 768) __attribute__((amdgpu_waves_per_eu(1, KCH > 13 ? 2 : 3)))
k_extract_tile(GxDev T, GxLds L, const uint8_t* __restrict__ lds_image, const uint8_t* __restrict__ at_global,
               const uint8_t* __restrict__ data,
               const OFF* __restrict__ off, uint64_t n, int32_t* __restrict__ match_id, int32_t* __restrict__ caps,
               int match_only, int strip_eol) {
    // ---- prologue: table image -> LDS ----
    {
        const uint4* src = reinterpret_cast<const uint4*>(lds_image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
    }
    __syncthreads();

    // LDS tier: automaton rows, final-tag records and group counts sit in LDS.  L2 tier: they are read from
    // global memory (the uploaded table image); only the byte->class map and the capture programs are in LDS.
    const uint8_t* at = GT ? at_global : gx_smem;              // match automaton rows (LDS tier: rows are LDS addresses)
    const uint8_t* at_c = GT ? at_global + L.c_base : at;      // fused / per-extraction capture rows
    const uint32_t* c_rule = reinterpret_cast<const uint32_t*>(gx_smem + L.c_rule);
    const uint16_t* fin_tags = GT ? T.fin_tags : reinterpret_cast<const uint16_t*>(gx_smem + L.fin_tags);
    const uint32_t info_off = L.row_bytes - 4u;  // per-state info column: accept / final-tags offset

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint8_t* stage = gx_smem + L.stage + wave * L.stage_bytes;
    // register r of this lane = regs[r * 64]; the column before register 0 is a write-only dummy
    uint16_t* regs = reinterpret_cast<uint16_t*>(gx_smem + L.regs + wave * L.regs_wave_bytes) + 64 + lane;

    const int slots = 2 * T.max_groups;
    const bool want_caps = (match_only == 0 || match_only == 3) && T.has_capture;
    const bool caps_aligned = ((reinterpret_cast<uintptr_t>(caps) | reinterpret_cast<uintptr_t>(match_id)) & 15u) == 0u;
    const uint64_t tiles = (n + 63) >> 6;
    const uint64_t wstride = static_cast<uint64_t>(gridDim.x) * L.nwaves;
    const uint8_t* data_end = data + static_cast<uint64_t>(off[n]);

    auto load_offsets = [&](uint64_t tile, uint64_t& o0, uint64_t& o1) {
        const uint64_t i = (tile << 6) + lane;
        const bool valid = i < n;
        o0 = off[valid ? i : n];
        o1 = off[valid ? i + 1 : n];
    };
    // Round of group `tile` starting at lane a: as many consecutive lines as fit the staging area.
    auto make_round = [&](uint64_t tile, uint32_t a, uint64_t o0, uint64_t o1) {
        TileInfo t;
        t.i = (tile << 6) + lane;
        const bool valid = t.i < n;
        t.pad_ = 0;
        t.o0 = o0; t.o1 = o1;
        t.a = a;
        const uint64_t lo = __shfl(static_cast<unsigned long long>(o0), static_cast<int>(a));  // lane a holds a valid line
        const uint8_t* g_lo = data + lo;
        const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(g_lo) & 15u);
        t.g_al = g_lo - skew;  // 16-byte aligned; still a global-address-space pointer for the compiler
        // offsets ascend, so the lines that fit are a run of lanes starting at a (+48: the walk looks two windows ahead)
        const bool fits = lane >= a && valid && (o1 - lo) + skew + 48u <= L.stage_bytes;
        const uint32_t cnt = static_cast<uint32_t>(__popcll(__ballot(fits)));
        if (cnt == 0) {  // line a alone is longer than the staging area
            t.b = a + 1u;
            t.mode = 2;
            t.nch = 1;
        } else {
            t.b = a + cnt;
            const uint64_t hi = __shfl(static_cast<unsigned long long>(o1), static_cast<int>(t.b - 1u));
            const uint64_t span = (hi - lo) + skew;
            t.nch = max(static_cast<uint32_t>((span + 15) >> 4), 1u);
            t.mode = (t.g_al >= data && t.g_al + (static_cast<uint64_t>(t.nch) << 4) <= data_end) ? 0u : 1u;
        }
        t.active = (lane >= a && lane < t.b) ? 1u : 0u;
        t.start = t.active ? skew + static_cast<uint32_t>(o0 - lo) : 0u;
        t.end = t.active ? skew + static_cast<uint32_t>(o1 - lo) : 0u;
        return t;
    };

    uint64_t tile = static_cast<uint64_t>(blockIdx.x) * L.nwaves + wave;
    if (tile >= tiles) return;
    u32x4 pre[KCH];  // the next round's bytes, in flight or landed
    uint64_t no0 = 0, no1 = 0;
    TileInfo cur;
    {
        uint64_t o0, o1;
        load_offsets(tile, o0, o1);
        cur = make_round(tile, 0, o0, o1);
    }
    load_offsets(min(tile + wstride, tiles - 1), no0, no1);
    tile_issue_loads<KCH>(cur, lane, pre, lds_image);

    for (;;) {
        tile_commit<KCH>(cur, lane, pre, stage, data, data_end, match_only != 3);
        // ---- software pipeline: start fetching the next round (and the offsets of the group after it) ----
        const uint32_t group_lines = static_cast<uint32_t>(min(static_cast<uint64_t>(64), n - (tile << 6)));
        const bool same_group = cur.b < group_lines;
        const uint64_t ntile = same_group ? tile : tile + wstride;
        const bool has_next = ntile < tiles;
        // Offsets of the group after the next one first, then the prefetch, and nothing in between that depends on
        // vector memory: both are unconditional (index clamped -- the same values again while the group is
        // unchanged; a dummy chunk when there is no next round), because a conditional load needs a register copy
        // at the join, and that copy would wait for every load issued before it.
        uint64_t nno0, nno1;
        load_offsets(min(ntile + wstride, tiles - 1), nno0, nno1);
        TileInfo nxt = make_round(has_next ? ntile : tile, has_next ? (same_group ? cur.b : 0u) : cur.a,
                                  has_next && !same_group ? no0 : cur.o0, has_next && !same_group ? no1 : cur.o1);
        if (!has_next) nxt.mode = 3;  // nothing to fetch: the loop ends after this round
        tile_issue_loads<KCH>(nxt, lane, pre, lds_image);

        const uint64_t i = cur.i;
        const bool valid = cur.active != 0u;
        const uint32_t start = cur.start;
        uint32_t end = cur.end;
        if (strip_eol && cur.mode != 2) {  // the terminator is staged with the line (trim_eol, from LDS)
            if (end > start && stage[end - 1u] == 0x0Au) --end;
            if (end > start && stage[end - 1u] == 0x0Du) --end;
        }
        if (cur.mode == 2) {
            // tile does not fit the staging area (very long lines): exact per-lane path from global memory
            if (valid) {
                int64_t len = static_cast<int64_t>(cur.o1 - cur.o0);
                if (strip_eol) len = trim_eol(data + cur.o0, len);
                extract_line_global<uint8_t, uint16_t>(T, T.m_next16, data + cur.o0, len, i, match_id, caps, nullptr, match_only);
            }
        } else if (match_only == 2) {  // timing ablation (GX_DEBUG_ABLATE=2): staging only
            if (valid) match_id[i] = stage[start];
        } else if (!want_caps) {
            // ---- hot loop #1 alone: PolyMatcher.match ----
            const uint32_t mrow = walk<false, GT>(stage, at, L.m_start, start, end, true, L.m_dead, regs, L);
            if (valid) match_id[i] = static_cast<int32_t>(tab_read<GT>(at, mrow, info_off, L.row_bytes));
        } else {
            int32_t result, f = -1, tag0 = 0;
            uint32_t ng = 0;
            if (L.u_start != 0xFFFFFFFFu) {
                // ---- fused pass: match automaton x joined capture automata, one walk ----
                const uint32_t urow = walk<true, GT>(stage, at_c, L.u_start, start, end, true, L.u_dead, regs, L);
                const int32_t info = static_cast<int32_t>(tab_read<GT>(at_c, urow, info_off, L.row_bytes));
                result = info;  // -1: null, -2-k: ExtractionException
                if (info >= 0) {
                    result = fin_tags[info];  // the record starts with the winning extraction
                    ng = GT ? static_cast<uint32_t>(T.c_ngroups[result]) : c_rule[2 * result + 1];
                    f = info;
                    tag0 = 1;
                }
            } else {
                // ---- hot loop #1, then hot loop #2 on extraction k's tagged automaton ----
                const uint32_t mrow = walk<false, GT>(stage, at, L.m_start, start, end, true, L.m_dead, regs, L);
                const int32_t k = static_cast<int32_t>(tab_read<GT>(at, mrow, info_off, L.row_bytes));
                result = k;
                uint32_t crow = GT ? 0u : L.m_dead;  // any valid row: the walk below is off for lanes without a match
                if (k >= 0) {
                    crow = c_rule[2 * k];
                    ng = c_rule[2 * k + 1];
                }
                crow = walk<true, GT>(stage, at_c, crow, start, end, k >= 0, 0xFFFFFFFFu, regs, L);
                if (k >= 0) {
                    f = static_cast<int32_t>(tab_read<GT>(at_c, crow, info_off, L.row_bytes));
                    if (f < 0) result = -2 - k;  // DFA said yes, capture regex says no -> ExtractionException
                }
            }
            // capture offsets of this lane's line: group g -> (begin, end), (-1, -1) when unset
            auto group_span = [&](int g, int32_t& pb, int32_t& pe) {
                pb = -1; pe = -1;
                if (f >= 0 && static_cast<uint32_t>(g) < ng) {
                    const int32_t len = static_cast<int32_t>(end - start);
                    const uint16_t vb = fin_tags[f + tag0 + 2 * g], ve = fin_tags[f + tag0 + 2 * g + 1];
                    pb = (vb == SRC_POS) ? len : (vb == SRC_NIL ? -1 : static_cast<int32_t>(regs[vb * 64]));
                    pe = (ve == SRC_POS) ? len : (ve == SRC_NIL ? -1 : static_cast<int32_t>(regs[ve * 64]));
                    if (pb < 0 || pe < 0) { pb = -1; pe = -1; }
                }
            };
            const int G = T.max_groups;
            const uint32_t row_b = static_cast<uint32_t>(slots) * 4u;
            const bool full_tile = cur.a == 0u && cur.b == 64u;
            if (full_tile && caps_aligned && 64u * row_b + 256u <= L.stage_bytes) {
                // The tile's 64 capture rows are one contiguous block of the output.  Transpose through the staging
                // area (free now: every lane has finished its walk) so that each store instruction writes 1 KiB of
                // consecutive bytes, instead of every lane writing pieces of its own row.
                uint8_t* my_row = stage + lane * row_b;
                for (int g = 0; g < G; ++g) {
                    int32_t pb, pe;
                    group_span(g, pb, pe);
                    *reinterpret_cast<int2*>(my_row + g * 8) = make_int2(pb, pe);
                }
                int32_t* ids = reinterpret_cast<int32_t*>(stage + 64u * row_b);  // the tile's 64 match ids = 256 bytes
                ids[lane] = result;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint8_t* out = reinterpret_cast<uint8_t*>(caps + (cur.i - lane) * static_cast<uint64_t>(slots));
                if (L.debug_ablate != 5)  // (timing ablation 5: no capture stores)
                for (uint32_t c = lane; c < 4u * row_b; c += 64u)  // 64 * row_b / 16 chunks
                    *reinterpret_cast<u32x4*>(out + (c << 4)) = *reinterpret_cast<const u32x4*>(stage + (c << 4));  // (nontemporal: measured slower)
                if (lane < 16u)
                    *reinterpret_cast<u32x4*>(match_id + (cur.i - lane) + 4u * lane) = *reinterpret_cast<const u32x4*>(ids + 4u * lane);
            } else if (valid) {
                int32_t* cp = caps + i * static_cast<uint64_t>(slots);
                for (int g = 0; g < G; ++g) {
                    int32_t pb, pe;
                    group_span(g, pb, pe);
                    cp[2 * g] = pb;
                    cp[2 * g + 1] = pe;
                }
                match_id[i] = result;
            }
        }
        // the staging area is reused by the next tile: all lanes must be done reading it
        __builtin_amdgcn_wave_barrier();
        if (!has_next) break;
        cur = nxt;
        tile = ntile;
        no0 = nno0;
        no1 = nno1;
    }
}

// ---------------------------------------------------------------------------
// Slice kernel: lines staged 64 bytes at a time, lanes refilled as their lines end
// ---------------------------------------------------------------------------
// The tile kernel stages whole lines, so LDS holds 64 x line-length bytes per wave: long lines mean few waves, a
// group of long or uneven lines has to go in several rounds of a few lanes each, and every lane waits for the
// longest line of its round.  Here a wave owns a contiguous range of lines and stages only the next 64-byte slice
// of the line each lane is on (a [64][80]-byte buffer, 5 KB per wave whatever the lengths).  A lane whose line
// has ended (or whose automaton has died) writes its result and takes the next line of the range, so all lanes
// stay busy however uneven the lines are, and a line may be as long as the 16-bit positions allow.  The price is
// the load pattern -- four instructions per slice, each fetching 16-byte pieces of 16 different lines (unaligned
// global loads), not prefetched -- and per-line result stores.  Used for batches whose lines do not fit the tile
// kernel's staging (mean length above 255 bytes).  Walks the fused automaton (or the match automaton alone);
// needs Tables::union_ok for captures.
constexpr uint32_t SLICE_BYTES = 64, SLICE_ROW = 80;

template <typename OFF, bool GT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_extract_slices(GxDev T, GxLds L, const uint8_t* __restrict__ lds_image, const uint8_t* __restrict__ at_global,
                 const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n, int32_t* __restrict__ match_id,
                 int32_t* __restrict__ caps, int match_only, int strip_eol) {
    {
        const uint4* src = reinterpret_cast<const uint4*>(lds_image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
    }
    __syncthreads();
    const uint8_t* at = GT ? at_global : gx_smem;
    const uint8_t* at_c = GT ? at_global + L.c_base : at;
    const uint32_t* c_rule = reinterpret_cast<const uint32_t*>(gx_smem + L.c_rule);
    const uint16_t* fin_tags = GT ? T.fin_tags : reinterpret_cast<const uint16_t*>(gx_smem + L.fin_tags);
    const uint32_t info_off = L.row_bytes - 4u;
    const uint32_t acc_off = L.row_bytes - 8u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint8_t* slice = gx_smem + L.stage + wave * L.stage_bytes;
    uint16_t* regs = reinterpret_cast<uint16_t*>(gx_smem + L.regs + wave * L.regs_wave_bytes) + 64 + lane;
    const int slots = 2 * T.max_groups;
    const bool want_caps = match_only == 0 && T.has_capture;
    const uint8_t* data_end = data + static_cast<uint64_t>(off[n]);
    const uint8_t* A = want_caps ? at_c : at;                       // the automaton this launch walks
    const uint32_t row0 = want_caps ? L.u_start : L.m_start, dead_row = want_caps ? L.u_dead : L.m_dead;

    // this wave's range of lines
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * L.nwaves;
    const uint64_t per_wave = (n + nwaves - 1) / nwaves;
    const uint64_t wid = static_cast<uint64_t>(blockIdx.x) * L.nwaves + wave;
    const uint64_t range_lo = min(n, wid * per_wave), range_hi = min(n, range_lo + per_wave);
    uint64_t next = range_lo;   // first line of the range not yet handed to a lane (wave-uniform)

    // the line this lane is on
    bool has_line = false;
    uint64_t i = 0, o0 = 0;
    uint32_t len = 0, pos = 0, row = row0;
    uint32_t lo4 = 0u, k4 = HI_BITS;
    const uint8_t* my = slice + lane * SLICE_ROW;

    for (;;) {
        // ---- lanes whose line is finished write its result -- not every time one finishes: lanes run in lock step,
        // so this block (and the refill below) costs every lane of the wave its full length whenever a single lane
        // enters it.  Finished lanes wait until a quarter of the wave is idle (or nothing is left to walk). ----
        const bool finished = has_line && (pos >= len || row == dead_row);
        const uint32_t idle = static_cast<uint32_t>(__popcll(__ballot(finished || !has_line)));
        const bool service = idle >= 16u || !__any(has_line && !finished);
        const bool done = service && finished;
        if (done) {
            const int32_t info = static_cast<int32_t>(tab_read<GT>(A, row, info_off, L.row_bytes));
            if (!want_caps) match_id[i] = info;
            else {
                int32_t result = info, f = -1;
                uint32_t ng = 0;
                if (info >= 0) {
                    result = fin_tags[info];  // the record starts with the winning extraction
                    ng = GT ? static_cast<uint32_t>(T.c_ngroups[result]) : c_rule[2 * result + 1];
                    f = info;
                }
                int32_t* cp = caps + i * static_cast<uint64_t>(slots);
                for (int g = 0; g < T.max_groups; ++g) {
                    int32_t pb = -1, pe = -1;
                    if (f >= 0 && static_cast<uint32_t>(g) < ng) {
                        const uint16_t vb = fin_tags[f + 1 + 2 * g], ve = fin_tags[f + 1 + 2 * g + 1];
                        pb = (vb == SRC_POS) ? static_cast<int32_t>(len) : (vb == SRC_NIL ? -1 : static_cast<int32_t>(regs[vb * 64]));
                        pe = (ve == SRC_POS) ? static_cast<int32_t>(len) : (ve == SRC_NIL ? -1 : static_cast<int32_t>(regs[ve * 64]));
                        if (pb < 0 || pe < 0) { pb = -1; pe = -1; }
                    }
                    cp[2 * g] = pb;
                    cp[2 * g + 1] = pe;
                }
                match_id[i] = result;
            }
            has_line = false;
        }
        // ---- free lanes take the next lines of the range, in lane order ----
        const uint64_t free_mask = __ballot(!has_line);
        if (service && free_mask && next < range_hi) {
            const uint32_t rank = static_cast<uint32_t>(__popcll(free_mask & ((1ull << lane) - 1ull)));
            const uint64_t cand = next + rank;
            if (!has_line && cand < range_hi) {
                i = cand;
                o0 = off[i];
                int64_t len64 = static_cast<int64_t>(static_cast<uint64_t>(off[i + 1]) - o0);
                if (strip_eol) len64 = trim_eol(data + o0, len64);
                if (len64 > 65535) {
                    // positions are 16-bit in the register block: such a line takes the per-lane path, whole
                    extract_line_global<uint8_t, uint16_t>(T, T.m_next16, data + o0, len64, i, match_id, caps, nullptr, match_only);
                } else {
                    has_line = true;
                    len = static_cast<uint32_t>(len64);
                    pos = 0;
                    row = row0;
                    const uint32_t acc = tab_read<GT>(A, row, acc_off, L.row_bytes);
                    lo4 = splat_byte0(acc);
                    k4 = splat_byte1(acc);
                }
            }
            next = min(range_hi, next + static_cast<uint64_t>(__popcll(free_mask)));
        }
        if (!__any(has_line)) {
            if (next >= range_hi) break;
            continue;  // (only lines for the per-lane path were handed out: hand out more)
        }
        // ---- stage the next slice of every lane's line: lane l fetches chunk l & 3 of the lines of lanes (l >> 2) + 16 r ----
        const bool walking = has_line && pos < len && row != dead_row;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = static_cast<int>(lane >> 2) + 16 * r;
            const uint64_t oq = __shfl(static_cast<unsigned long long>(o0), q);
            const uint32_t pq = static_cast<uint32_t>(__shfl(static_cast<int>(pos), q));
            const uint32_t lq = static_cast<uint32_t>(__shfl(static_cast<int>(walking ? len : 0u), q));
            const uint32_t at_byte = pq + (lane & 3u) * 16u;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (at_byte < lq) {
                const uint8_t* src = data + oq + at_byte;
                if (src + 16 <= data_end) v = __builtin_nontemporal_load(&reinterpret_cast<const UnalignedWindow*>(src)->v);
                else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; ++b)
                        if (src + b < data_end) w[b >> 2] |= static_cast<uint32_t>(src[b]) << ((b & 3) * 8);
                    v = u32x4{w[0], w[1], w[2], w[3]};
                }
            }
            *reinterpret_cast<u32x4*>(slice + static_cast<uint32_t>(q) * SLICE_ROW + (lane & 3u) * 16u) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- walk the slice: four 16-byte windows of every lane's own line ----
        bool more = walking;
        for (uint32_t w = 0; w < SLICE_BYTES / 16u; ++w) {
            const uint32_t rel = pos + w * 16u;
            const bool act = more && rel < len;
            if (!__any(act)) break;
            const uint4 w0 = *reinterpret_cast<const uint4*>(my + w * 16u);
            const bool full0 = rel + 16u <= len;
            const uint32_t bx = outside_bits(w0.x, lo4, k4), by = outside_bits(w0.y, lo4, k4);
            const uint32_t bz = outside_bits(w0.z, lo4, k4), bw = outside_bits(w0.w, lo4, k4);
            const bool ok0 = full0 & (((or3(bx, by, bz) | bw) & HI_BITS) == 0u);
            bool step = act && !ok0;
            uint32_t mask = 0xFFFFu;
            if (step && !full0) {
                mask = window_mask(0u, len, rel);
                step = !partial_window_ok(bx, by, bz, bw, mask);
            }
            const bool masked = __any(step && !full0);
            if (step) {
                if (want_caps) {
                    if (L.simple_ops) {
                        if (!masked) row = steps16<true, false, true, GT>(w0, mask, A, row, rel, regs, L);
                        else row = steps16<true, true, true, GT>(w0, mask, A, row, rel, regs, L);
                    } else {
                        if (!masked) row = steps16<true, false, false, GT>(w0, mask, A, row, rel, regs, L);
                        else row = steps16<true, true, false, GT>(w0, mask, A, row, rel, regs, L);
                    }
                } else {
                    if (!masked) row = steps16<false, false, false, GT>(w0, mask, A, row, rel, regs, L);
                    else row = steps16<false, true, false, GT>(w0, mask, A, row, rel, regs, L);
                }
                const uint32_t acc = tab_read<GT>(A, row, acc_off, L.row_bytes);
                lo4 = splat_byte0(acc);
                k4 = splat_byte1(acc);
            }
            more = more && row != dead_row;
        }
        if (walking) pos += SLICE_BYTES;
        // the slice buffer is rewritten by the next iteration
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t launch_extract_generic(const GxDev& dev, const GxBatch& b, hipStream_t stream) {
    if (b.wide) {
        return b.offsets64 ? launch_generic_t<uint16_t, uint64_t>(dev, b, stream) : launch_generic_t<uint16_t, uint32_t>(dev, b, stream);
    }
    return b.offsets64 ? launch_generic_t<uint8_t, uint64_t>(dev, b, stream) : launch_generic_t<uint8_t, uint32_t>(dev, b, stream);
}

namespace {
template <typename OFF, int KCH, bool GT>
hipError_t launch_tile_t(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, dim3 grid, dim3 block,
                         const GxBatch& b, hipStream_t stream) {
    static bool prepared = false;  // per instantiation; the attribute is sticky for the process
    if (!prepared) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extract_tile<OFF, KCH, GT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return e;
        prepared = true;
    }
    hipLaunchKernelGGL((k_extract_tile<OFF, KCH, GT>), grid, block, lds.total_bytes, stream, dev, lds, lds_image, at_global,
                       static_cast<const uint8_t*>(b.data), static_cast<const OFF*>(b.offsets), b.n, b.match_id, b.caps, b.match_only, b.strip_eol);
    return hipGetLastError();
}
template <typename OFF, bool GT>
hipError_t launch_tile_o(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, dim3 grid, dim3 block,
                         const GxBatch& b, hipStream_t stream) {
    const uint32_t kch = (lds.stage_bytes + 1023u) / 1024u;
    if (kch <= 4) return launch_tile_t<OFF, 4, GT>(dev, lds, lds_image, at_global, grid, block, b, stream);
    if (kch <= 8) return launch_tile_t<OFF, 8, GT>(dev, lds, lds_image, at_global, grid, block, b, stream);
    if (kch <= 13) return launch_tile_t<OFF, 13, GT>(dev, lds, lds_image, at_global, grid, block, b, stream);
    if (kch <= 16) return launch_tile_t<OFF, 16, GT>(dev, lds, lds_image, at_global, grid, block, b, stream);
    return hipErrorInvalidValue;
}
}  // namespace

hipError_t prepare_tile_kernels(uint32_t) { return hipSuccess; }

namespace {
template <typename OFF, bool GT>
hipError_t launch_slices_t(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, dim3 grid, dim3 block,
                           const GxBatch& b, hipStream_t stream) {
    static bool prepared = false;
    if (!prepared) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extract_slices<OFF, GT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return e;
        prepared = true;
    }
    hipLaunchKernelGGL((k_extract_slices<OFF, GT>), grid, block, lds.total_bytes, stream, dev, lds, lds_image, at_global,
                       static_cast<const uint8_t*>(b.data), static_cast<const OFF*>(b.offsets), b.n, b.match_id, b.caps, b.match_only, b.strip_eol);
    return hipGetLastError();
}
}  // namespace

// Slice kernel (see k_extract_slices): lds.stage_bytes = 64 * 80, lds.nwaves <= 16.
hipError_t launch_extract_slices(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                 const GxBatch& b, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    // every wave owns a contiguous range of lines; at least 256 lines per wave so that lanes can be refilled
    uint64_t blocks = static_cast<uint64_t>(num_cus);
    const uint64_t need = (b.n + 256ull * lds.nwaves - 1) / (256ull * lds.nwaves);
    if (blocks > need) blocks = need;
    dim3 grid(static_cast<unsigned>(blocks)), block(lds.nwaves * 64);
    if (at_global) {
        if (b.offsets64) return launch_slices_t<uint64_t, true>(dev, lds, lds_image, at_global, grid, block, b, stream);
        return launch_slices_t<uint32_t, true>(dev, lds, lds_image, at_global, grid, block, b, stream);
    }
    if (b.offsets64) return launch_slices_t<uint64_t, false>(dev, lds, lds_image, nullptr, grid, block, b, stream);
    return launch_slices_t<uint32_t, false>(dev, lds, lds_image, nullptr, grid, block, b, stream);
}

hipError_t launch_extract_tile(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                               const GxBatch& b, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    const uint64_t tiles = (b.n + 63) >> 6;
    uint32_t blocks_per_cu = 163840u / lds.total_bytes;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * lds.nwaves > 12) blocks_per_cu = 1;  // the kernel is built for <= 3 waves per SIMD
    uint64_t blocks = static_cast<uint64_t>(num_cus) * blocks_per_cu;
    const uint64_t need = (tiles + lds.nwaves - 1) / lds.nwaves;
    if (blocks > need) blocks = need;
    dim3 grid(static_cast<unsigned>(blocks)), block(lds.nwaves * 64);
    if (at_global) {
        if (b.offsets64) return launch_tile_o<uint64_t, true>(dev, lds, lds_image, at_global, grid, block, b, stream);
        return launch_tile_o<uint32_t, true>(dev, lds, lds_image, at_global, grid, block, b, stream);
    }
    if (b.offsets64) return launch_tile_o<uint64_t, false>(dev, lds, lds_image, nullptr, grid, block, b, stream);
    return launch_tile_o<uint32_t, false>(dev, lds, lds_image, nullptr, grid, block, b, stream);
}

}  // namespace gx
