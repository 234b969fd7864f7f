// gx_walk.hpp -- device-side building blocks shared by the batch kernels (gx_tile.hip, gx_kernels.hip):
// LDS access by raw byte address, the SWAR self-loop test, and the exact automaton steps over one 16-byte window.
//
// The steps restate, per byte,   p = _transitions[p * stride + _alphabet[c]]   (core/autom/Automata.java:133-135)
// for the match automaton, and the tagged-automaton step that stands for java.util.regex's capture scan
// (core/jdkre/JDKRegexpCookedExtraction.java:36-59) for the fused / per-extraction capture automata.
#pragma once
#include "gx_device.hpp"

namespace gx {

constexpr uint16_t SRC_POS = 0xFFFF;
constexpr uint16_t SRC_NIL = 0xFFFE;
constexpr uint32_t HI_BITS = 0x80808080u;

// native vector types: plain 128/64-bit values the optimiser keeps in registers (HIP's uint4 struct is copied with
// memcpy, which pins arrays of it in scratch memory)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- LDS by byte address ----------------------------------------------------------------------------------
// The kernels own all of LDS as one dynamic allocation that starts at address 0 (no static __shared__ anywhere),
// and every table offset in GxLds is therefore an LDS address.  Going through an address-space-3 pointer made
// from the integer keeps the compiler from adding the (zero) base of a __shared__ symbol to every address --
// one vector instruction per access, and on the dependent chain of the automaton steps.
#define GX_LDS __attribute__((address_space(3)))
template <typename V> __device__ __forceinline__ V lds_ld(uint32_t a) {
    return *(GX_LDS const V*)(uintptr_t)a;
}
template <typename V> __device__ __forceinline__ void lds_st(uint32_t a, V v) {
    *(GX_LDS V*)(uintptr_t)a = v;
}
// 16 bytes at any byte address (gfx950 reads LDS unaligned: ds_read_b128): lets every line start its windows at
// its own first byte, so only the last window of a line can be partial.
struct __attribute__((packed)) UnalignedWindow { u32x4 v; };
__device__ __forceinline__ uint4 lds_window(uint32_t a) {
    const u32x4 v = ((GX_LDS const UnalignedWindow*)(uintptr_t)a)->v;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The same where the wave can tell the alignment: an unaligned ds_read_b128 stalls the LDS pipe for dozens of cycles
// (SQ_LDS_UNALIGNED_STALL: ~48 per wave-instruction on gfx950), two 8-byte-aligned halves (ds_read2_b64) do not.
// Lines of a multiple-of-8 length in an 8-aligned buffer -- fixed-width records -- keep every window 8-aligned.
__device__ __forceinline__ uint4 lds_window_any(uint32_t a) {
    if (!__any((a & 7u) != 0u)) {
        const u32x2 lo = lds_ld<u32x2>(a), hi = lds_ld<u32x2>(a + 8u);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    return lds_window(a);
}

__device__ __forceinline__ uint32_t splat_byte0(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x00000000u); }
__device__ __forceinline__ uint32_t splat_byte1(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x01010101u); }

__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Bit 7 of some byte of the result is set iff some byte of x lies outside [lo, hi] (lo, hi < 0x80), where
// lo4 = lo in every byte and k4 = 0x7F - hi in every byte.  An "any byte" test only: a borrow or carry can
// only leave a byte that is itself out of range, so the lowest offending byte is always reported.
// k = 0x80 encodes "no interval" (every byte fails).
__device__ __forceinline__ uint32_t outside_bits(uint32_t x, uint32_t lo4, uint32_t k4) {
    return or3(x - lo4, x + k4, x);
}
// all 16 bytes of a chunk inside [lo, hi]?  HI7F: hi = 0x7F (k4 = 0; what \\S+, .* and [^x] loops give), where
// "x + k4" is x itself: 10 instructions instead of 16.
template <bool HI7F>
__device__ __forceinline__ bool chunk_inside(const u32x4& v, uint32_t lo4, uint32_t k4) {
    if (HI7F) {
        const uint32_t a = or3(v.x - lo4, v.y - lo4, v.z - lo4), b = or3(v.w - lo4, v.x, v.y);
        return ((or3(a, b, v.z) | v.w) & HI_BITS) == 0u;
    }
    const uint32_t a = or3(outside_bits(v.x, lo4, k4), outside_bits(v.y, lo4, k4), outside_bits(v.z, lo4, k4));
    return ((a | outside_bits(v.w, lo4, k4)) & HI_BITS) == 0u;
}

// Partial window (the last of a line), given the per-dword outside bits: every dword must lie wholly outside the
// line, or wholly inside and in range; a dword that the line boundary cuts through fails (the exact steps handle it).
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

// Where the automaton rows live.  TIER_LDS: dense rows in LDS, a state is the LDS byte address of its row (the host
// bakes the table's position into every successor field).  TIER_L2: dense rows in global memory (they stay resident
// in the XCD's L2), a state is its index and the row address is index * row_bytes.  TIER_REC: sparse range records
// in LDS (gx_api.cpp: records_from_dense), a state is the index of its first 8-byte record.
// TIER_RECG: the same records in global memory (64 KB - 512 KB: they live in the CUs' vector L1 and the XCD's L2, and
// LDS is left to the waves' staging areas).
// TIER_HOP: hop records (run + chain per state) over dense rows in global memory; the staging area holds class ids
// (gx_hop.hpp, gx_hop_dev.hpp).
enum { TIER_LDS = 0, TIER_L2 = 1, TIER_REC = 2, TIER_RECG = 3, TIER_HOP = 4 };
template <int TIER> struct TierTraits { static constexpr bool records = TIER == TIER_REC || TIER == TIER_RECG; };

// wave-uniform description of the automaton being walked
struct WalkTab {
    const uint8_t* at;     // TIER_L2: base of the rows in global memory; TIER_RECG: base of the records
    uint32_t row_bytes;    // dense row stride
    uint32_t ops_off, ops; // LDS addresses of the capture program lists
    uint32_t acc_tab;      // record tiers: LDS address of the interval table
    uint32_t ncls;         // TIER_REC
    uint32_t indexed;      // TIER_REC: states from this index on keep one record per class (read record [state + class])
};

// Record tiers (gx_api.cpp: records_from_dense).  The class map at LDS address 0 has 32-bit entries, class | class * 8
// << 16 (entry 256, for bytes outside the line: REC_IDC, and 0 in the upper half); the records of the LDS tier follow it
// at the fixed address REC_AT, so that a record read is "ds_read_b64 index * 8, offset: REC_AT".  Record 0 is the one
// dead state all automata of the image share.
constexpr uint32_t REC_MORE = 0x80u << 24, REC_HDR = 2u << 24, REC_IDC = 254u, REC_AT = 1088u;
// Second word of a record.  In LDS: successor (16 bits) | program << 16 | flags << 24 (0x80: a second record follows, 2: a
// header precedes).  In global memory the successor has 22 bits (automata of up to 65 536 states each can take more than
// 65 535 records together): successor | program << 22 | header << 30 | second record << 31.
template <int TIER> __device__ __forceinline__ uint32_t rec_target(uint32_t w1) { return TIER == TIER_RECG ? (w1 & 0x3FFFFFu) : (w1 & 0xFFFFu); }
template <int TIER> __device__ __forceinline__ uint32_t rec_op(uint32_t w1) { return TIER == TIER_RECG ? ((w1 >> 22) & 0xFFu) : ((w1 >> 16) & 0xFFu); }
template <int TIER> __device__ __forceinline__ bool rec_has_header(uint32_t w1) { return (w1 & (TIER == TIER_RECG ? (1u << 30) : REC_HDR)) != 0u; }

template <int TIER>
__device__ __forceinline__ uint32_t tab_word(const WalkTab& W, uint32_t row, uint32_t off) {
    if (TIER == TIER_L2) return *reinterpret_cast<const uint32_t*>(W.at + (static_cast<uint64_t>(row) * W.row_bytes + off));
    return lds_ld<uint32_t>(row + off);
}
// one record (two dwords) of state / record index `idx`
template <int TIER>
__device__ __forceinline__ u32x2 rec_ld(const WalkTab& W, uint32_t idx) {
    if (TIER == TIER_RECG) return *reinterpret_cast<const u32x2*>(W.at + (static_cast<uint64_t>(idx) << 3));
    return lds_ld<u32x2>(REC_AT + (idx << 3));
}
// a state's self-loop interval word (lo | (0x7F - hi) << 8 | hot << 16) and its info word
template <int TIER>
__device__ __forceinline__ uint32_t state_acc(const WalkTab& W, uint32_t row) {
    if (TierTraits<TIER>::records) {
        const uint32_t self_lo = rec_ld<TIER>(W, row).x & 0xFFu;            // 255: no self range
        return lds_ld<uint32_t>(W.acc_tab + 4u * min(self_lo, W.ncls));      // entry ncls: no interval
    }
    return tab_word<TIER>(W, row, W.row_bytes - 8u);
}
template <int TIER>
__device__ __forceinline__ int32_t state_info(const WalkTab& W, uint32_t row) {
    if (TierTraits<TIER>::records) {
        const bool hdr = rec_has_header<TIER>(rec_ld<TIER>(W, row).y);
        return hdr ? static_cast<int32_t>(rec_ld<TIER>(W, row - (hdr ? 1u : 0u)).y) : -1;  // the header record precedes the state's first
    }
    return static_cast<int32_t>(tab_word<TIER>(W, row, W.row_bytes - 4u));
}

// General capture program (anything but a single "tag := position"): executed from the LDS copy of the op lists
// (rare).  Inlined on purpose: a real call would force registers that are live across the whole walk to be spilled.
// regs = LDS address of this lane's slot in register column 0.
__device__ __forceinline__ void run_op_list(const WalkTab& W, uint32_t regs, uint32_t op, uint16_t pos) {
    const uint32_t b = lds_ld<uint32_t>(W.ops_off + 4u * op), e = lds_ld<uint32_t>(W.ops_off + 4u * op + 4u);
    for (uint32_t q = b; q < e; ++q) {
        const uint32_t dst = lds_ld<uint16_t>(W.ops + 4u * q), src = lds_ld<uint16_t>(W.ops + 4u * q + 2u);
        lds_st<uint16_t>(regs + dst * 128u, (src == SRC_POS) ? pos : lds_ld<uint16_t>(regs + src * 128u));
    }
}

// 16 exact automaton steps over one staged window.  An entry is (successor | capture program << 16).  The class map
// at LDS address 0 holds class * 4 as 16-bit entries (entry 256 = the identity column, which maps every state to
// itself with no program: MASKED windows -- the last of a line -- send out-of-line bytes there), so the dependent
// chain per byte is one table read plus one add (successor + column offset; the 16-bit halves of the entry are
// picked by the operand selectors of the adds, nothing is unpacked).  The byte->class lookups do not depend on
// the state and are issued up front.
// SIMPLE: every program of the definition is "one register := position"; the program field is then the byte
// offset of that register's column in the wave's register block (0 = a write-only dummy column), so a step is
// branch-free.  regs = LDS address of this lane's slot in register column 0 (one column = 64 lanes x u16; the
// dummy column sits just before it).
template <int TIER, bool CAPTURE, bool MASKED, bool SIMPLE>
__device__ __forceinline__ uint32_t steps16(const uint4& win, uint32_t mask, const WalkTab& W, uint32_t row, uint32_t rel, uint32_t regs) {
    const uint32_t d[4] = {win.x, win.y, win.z, win.w};
    uint32_t c4[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t b = (d[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        if (MASKED) {
            // out-of-line bytes look up entry 256; selecting the index (not the class) keeps the 16 lookups
            // independent of each other, so they are issued back to back
            uint32_t in_line;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(in_line) : "v"(mask), "n"(j));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(b) : "v"(in_line), "v"(b), "v"(256u));
        }
        c4[j] = TierTraits<TIER>::records ? lds_ld<uint32_t>(b << 2) : lds_ld<uint16_t>(b << 1);
    }
    const uint32_t dummy_col = regs - 128u;
    if (TierTraits<TIER>::records) {
        // c4[] holds class-map entries (class | class * 8 << 16).  One record read per byte, two range tests (c - lo <=
        // span, the byte fields picked by operand selectors), three selects: straight-line code.  A state has one or two
        // records, or -- with three exit ranges and more -- one record per class (read at [state + class]); when some
        // lane of the wave finds its class in neither range of a first record that has a second, the wave reads the
        // second records too (a wave-uniform branch, taken once in a few hundred bytes per lane).
        constexpr bool ASM_STEP = TIER == TIER_REC && !MASKED && (SIMPLE || !CAPTURE);
        uint32_t zero = 0u;
        if (ASM_STEP) asm volatile("v_mov_b32 %0, 0" : "=v"(zero));  // (kept in a register: SDWA operands are registers)
        const uint32_t row_in = row;
        uint64_t more = 0ull;  // lanes that needed a second record somewhere in the window
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t ce = c4[j];
            uint32_t next, op;
            if (ASM_STEP) {
                // The step as the instructions it should be (the compiler's version of the C++ below costs 19 vector
                // instructions a byte, and the walk is bound by their issue): 12 for captures, 10 without.
                uint32_t a, x, y;
                uint64_t in_self, pending;
                asm("v_cmp_le_u32 vcc, %1, %2\n\t"
                    "v_cndmask_b32_sdwa %0, %3, %4, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                    "v_lshl_add_u32 %0, %2, 3, %0"
                    : "=&v"(a) : "s"(W.indexed), "v"(row), "v"(zero), "v"(ce) : "vcc");
                const u32x2 it = lds_ld<u32x2>(REC_AT + a);
                asm("v_sub_u32_sdwa %0, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                    "v_sub_u32_sdwa %1, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2\n\t"
                    "v_cmp_le_u32_sdwa %4, %0, %8 src0_sel:DWORD src1_sel:BYTE_1\n\t"
                    "v_cmp_le_u32_sdwa vcc, %1, %8 src0_sel:DWORD src1_sel:BYTE_3\n\t"
                    "v_cmp_gt_i32_e64 %5, 0, %9\n\t"
                    "v_cndmask_b32_e64 %2, 0, %10, %4\n\t"
                    "v_cndmask_b32_sdwa %2, %2, %9, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                    "v_cndmask_b32_sdwa %3, %6, %9, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                    "s_andn2_b64 %5, %5, vcc\n\t"
                    "s_andn2_b64 %5, %5, %4"
                    : "=&v"(x), "=&v"(y), "=&v"(next), "=&v"(op), "=&s"(in_self), "=&s"(pending)
                    : "v"(zero), "v"(ce), "v"(it.x), "v"(it.y), "v"(row) : "vcc", "scc");
                more |= pending;
            } else {
                const uint32_t c = ce & 0xFFu;
                const uint32_t a = row + (row >= W.indexed ? ce >> 19 : 0u);
                const u32x2 it = rec_ld<TIER>(W, a);
                const bool in_exit = (c - ((it.x >> 16) & 0xFFu)) <= (it.x >> 24);
                const bool in_self = (c - (it.x & 0xFFu)) <= ((it.x >> 8) & 0xFFu);
                next = in_exit ? rec_target<TIER>(it.y) : (in_self ? row : 0u);
                op = in_exit ? rec_op<TIER>(it.y) : 0u;
                bool pending = !in_exit && !in_self && (it.y & REC_MORE) != 0u;
                if (MASKED) {
                    const bool idc = c == REC_IDC;  // a byte outside the line: stay, no program
                    next = idc ? row : next;
                    op = idc ? 0u : op;
                    pending = pending && !idc;
                }
                if (__any(pending)) {
                    const u32x2 it2 = rec_ld<TIER>(W, a + 1u);
                    const bool in2 = pending && (c - ((it2.x >> 16) & 0xFFu)) <= (it2.x >> 24);
                    next = in2 ? rec_target<TIER>(it2.y) : next;
                    op = in2 ? rec_op<TIER>(it2.y) : op;
                }
            }
            row = next;
            if (CAPTURE) {
                const uint16_t pos = static_cast<uint16_t>(rel + j);
                if (SIMPLE) {
                    lds_st<uint16_t>(dummy_col + (op << 7), pos);
                } else if (op) {
                    if (op & 0x80u) lds_st<uint16_t>(regs + (op & 0x7Fu) * 128u, pos);
                    else run_op_list(W, regs, op, pos);
                }
            }
        }
        if (ASM_STEP && more != 0ull) {
            // Some lane met a state with two records and found its class in neither range of the first: that lane went to
            // the dead state above (and wrote nothing but the dummy column from there on).  Walk the window again with
            // the second records -- same start, same writes, so this pass simply overwrites the first.
            row = row_in;
            for (int j = 0; j < 16; ++j) {  // (a real loop: no register array indexed by j)
                const uint32_t dw = j < 8 ? (j < 4 ? win.x : win.y) : (j < 12 ? win.z : win.w);
                const uint32_t ce = lds_ld<uint32_t>(((dw >> ((j & 3) * 8)) & 0xFFu) << 2), c = ce & 0xFFu;
                const uint32_t a = row + (row >= W.indexed ? ce >> 19 : 0u);
                const u32x2 it = rec_ld<TIER>(W, a), it2 = rec_ld<TIER>(W, a + 1u);
                const bool in_exit = (c - ((it.x >> 16) & 0xFFu)) <= (it.x >> 24);
                const bool in_self = (c - (it.x & 0xFFu)) <= ((it.x >> 8) & 0xFFu);
                const bool in2 = !in_exit && !in_self && (it.y & REC_MORE) != 0u && (c - ((it2.x >> 16) & 0xFFu)) <= (it2.x >> 24);
                const uint32_t op = in_exit ? rec_op<TIER>(it.y) : in2 ? rec_op<TIER>(it2.y) : 0u;
                row = in_exit ? rec_target<TIER>(it.y) : in_self ? row : in2 ? rec_target<TIER>(it2.y) : 0u;
                if (CAPTURE) lds_st<uint16_t>(dummy_col + (op << 7), static_cast<uint16_t>(rel + j));  // (SIMPLE programs only here)
            }
        }
        return row;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t op = 0;
        if (TIER == TIER_L2) {
            const uint32_t e = *reinterpret_cast<const uint32_t*>(W.at + (static_cast<uint64_t>(row) * W.row_bytes + c4[j]));
            row = e & 0xFFFFu;
            op = e >> 16;
        } else if (CAPTURE) {
            const uint32_t e = lds_ld<uint32_t>(row + c4[j]);
            row = e & 0xFFFFu;
            op = e >> 16;
        } else {
            row = lds_ld<uint16_t>(row + c4[j]);
        }
        if (CAPTURE) {
            const uint16_t pos = static_cast<uint16_t>(rel + j);
            if (SIMPLE) {
                lds_st<uint16_t>(dummy_col + op, pos);
            } else if (op) {
                if (op & 0x8000u) lds_st<uint16_t>(regs + (op & 0x7FFFu) * 128u, pos);  // the common program: one register := position
                else run_op_list(W, regs, op, pos);
            }
        }
    }
    return row;
}

// The result of one line, from the info word of the state its walk ended in.  info < 0: -1 (null) or -2-k
// (ExtractionException), no groups.  info >= 0: byte offset of a final record in the record array (LDS address
// fin_lds, or global pointer fin_g in the L2 tier; 16-byte aligned): u16 [begin tag, end tag] x max_groups, padded to a
// multiple of four groups, then the extraction id; a tag being
// 0 = unset, 1 = the line length, else the byte offset of a register column from the wave's dummy column.  Record 0
// has every tag unset and serves the lines without a match, so the loads below are unconditional and independent:
// all tags, then all registers, then the selects -- two LDS round trips per four groups instead of two per group.
// (Round 5: the first twelve groups' tags, then all their registers, then the rows -- two trips for config 3's ten groups instead of six:
// 0.778 against 0.755 ms, one device; the 36 values it holds at once cost more than the trips.  Not kept.)
// emit(g, begin, end) is called for g = 0 .. G-1 with (-1, -1) for an unset group; returns the match id.
// TIER_HOP (round 5): every tag names a column -- a register's, "the length" or "unset" (hop_unset: its offset; the lane has filled
// both in) -- and a group with one end unset has both unset: two reads and one test per group, no selects on tag values.
template <int TIER, typename EMIT>
__device__ __forceinline__ int32_t line_result(int32_t info, uint32_t fin_lds, const uint8_t* fin_g, uint32_t regs, uint32_t len, int G,
                                               EMIT emit, uint32_t hop_unset = 0u) {
    const uint32_t rec = info >= 0 ? static_cast<uint32_t>(info) : 0u;
    const uint32_t dummy_col = regs - 128u;
    const uint32_t id_at = rec + 16u * static_cast<uint32_t>((G + 3) >> 2);
    uint32_t id;
    // (TIER_HOP keeps small final records in LDS: fin_g == nullptr says so -- wave-uniform)
    const bool FIN_GLOBAL = TIER == TIER_L2 || TIER == TIER_RECG || (TIER == TIER_HOP && fin_g != nullptr);
    if (FIN_GLOBAL) id = *reinterpret_cast<const uint16_t*>(fin_g + id_at);
    else id = lds_ld<uint16_t>(fin_lds + id_at);
    for (int g0 = 0; g0 < G; g0 += 4) {
        u32x4 t;
        if (FIN_GLOBAL) t = *reinterpret_cast<const u32x4*>(fin_g + rec + 4u * g0);
        else t = lds_ld<u32x4>(fin_lds + rec + 4u * g0);
        const uint32_t tw[4] = {t.x, t.y, t.z, t.w};  // one dword = (begin tag, end tag) of one group
        uint32_t vb[4], ve[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vb[q] = lds_ld<uint16_t>(dummy_col + (tw[q] & (TIER == TIER_HOP ? 0xFFFFu : 0xFF80u)));
            ve[q] = lds_ld<uint16_t>(dummy_col + (TIER == TIER_HOP ? tw[q] >> 16 : (tw[q] >> 16) & 0xFF80u));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (g0 + q < G) {
                const uint32_t tb = tw[q] & 0xFFFFu, te = tw[q] >> 16;
                int32_t pb, pe;
                if (TIER == TIER_HOP) {
                    const bool unset = tb == hop_unset;
                    pb = unset ? -1 : static_cast<int32_t>(vb[q]);
                    pe = unset ? -1 : static_cast<int32_t>(ve[q]);
                } else {
                    pb = tb == 1u ? static_cast<int32_t>(len) : static_cast<int32_t>(vb[q]);
                    pe = te == 1u ? static_cast<int32_t>(len) : static_cast<int32_t>(ve[q]);
                    if (tb == 0u || te == 0u) { pb = -1; pe = -1; }
                }
                emit(g0 + q, pb, pe);
            }
        }
    }
    return info >= 0 ? static_cast<int32_t>(static_cast<int16_t>(id)) : info;
}

}  // namespace gx
