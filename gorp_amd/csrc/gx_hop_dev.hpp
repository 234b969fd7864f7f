// gx_hop_dev.hpp -- device side of the HOP tier (tables: gx_hop.hpp): the walk of the fused automaton that consumes a
// run of bytes and a chain of up to eight bytes per iteration instead of one byte per step.
//
// Replaces the same reference loops as walk<> (gx_tile_body.hpp): PolyMatcher.match core/autom/PolyMatcher.java:123-133
// over Automata.step core/autom/Automata.java:133-135, with JDKRegexpCookedExtraction.match
// core/jdkre/JDKRegexpCookedExtraction.java:36-59 fused in.
//
// The staging area holds the lines' BYTES as they are (until round 4 it held class ids: 208 LDS lookups per lane and tile
// to map them, a fifth of the kernel; the tables' intervals are byte intervals now, gx_hop.cpp), every lane walks its own
// line at its own position:
//   1. the state's hop record (24 bytes; LDS for the hot states, global memory / L2 for the others);
//   2. 16 bytes at the lane's position; one SWAR interval test finds how many of them the state's RUN covers;
//   3. 8 bytes behind the run against the chain: its single bytes with two v_msad_u8 (a reference byte of 0 is not compared), its one
//      interval element -- the "tail", typically the first byte of the next field -- picked with v_perm.  Match: the lane is at the
//      chain's target, klen bytes further, and the chain's (at most two) capture programs are written;
//   4. no match: ONE exact step through the state's dense row (global memory / L2; the byte's class id from the map at LDS
//      address 0) -- a wave-uniform branch that the waves of a well-formed log take for the branching nodes of the literal
//      trie only.  A byte with its top bit set fails every interval test: it always takes the exact step.
// LDS is read in aligned dwords only (an unaligned ds_read stalls the LDS pipe for dozens of cycles on gfx950): a window at
// byte address p is five (or three) dwords from p & ~3, joined by v_alignbyte.
#pragma once
#include "gx_walk.hpp"

namespace gx {

constexpr uint32_t HOP_REC_B = 24, HOP_LDS_AT = 272, LOW7 = 0x7F7F7F7Fu;

struct HopTab {
    const uint8_t* rows;   // dense rows u32[n_states][ncls + 1] (global): successor | register column << 16; last column: info
    const uint8_t* hops;   // hop records of all states (global)
    uint32_t row_bytes;
    uint32_t info_off;     // byte offset of the info column in a row
    uint32_t n_hot;        // states [0, n_hot) have their records in LDS at HOP_LDS_AT
    uint32_t lrow_cols;    // a dense row's copy in LDS: u16 successors, and at this byte offset its u8 register columns
    uint32_t sets;         // LDS address of the loop sets: u8 lo[4], u8 k[4] per entry (a record's byte 3 is its state's entry; 0 = none)
};

// two / one dwords at a 4-byte aligned LDS address (ds_read2_b32 / ds_read_b32)
struct __attribute__((packed, aligned(4))) HopPair { u32x2 v; };
__device__ __forceinline__ u32x2 lds_pair4(uint32_t a) { return ((GX_LDS const HopPair*)(uintptr_t)a)->v; }
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (mask & a) | (~mask & b)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
// The record's fields sit on byte and word boundaries so that an SDWA operand takes them as they are: a + byte / word N of w
#define GX_SDWA_ADD(NAME, SEL)                                                                                                   \
    __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t w) {                                                           \
        uint32_t r;                                                                                                              \
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL : "=v"(r) : "v"(a), "v"(w)); \
        return r;                                                                                                                \
    }
GX_SDWA_ADD(add_byte2, "BYTE_2")
GX_SDWA_ADD(add_byte3, "BYTE_3")
GX_SDWA_ADD(add_word0, "WORD_0")
GX_SDWA_ADD(add_word1, "WORD_1")
#undef GX_SDWA_ADD
// byte 0 of a, minus byte 1 of w
__device__ __forceinline__ uint32_t sub_b0_b1(uint32_t a, uint32_t w) {
    uint32_t r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(w));
    return r;
}

// (the compiler turns __any into a select, a compare and a scalar test; the ballot is the scalar test alone)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// Walks the line [start, end) of the wave's staging area from state `s`; returns the final state.  regs = LDS
// address of this lane's slot in register column 0 (the write-only dummy column sits 128 bytes before it).
// A lane that is done keeps running the loop with the wave: at the end of its line its position no longer moves (no
// bytes left), and a lane that steps into the dead state is moved to the end of its line.  Wave-level tests are ANDs of the
// lane masks of single compares (the compiler turns the ballot of a compound condition into a select and a compare).
__device__ __forceinline__ uint32_t run_flags(uint32_t x, uint32_t lo4, uint32_t k4) {
    // bit 7 of a byte of the result is set iff that byte lies outside [lo, hi] (hi <= 0x7F: a byte with its top bit set is
    // outside by itself): (x | x - lo4 | x + k4) & 0x80808080; exact for the lowest offending byte -- borrows and carries only
    // travel upwards from one -- which is all the walk uses
    return (x | (x - lo4) | (x + k4)) & HI_BITS;
}
// General form (the slice kernel stages a piece of the line at a time): the lane is at LDS address p; staged bytes end
// at `e` (run lengths are cut there); it iterates while p < limit (limit = e when the line ends at e, else 24 bytes
// before it, so that a whole window and a whole chain are staged); a chain must end at or before e_chain (the end of the
// LINE; no constraint when the line goes on beyond the staged piece); a capture program's position is its LDS address minus p0.
// ALL_HOT: every state that a chain or a plausible exact step leads to has its record in LDS (the usual case: a few hundred
// states).  A lane then leaves the hot set only by an exact step on a byte the definition does not expect there; it takes
// exact steps -- inside that branch -- until it is back in the hot set or its line is over, and the loop proper never looks
// for a record in global memory.
// CAPTURE false: the match automaton's tables (no programs anywhere): the register writes are left out.
// The record a lane holds while it stays in its state (the !ALL_HOT walk; the hop slice kernel keeps it across its rounds and
// reads the run interval out of it for the loaders' run test)
struct HopKept {
    uint32_t hint = 0u;   // see walk_hop_span's SECOND
#ifdef GX_DEV
    uint32_t dev_iters = 0u, dev_lane_iters = 0u;   // developer build: iterations of the walk, and lanes that had something to walk in them
#endif
    uint32_t kept = 0xFFFFFFFFu;
    u32x2 k0 = {0u, 0u}, k1 = {0u, 0u}, k2 = {0u, 0u};
};
// SECOND: the second chances (loop sets, tail sets) are in the loop.  Without them the walk is the same walk -- a lane takes its exact
// step -- and only notes in K.hint that a lane took one in a state that HAS a set: the tile kernel then walks its next tile with them
// (K.hint from such a walk: a second chance applied), and goes back when a tile needed none.  The code of the second chances costs
// the loop 4 % even where it never runs (config 3, lower-case text: 0.930 against 0.891 ms, one device).
template <bool ALL_HOT, bool CAPTURE, bool SECOND = true>
__device__ __forceinline__ uint32_t walk_hop_span(const HopTab& H, uint32_t& p, uint32_t e, uint32_t limit, uint32_t e_chain, uint32_t p0,
                                                  uint32_t s, uint32_t dead, uint32_t regs, uint32_t leave_at, HopKept& K) {
    const uint32_t dummy_col = regs - 128u;
    const uint32_t last_hot = H.n_hot - 1u;
    // one exact step of this lane from state s on the byte `bt` at LDS address q: the dense row (lrow: its copy in LDS, 0: none)
    auto exact_step_c = [&](uint32_t lrow, uint32_t c, uint32_t q) {   // c: the class id of the byte at q (u8[256] at LDS address 0)
        uint32_t xe;
        if (lrow) xe = lds_ld<uint16_t>(lrow + (c << 1)) | static_cast<uint32_t>(lds_ld<uint8_t>(lrow + H.lrow_cols + c)) << 16;  // (u16 successors, then u8 columns)
        else xe = *reinterpret_cast<const uint32_t*>(H.rows + (static_cast<uint64_t>(s) * H.row_bytes + (c << 2)));
        if (CAPTURE) lds_st<uint16_t>(dummy_col + ((xe >> 16) << 7), static_cast<uint16_t>(q - p0));
        s = xe & 0xFFFFu;
        p = s == dead ? max(limit, q + 1u) : q + 1u;  // (nothing leaves the dead state: the line is over)
    };
    auto exact_step = [&](uint32_t lrow, uint32_t bt, uint32_t q) { exact_step_c(lrow, lds_ld<uint8_t>(bt), q); };
    if (ALL_HOT && __builtin_amdgcn_ballot_w64(s > last_hot) != 0ull)   // (the slice kernel: a lane may come back off the path)
        while (s > last_hot && p < limit) exact_step(0u, lds_ld<uint8_t>(p), p);
    // Where not every reachable state's record is in LDS (the hop slice kernel on definitions of hundreds of extractions), a record
    // stays with its lane while the lane stays in its state: a value of several windows is as many iterations in one state
    // (configs[4]: 34 of a line's 43 iterations sit in states whose records are not in LDS, 8 of them enter one -- tools/hop_stats.py;
    // re-reading them came out of L1, keeping them was worth nothing by itself, keeping the hot ones' LDS reads down as well 4 %:
    // 0.856 -> 0.825 ms per 2 M lines).  With every record in LDS (config 3) the plain read per iteration is faster (0.895 against
    // 0.939 ms per 10 M lines): ALL_HOT keeps nothing.
    uint32_t kept = K.kept;
    u32x2 k0 = K.k0, k1 = K.k1, k2 = K.k2;
    for (;;) {
        // (leave_at: the hop slice kernel leaves a round's walk when no more than that many lanes still have bytes -- the others have
        // used up their pieces, and a lane in a long value does so in a third of the iterations a lane in literals needs)
        const uint64_t unfinished = __builtin_amdgcn_ballot_w64(p < limit);
        if (static_cast<uint32_t>(__builtin_popcountll(unfinished)) <= leave_at) {
            if (!ALL_HOT) { K.kept = kept; K.k0 = k0; K.k1 = k1; K.k2 = k2; }
            break;
        }
#ifdef GX_DEV
        ++K.dev_iters;
        K.dev_lane_iters += static_cast<uint32_t>(__builtin_popcountll(unfinished));
#endif
        // ---- 1. the state's record ----
        u32x2 h0, h1, h2;
        if (ALL_HOT) {
            const uint32_t la = __umul24(min(s, last_hot), HOP_REC_B) + HOP_LDS_AT;  // (a lane that is done may sit in any state)
            // Three ds_read_b64, not a ds_read2_b64 and one: the pair is two accesses of four 16-lane groups on 32 banks (8 LDS cycles before
            // any conflict), a ds_read_b64 two 32-lane groups on 64 banks (2).  tools/hop_stats.py's bank model on config 3: 20.4 cycles per
            // iteration for the record as the compiler pairs it, 11.3 as three reads.  The addresses are made opaque so that it cannot pair them.
            uint32_t la1 = la + 8u, la2 = la + 16u;
            asm volatile("" : "+v"(la1), "+v"(la2));
            h0 = lds_ld<u32x2>(la); h1 = lds_ld<u32x2>(la1); h2 = lds_ld<u32x2>(la2);
        } else {
            if ((__builtin_amdgcn_ballot_w64(s != kept) & unfinished) != 0ull) {
                if (s != kept && p < limit) {
                    if (s <= last_hot) {
                        const uint32_t la = __umul24(s, HOP_REC_B) + HOP_LDS_AT;
                        k0 = lds_ld<u32x2>(la); k1 = lds_ld<u32x2>(la + 8u); k2 = lds_ld<u32x2>(la + 16u);   // (as three ds_read_b64, like the ALL_HOT read below: 0.817 against 0.814 ms, configs[4] -- not here)
                    } else {
                        const u32x2* g = reinterpret_cast<const u32x2*>(H.hops + static_cast<uint64_t>(s) * HOP_REC_B);
                        k0 = g[0]; k1 = g[1]; k2 = g[2];
                    }
                    kept = s;
                    // (A lane that ASKS for such a record and sits the iteration out -- the record in other registers, in its hands when the next
                    // iteration begins, the wave not waiting for the trip; round 4's plan -- was built in round 5: 168 registers, no spills,
                    // and 0.972 against 0.815 ms per 3.8 M lines of configs[4], one device: a line enters 8 such states in its 43
                    // iterations, the trip ends in L1, and the iterations sat out cost more than the waits did.  Not kept.)
                    // (asking for the chain target's record one state ahead -- a line's states are a path -- was measured in round 5:
                    // 1.062 against 1.046 ms per 3.8 M lines of configs[4]: the kernel is bound by the instructions it issues, not by that trip)
                }
            }
            h0 = k0; h1 = k1; h2 = k2;   // (a lane that is done may hold any record: nothing of it is used)
        }
        // ---- 2. the run: how many of the next 16 class ids lie in [run_lo, run_hi] ----
        // (The window as three 8-byte aligned ds_read_b64 and a select per dword models at 13.4 instead of 27.8 LDS-array cycles per iteration
        // -- tools/hop_stats.py -- and costs nine vector instructions: 0.771 against 0.758 ms on config 3, one device.  Not kept.)
        // (four dwords, not five: a window that does not begin on a dword is 16 - (p & 3) bytes long -- the bytes behind the fourth dword
        // read as 0xFF, outside every run -- and wlen is what "the run fills its window" means.  One LDS read in eight less per
        // iteration for a long value's windows being 14.5 bytes on average instead of 16.)
        const uint32_t lo4 = splat_byte0(h0.x), k4 = splat_byte1(h0.x);
        uint32_t wlen, n, q;
        auto run_window = [&]() {
            const uint32_t a1 = p & ~3u, sh1 = p & 3u;
            const u32x2 d01 = lds_pair4(a1), d23 = lds_pair4(a1 + 8u);
            const uint32_t x0 = __builtin_amdgcn_alignbyte(d01.y, d01.x, sh1), x1 = __builtin_amdgcn_alignbyte(d23.x, d01.y, sh1);
            const uint32_t x2 = __builtin_amdgcn_alignbyte(d23.y, d23.x, sh1), x3 = __builtin_amdgcn_alignbyte(0xFFFFFFFFu, d23.y, sh1);
            const uint32_t f0 = static_cast<uint32_t>(__ffs(static_cast<int>(run_flags(x0, lo4, k4))) - 1);  // 7, 15, 23, 31 or 0xFFFFFFFF
            const uint32_t f1 = static_cast<uint32_t>(__ffs(static_cast<int>(run_flags(x1, lo4, k4))) - 1);
            const uint32_t f2 = static_cast<uint32_t>(__ffs(static_cast<int>(run_flags(x2, lo4, k4))) - 1);
            const uint32_t f3 = static_cast<uint32_t>(__ffs(static_cast<int>(run_flags(x3, lo4, k4))) - 1);
            n = min3u(f0, sat_add(f1, 32u), min3u(sat_add(f2, 64u), sat_add(f3, 96u), 128u)) >> 3;  // 0 .. 16
            n = min(n, e - p);
            wlen = 16u - sh1;
            q = p + n;
        };
        run_window();
        // (Round 5, after the LDS array turned out to be what binds the walk: lanes whose run fills the window taking their NEXT window right
        // here -- four dword reads and the run test, by those lanes alone, while at least 16 / 24 / 40 lanes do -- instead of in an iteration
        // with its record, chain window and capture stores: 0.967 / 0.929 / 0.915 against 0.743 ms on config 3, one device.  The lanes
        // reach their long values in different iterations, and a group that runs ahead does so while the others wait: the walk's
        // strength is that every iteration serves every lane, whatever it is in the middle of.  Not kept.)
        // (a lane steps when its run ended inside the window and inside the staged bytes; a lane past its limit does not)
        const uint64_t m_step = __builtin_amdgcn_ballot_w64(n < wlen) & __builtin_amdgcn_ballot_w64(q < e) & unfinished;
        // (Round 5: a loop of its own for the iterations in which NO lane steps -- every lane in a run that fills its window: the tile
        // kernel's lanes reach their lines' long last values together -- window and run test alone, no record, chain or capture stores.
        // Measured on config 3, one device: 0.939 against 0.938 ms with captures, 0.652 against 0.608 match only.  The lanes' values end
        // in different windows, so few iterations qualify, and the loop's own tests cost what it saves.  Not kept.)
        // ---- 3. the chain: the 8 bytes at q ----
        // (Picking the chain's bytes out of the run's window with selects instead of a second, dependent LDS read -- an eight-dword window --
        // was measured twice, in round 3 and again in round 4 on the byte-space tables: 7 % SLOWER on config 3, 0.968 against 0.903 ms, dense
        // results, one device, and 2.5 % on config 5.)
        // (a second, dependent LDS read: see above for what picking the bytes out of the first window costs instead)
        // (As two 8-byte aligned ds_read_b64 and a select per dword -- 9.1 instead of 16.4 LDS-array cycles in tools/hop_stats.py's model --
        // the chain's window made config 3 2.4 % SLOWER, 0.756 against 0.739 ms, as the run's window did: not kept.)
        // (read by the lanes whose run ended alone -- fewer lanes, fewer bank conflicts: 0.766 against 0.748 ms, config 3, one device.  Not kept.)
        const uint32_t a2 = q & ~3u, sh2 = q & 3u;
        const u32x2 r01 = lds_pair4(a2);
        const uint32_t r2 = lds_ld<uint32_t>(a2 + 8u);
        const uint32_t v0 = __builtin_amdgcn_alignbyte(r01.y, r01.x, sh2), v1 = __builtin_amdgcn_alignbyte(r2, r01.y, sh2);
        // the single bytes: sum of |v - literal| over the reference bytes that are not 0; the tail: byte `pos` of the eight in [lo, lo + span]
        // (record: h0.x run_lo | run_k << 8 | klen << 16; h0.y target | off1 << 16 | off2 << 24; h1.x column1 * 128 | column2 * 128 << 16;
        //  h1.y tail pos | lo << 8 | span << 16; h2 the single bytes)
        const uint32_t sad = __builtin_amdgcn_msad_u8(v1, h2.y, __builtin_amdgcn_msad_u8(v0, h2.x, 0u));
        const uint32_t tail = sub_b0_b1(__builtin_amdgcn_perm(v1, v0, h1.y), h1.y);   // (the selector's other bytes pick what nobody reads)
        const uint32_t qk = add_byte2(q, h0.x);   // where the chain ends
        const uint32_t span = (h1.y >> 16) & 0xFFu;
        const bool chain_ok = sad == 0u && tail <= span && qk <= e_chain;
        // (the masks of the three compares, ANDed: the ballot of the compound condition was a select and a compare more)
        const uint64_t m_chain = m_step & __builtin_amdgcn_ballot_w64(sad == 0u) & __builtin_amdgcn_ballot_w64(tail <= span) & __builtin_amdgcn_ballot_w64(qk <= e_chain);
        const bool stepping = n < wlen && q < e && p < limit;  // (the same compares, per lane: their masks ARE the select conditions)
        const bool chained = stepping && chain_ok;
        // a lane that does not take its chain reads its record as "no bytes, same state, no programs"
        const uint32_t cols = chained ? h1.x : 0u;
        // ---- capture programs of the chain: register column := position (column 0 is the write-only dummy) ----
        const uint32_t rel = q - p0;  // (the position in the line)
        if (CAPTURE) {
            lds_st<uint16_t>(add_word0(dummy_col, cols), static_cast<uint16_t>(add_byte2(rel, h0.y)));
            lds_st<uint16_t>(add_word1(dummy_col, cols), static_cast<uint16_t>(add_byte3(rel, h0.y)));
        }
        p = p < limit ? (chained ? qk : q) : p;
        s = chained ? (h0.y & 0xFFFFu) : s;
        // ---- 4. one exact step where the chain does not apply ----
        const uint64_t m_exact = m_step & ~m_chain;
        if (m_exact != 0ull) {
            // ---- second chance: the state loops on more than its run interval (\\w: digits, upper case, '_' beside lower case).  The
            // window is tested against the union of the loop set's intervals (up to four); bytes inside it leave the lane where it is,
            // in its state, with no program -- as a run does.  Only lanes whose run ended and whose chain did not apply get here, and
            // only in states that have a loop set: text inside the run intervals never pays for it.  (The flags of an interval are
            // exact up to its own first offender and only ever too many above it, so the AND over the intervals never calls a byte
            // inside that is not.) ----
            bool exact = stepping && !chained;
            // Everything this block may look up is asked for at once -- the tail's set, the loop set, the class of the byte at q: three
            // reads, one trip to LDS.  One behind the other they were three trips before the exact step's row, and in mixed-case text
            // the lanes come here at different times: 21 of a tile's 26 iterations pay for the block (tools/hop_stats.py, GX_MIXED_CASE=1).
            const uint32_t cls_q = lds_ld<uint8_t>(v0 & 0xFFu);
            if (!SECOND) {
                if (__builtin_amdgcn_ballot_w64(exact && ((h0.x | h1.y) >> 24) != 0u) != 0ull) K.hint = 1u;
            } else {
            // (the chain first: its single bytes matched and its tail byte lies in another interval of the exit's bytes -- a value that
            // begins with a digit or an upper-case letter; the intervals stand in ascending order, so no borrow reaches a byte that is
            // not outside by itself: the test is exact)
            const u32x2 ts = lds_ld<u32x2>(H.sets + ((h1.y >> 24) << 3)), ls = lds_ld<u32x2>(H.sets + ((h0.x >> 24) << 3));   // (entry 0: no interval)
            if ((__builtin_amdgcn_ballot_w64(sad == 0u && (h1.y >> 24) != 0u && qk <= e_chain) & m_exact) != 0ull) {
                const bool retry = exact && sad == 0u && qk <= e_chain;
                const uint32_t tb = __builtin_amdgcn_perm(v1, v0, h1.y) & 0xFFu;
                if (retry && run_flags(tb * 0x01010101u, ts.x, ts.y) != HI_BITS) {
                    if (CAPTURE) {
                        lds_st<uint16_t>(add_word0(dummy_col, h1.x), static_cast<uint16_t>(add_byte2(rel, h0.y)));
                        lds_st<uint16_t>(add_word1(dummy_col, h1.x), static_cast<uint16_t>(add_byte3(rel, h0.y)));
                    }
                    p = qk;
                    s = h0.y & 0xFFFFu;
                    exact = false;
                }
            }
            const uint32_t set = h0.x >> 24;
            if (__builtin_amdgcn_ballot_w64(exact && set != 0u) != 0ull) {
                // (the eight bytes at q, where the run ended -- the chain's window, still in registers: the bytes before q lie in the run
                // and so in the union.  Round 5 first tested the sixteen bytes from p again, out of LDS: 100 instructions and three reads
                // where mixed-case text comes through here in 21 of a tile's 26 iterations -- tools/hop_stats.py with GX_MIXED_CASE=1.)
                uint32_t o0 = HI_BITS, o1 = HI_BITS;
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t sel = j * 0x01010101u;
                    const uint32_t l4 = __builtin_amdgcn_perm(ls.x, ls.x, sel), kk4 = __builtin_amdgcn_perm(ls.y, ls.y, sel);
                    o0 &= run_flags(v0, l4, kk4); o1 &= run_flags(v1, l4, kk4);
                }
                const uint32_t g0 = static_cast<uint32_t>(__ffs(static_cast<int>(o0)) - 1), g1 = static_cast<uint32_t>(__ffs(static_cast<int>(o1)) - 1);
                uint32_t nu = min3u(g0, sat_add(g1, 32u), 64u) >> 3;   // 0 .. 8 bytes at q inside the union
                nu = min(nu, e - q);
                if (exact && nu != 0u) { p = q + nu; exact = false; }
            }
            if (__builtin_amdgcn_ballot_w64(stepping && !chained && !exact) != 0ull) K.hint = 1u;   // (a second chance applied)
            }
            if (exact) {
                // a state without a chain keeps the LDS address / 4 of its dense row's copy in the target field (0: none)
                exact_step_c(((h0.x >> 16) & 0xFFu) == 0u ? (h0.y & 0xFFFFu) << 2 : 0u, cls_q, q);
                if (ALL_HOT)
                    while (s > last_hot && p < limit) exact_step(0u, lds_ld<uint8_t>(p), p);   // (off the expected path: rare, and short)
            }
        }
    }
    return s;
}

// The tile kernel's case: the whole line [start, end) is staged.
template <bool CAPTURE>
__device__ __forceinline__ uint32_t walk_hop(const HopTab& H, bool all_hot, uint32_t stage, uint32_t s, uint32_t start, uint32_t end, bool on,
                                             uint32_t dead, uint32_t regs, bool& second) {
    const uint32_t p0 = stage + start, e = stage + end;
    uint32_t p = on ? p0 : e;
    HopKept K;
    uint32_t r;
    if (all_hot) r = second ? walk_hop_span<true, CAPTURE, true>(H, p, e, e, e, p0, s, dead, regs, 0u, K) : walk_hop_span<true, CAPTURE, false>(H, p, e, e, e, p0, s, dead, regs, 0u, K);
    else r = walk_hop_span<false, CAPTURE, true>(H, p, e, e, e, p0, s, dead, regs, 0u, K);
    second = K.hint != 0u;   // (wave-uniform: the next tile's mode)
    return r;
}

}  // namespace gx
