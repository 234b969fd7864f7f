// gx_kernels.hip -- gfx950 kernels for the Gorp match-and-extract hot path other than the tile kernel
// (gx_tile.hip): the per-line kernel (any table size, any line length, UTF-16) and the slice kernel (long lines).
//
// Replaces, per line (lane-per-line, no cross-line state):
//   hot loop #1  PolyMatcher.match      core/autom/PolyMatcher.java:123-133
//                Automata.step/accept   core/autom/Automata.java:133-139
//   hot loop #2  JDKRegexpCookedExtraction.match/_constructMatch
//                                       core/jdkre/JDKRegexpCookedExtraction.java:36-59
//   driver       Gorp.extract           core/Gorp.java:159-186
#include "gx_walk.hpp"
#include "gx_hop_dev.hpp"

namespace gx {

namespace {

constexpr int GENERIC_MAX_REGS = 96;  // must cover GxDev::max_regs (checked on the host)

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

// Where one line's result goes: dense (match_id[n] + caps[n][slots], int32) or compact rows
// (u16[1 + slots] per line: int16 id, then offsets with 0xFFFF = unset; an offset above 65534 is stored as 65534
// and counted in *overflow, and the caller then takes that batch in the dense format).
struct LineOut {
    int32_t* match_id;
    int32_t* caps;
    uint16_t* packed;
    unsigned long long* overflow;
    int slots;
    int narrow;  // `packed` holds u8 rows (int8 id, offsets with 0xFF = unset, above 254: stored as 254 and counted)
    __device__ __forceinline__ void id(uint64_t i, int32_t k) const {
        if (packed && narrow) reinterpret_cast<uint8_t*>(packed)[i * static_cast<uint64_t>(1 + slots)] = static_cast<uint8_t>(k);
        else if (packed) packed[i * static_cast<uint64_t>(1 + slots)] = static_cast<uint16_t>(k);
        else match_id[i] = k;
    }
    __device__ __forceinline__ void cap(uint64_t i, int t, int32_t v) const {
        if (packed && narrow) {
            uint8_t w = 0xFFu;
            if (v >= 0) {
                if (v > 254) { v = 254; if (overflow) atomicAdd(overflow, 1ull); }
                w = static_cast<uint8_t>(v);
            }
            reinterpret_cast<uint8_t*>(packed)[i * static_cast<uint64_t>(1 + slots) + 1 + t] = w;
        } else if (packed) {
            uint16_t w = 0xFFFFu;
            if (v >= 0) {
                if (v > 65534) { v = 65534; if (overflow) atomicAdd(overflow, 1ull); }
                w = static_cast<uint16_t>(v);
            }
            packed[i * static_cast<uint64_t>(1 + slots) + 1 + t] = w;
        } else caps[i * static_cast<uint64_t>(slots) + t] = v;
    }
};

// A line's whole result row from its lane in as few stores as the format allows (the slice kernels: a lane holds a line of its
// own, its row goes where no neighbour's does -- 1 + 2 G scattered stores of two bytes each were a sixth of the hop slice kernel on
// BASELINE configs[4]).  Four groups at a time, as the final records hold their tags: dense results leave as two 16-byte stores,
// u16 rows as one (the row's first halfword is the id, so a word is one group's end and the next group's begin: `carry`), u8 rows
// as one of 8 bytes; global memory takes them at any alignment.  Returns nothing: the id is the row's (or match_id's) to keep.
struct __attribute__((packed)) UnalignedU32x4 { u32x4 v; };
struct __attribute__((packed)) UnalignedU32x2 { u32x2 v; };
struct __attribute__((packed)) UnalignedU32 { uint32_t v; };
struct __attribute__((packed)) UnalignedU16 { uint16_t v; };

template <int TIER>
__device__ __forceinline__ void write_row(const LineOut& out, uint64_t i, int32_t info, uint32_t fin_lds, const uint8_t* fin_g, uint32_t regs,
                                          uint32_t len, int G, uint32_t hop_unset = 0u) {
    const uint32_t rec = info >= 0 ? static_cast<uint32_t>(info) : 0u;
    const uint32_t dummy_col = regs - 128u;
    const uint32_t id_at = rec + 16u * static_cast<uint32_t>((G + 3) >> 2);
    const bool FIN_GLOBAL = TIER == TIER_L2 || TIER == TIER_RECG || (TIER == TIER_HOP && fin_g != nullptr);
    uint32_t idw;
    if (FIN_GLOBAL) idw = *reinterpret_cast<const uint16_t*>(fin_g + id_at);
    else idw = lds_ld<uint16_t>(fin_lds + id_at);
    const int32_t mid = info >= 0 ? static_cast<int32_t>(static_cast<int16_t>(idw)) : info;
    const uint32_t unit = out.packed ? (out.narrow ? 1u : 2u) : 4u;
    uint8_t* row = out.packed ? reinterpret_cast<uint8_t*>(out.packed) + i * static_cast<uint64_t>(1 + out.slots) * unit
                              : reinterpret_cast<uint8_t*>(out.caps + i * static_cast<uint64_t>(out.slots));
    if (!out.packed) out.match_id[i] = mid;
    uint32_t carry = static_cast<uint32_t>(mid) & (out.narrow ? 0xFFu : 0xFFFFu);
    uint32_t clipped = 0u;
    for (int g0 = 0; g0 < G; g0 += 4) {
        u32x4 t;
        if (FIN_GLOBAL) t = *reinterpret_cast<const u32x4*>(fin_g + rec + 4u * g0);
        else t = lds_ld<u32x4>(fin_lds + rec + 4u * g0);
        const uint32_t tw[4] = {t.x, t.y, t.z, t.w};
        uint32_t vb[4], ve[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vb[q] = lds_ld<uint16_t>(dummy_col + (tw[q] & (TIER == TIER_HOP ? 0xFFFFu : 0xFF80u)));
            ve[q] = lds_ld<uint16_t>(dummy_col + (TIER == TIER_HOP ? tw[q] >> 16 : (tw[q] >> 16) & 0xFF80u));
        }
        int32_t pb[4], pe[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t tb = tw[q] & 0xFFFFu, te = tw[q] >> 16;
            if (TIER == TIER_HOP) {   // (tags name columns, "the length" and "unset" among them: gx_hop.cpp)
                const bool unset = tb == hop_unset || info < 0;
                pb[q] = unset ? -1 : static_cast<int32_t>(vb[q]);
                pe[q] = unset ? -1 : static_cast<int32_t>(ve[q]);
            } else {
                pb[q] = tb == 1u ? static_cast<int32_t>(len) : static_cast<int32_t>(vb[q]);
                pe[q] = te == 1u ? static_cast<int32_t>(len) : static_cast<int32_t>(ve[q]);
                if (tb == 0u || te == 0u || info < 0) { pb[q] = -1; pe[q] = -1; }
            }
        }
        const int cnt = G - g0 < 4 ? G - g0 : 4;
        if (!out.packed) {
            uint8_t* dst = row + 8u * g0;
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x4*>(dst)->v = u32x4{static_cast<uint32_t>(pb[0]), static_cast<uint32_t>(pe[0]), static_cast<uint32_t>(pb[1]), static_cast<uint32_t>(pe[1])};
                reinterpret_cast<UnalignedU32x4*>(dst + 16)->v = u32x4{static_cast<uint32_t>(pb[2]), static_cast<uint32_t>(pe[2]), static_cast<uint32_t>(pb[3]), static_cast<uint32_t>(pe[3])};
            } else {
                for (int q = 0; q < cnt; ++q) reinterpret_cast<UnalignedU32x2*>(dst + 8 * q)->v = u32x2{static_cast<uint32_t>(pb[q]), static_cast<uint32_t>(pe[q])};
            }
        } else if (!out.narrow) {
            uint32_t hb[4], he[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                clipped += (pb[q] > 65534 ? 1u : 0u) + (pe[q] > 65534 ? 1u : 0u);
                hb[q] = pb[q] < 0 ? 0xFFFFu : static_cast<uint32_t>(min(pb[q], 65534));
                he[q] = pe[q] < 0 ? 0xFFFFu : static_cast<uint32_t>(min(pe[q], 65534));
            }
            uint8_t* dst = row + 4u * g0;   // (the halfword before group g0's begin: the id, or the end of the group before)
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x4*>(dst)->v = u32x4{carry | hb[0] << 16, he[0] | hb[1] << 16, he[1] | hb[2] << 16, he[2] | hb[3] << 16};
                carry = he[3];
            } else {
                for (int q = 0; q < cnt; ++q) {
                    reinterpret_cast<UnalignedU32*>(dst + 4 * q)->v = carry | hb[q] << 16;
                    carry = he[q];
                }
            }
        } else {
            uint32_t hb[4], he[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                clipped += (pb[q] > 254 ? 1u : 0u) + (pe[q] > 254 ? 1u : 0u);
                hb[q] = pb[q] < 0 ? 0xFFu : static_cast<uint32_t>(min(pb[q], 254));
                he[q] = pe[q] < 0 ? 0xFFu : static_cast<uint32_t>(min(pe[q], 254));
            }
            uint8_t* dst = row + 2u * g0;
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x2*>(dst)->v = u32x2{carry | hb[0] << 8 | he[0] << 16 | hb[1] << 24, he[1] | hb[2] << 8 | he[2] << 16 | hb[3] << 24};
                carry = he[3];
            } else {
                for (int q = 0; q < cnt; ++q) {
                    reinterpret_cast<UnalignedU16*>(dst + 2 * q)->v = static_cast<uint16_t>(carry | hb[q] << 8);
                    carry = he[q];
                }
            }
        }
    }
    if (out.packed) {
        // the row's last unit; slots beyond the definition's groups do not exist (slots == 2 * max_groups == 2 G)
        if (out.narrow) row[2 * G] = static_cast<uint8_t>(carry);
        else reinterpret_cast<UnalignedU16*>(row + 4 * G)->v = static_cast<uint16_t>(carry);
        if (clipped && out.overflow) atomicAdd(out.overflow, static_cast<unsigned long long>(clipped));
    }
}

// The same from the final record OF THE STATE (gx_hop.cpp: fin_state_off): `recp` = its tags (u16 begin, end per group, padded to
// four groups) and behind them the extraction's index, or -1 / -2-k for a state that accepts nothing.  Everything the row needs
// from global memory is one cache line, read at once (the first twelve groups' tags and the index before anything waits).
// (the first twelve groups' tags and the index of a final record, as a lane asks for them when its line is over: FinAhead)
struct FinAhead {
    u32x4 t0 = {0u, 0u, 0u, 0u}, t1 = {0u, 0u, 0u, 0u}, t2 = {0u, 0u, 0u, 0u};
    uint32_t id = 0u;
    __device__ __forceinline__ void load(const uint8_t* __restrict__ recp, int G) {
        const int nblk = (G + 3) >> 2;
        t0 = *reinterpret_cast<const u32x4*>(recp);
        t1 = *reinterpret_cast<const u32x4*>(recp + (nblk > 1 ? 16 : 0));
        t2 = *reinterpret_cast<const u32x4*>(recp + (nblk > 2 ? 32 : 0));
        id = *reinterpret_cast<const uint16_t*>(recp + 16 * nblk);
    }
};
__device__ __forceinline__ void write_row_rec(const LineOut& out, uint64_t i, const uint8_t* __restrict__ recp, const FinAhead& F, uint32_t regs, uint32_t len, int G,
                                              uint32_t hop_unset) {
    const uint32_t dummy_col = regs - 128u;
    const u32x4 pre[3] = {F.t0, F.t1, F.t2};
    const int32_t mid = static_cast<int16_t>(F.id);
    const uint32_t unit = out.packed ? (out.narrow ? 1u : 2u) : 4u;
    uint8_t* row = out.packed ? reinterpret_cast<uint8_t*>(out.packed) + i * static_cast<uint64_t>(1 + out.slots) * unit
                              : reinterpret_cast<uint8_t*>(out.caps + i * static_cast<uint64_t>(out.slots));
    if (!out.packed) out.match_id[i] = mid;
    uint32_t carry = static_cast<uint32_t>(mid) & (out.narrow ? 0xFFu : 0xFFFFu);
    uint32_t clipped = 0u;
    const bool may_clip = __builtin_amdgcn_ballot_w64(len > 65534u) != 0ull;   // (wave-uniform: of the lanes that are here)
    for (int g0 = 0; g0 < G; g0 += 4) {
        u32x4 t = g0 == 0 ? pre[0] : g0 == 4 ? pre[1] : pre[2];
        if (g0 >= 12) t = *reinterpret_cast<const u32x4*>(recp + 4 * g0);
        const uint32_t tw[4] = {t.x, t.y, t.z, t.w};
        uint32_t vb[4], ve[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vb[q] = lds_ld<uint16_t>(dummy_col + (tw[q] & 0xFFFFu));
            ve[q] = lds_ld<uint16_t>(dummy_col + (tw[q] >> 16));
        }
        int32_t pb[4], pe[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool unset = (tw[q] & 0xFFFFu) == hop_unset;   // (a state that accepts nothing has every tag unset)
            pb[q] = unset ? -1 : static_cast<int32_t>(vb[q]);
            pe[q] = unset ? -1 : static_cast<int32_t>(ve[q]);
        }
        const int cnt = G - g0 < 4 ? G - g0 : 4;
        if (!out.packed) {
            uint8_t* dst = row + 8u * g0;
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x4*>(dst)->v = u32x4{static_cast<uint32_t>(pb[0]), static_cast<uint32_t>(pe[0]), static_cast<uint32_t>(pb[1]), static_cast<uint32_t>(pe[1])};
                reinterpret_cast<UnalignedU32x4*>(dst + 16)->v = u32x4{static_cast<uint32_t>(pb[2]), static_cast<uint32_t>(pe[2]), static_cast<uint32_t>(pb[3]), static_cast<uint32_t>(pe[3])};
            } else {
                for (int q = 0; q < cnt; ++q) reinterpret_cast<UnalignedU32x2*>(dst + 8 * q)->v = u32x2{static_cast<uint32_t>(pb[q]), static_cast<uint32_t>(pe[q])};
            }
        } else if (!out.narrow) {
            uint32_t hb[4], he[4];
            if (!may_clip) {   // (no line of this service reaches the 65 535th byte: -1 is 0xFFFF, everything else fits; config 5: 1 %)
#pragma unroll
                for (int q = 0; q < 4; ++q) { hb[q] = static_cast<uint32_t>(pb[q]) & 0xFFFFu; he[q] = static_cast<uint32_t>(pe[q]) & 0xFFFFu; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    clipped += (pb[q] > 65534 ? 1u : 0u) + (pe[q] > 65534 ? 1u : 0u);
                    hb[q] = pb[q] < 0 ? 0xFFFFu : static_cast<uint32_t>(min(pb[q], 65534));
                    he[q] = pe[q] < 0 ? 0xFFFFu : static_cast<uint32_t>(min(pe[q], 65534));
                }
            }
            uint8_t* dst = row + 4u * g0;   // (the halfword before group g0's begin: the id, or the end of the group before)
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x4*>(dst)->v = u32x4{carry | hb[0] << 16, he[0] | hb[1] << 16, he[1] | hb[2] << 16, he[2] | hb[3] << 16};
                carry = he[3];
            } else {
                for (int q = 0; q < cnt; ++q) {
                    reinterpret_cast<UnalignedU32*>(dst + 4 * q)->v = carry | hb[q] << 16;
                    carry = he[q];
                }
            }
        } else {
            uint32_t hb[4], he[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                clipped += (pb[q] > 254 ? 1u : 0u) + (pe[q] > 254 ? 1u : 0u);
                hb[q] = pb[q] < 0 ? 0xFFu : static_cast<uint32_t>(min(pb[q], 254));
                he[q] = pe[q] < 0 ? 0xFFu : static_cast<uint32_t>(min(pe[q], 254));
            }
            uint8_t* dst = row + 2u * g0;
            if (cnt == 4) {
                reinterpret_cast<UnalignedU32x2*>(dst)->v = u32x2{carry | hb[0] << 8 | he[0] << 16 | hb[1] << 24, he[1] | hb[2] << 8 | he[2] << 16 | hb[3] << 24};
                carry = he[3];
            } else {
                for (int q = 0; q < cnt; ++q) {
                    reinterpret_cast<UnalignedU16*>(dst + 2 * q)->v = static_cast<uint16_t>(carry | hb[q] << 8);
                    carry = he[q];
                }
            }
        }
    }
    if (out.packed) {
        if (out.narrow) row[2 * G] = static_cast<uint8_t>(carry);
        else reinterpret_cast<UnalignedU16*>(row + 4 * G)->v = static_cast<uint16_t>(carry);
        if (clipped && out.overflow) atomicAdd(out.overflow, static_cast<unsigned long long>(clipped));
    }
}

// ---------------------------------------------------------------------------
// An extraction whose capture automaton is not built ahead of time (2^n register patterns: gx_compile.hpp, RuleTables::pike):
// its prioritised Thompson program run as it is -- a Pike VM.  Thread lists in priority order, a thread = (pc, its group
// boundaries); a step moves every thread whose CHAR takes the code unit's class and closes over SPLIT (preferred branch first) /
// JMP / TAG (:= position), first arrival at a pc wins.  Matcher.matches() (core/jdkre/JDKRegexpCookedExtraction.java:36-39) is
// the first thread, in that order, that stands on MATCH when the line is over: the path backtracking would have found first.
// Linear in line x program, exact, slow: the fallback that lets gx_create_* accept what Pattern.compile accepts.
// scratch: this lane's ints (GxDev::pike_lane_ints): clist | nlist [n_inst][1 + 2 G] each, stack [3][2 n_inst + 2], mark [n_inst], caps [2 G].
// ---------------------------------------------------------------------------
template <typename CH>
__device__ bool pike_capture(const GxDev& T, int k, const CH* __restrict__ s, int64_t len, int32_t* __restrict__ scratch, int32_t* __restrict__ caps_out) {
    const uint32_t base = T.pike_off[k], n_inst = T.pike_off[k + 1] - base;
    const uint32_t* code = T.pike_code + 2ull * base;
    const int ng = T.c_ngroups[k];
    const uint32_t W = 1u + 2u * static_cast<uint32_t>(ng);
    int32_t* clist = scratch;
    int32_t* nlist = clist + static_cast<size_t>(n_inst) * W;
    int32_t* stack = nlist + static_cast<size_t>(n_inst) * W;   // entries of three ints: pc (or -1: restore), slot, value
    int32_t* mark = stack + 3u * (2u * n_inst + 2u);
    int32_t* caps = mark + n_inst;
    for (uint32_t q = 0; q < n_inst; ++q) mark[q] = 0;
    for (uint32_t t = 0; t < 2u * static_cast<uint32_t>(ng); ++t) caps[t] = -1;
    int32_t gen = 1;
    uint32_t ccount = 0, ncount = 0;
    // the closure of pc0 with the boundaries in `caps`, appended to list[0 .. count) in priority order
    auto add_thread = [&](int32_t* list, uint32_t& count, uint32_t pc0, int32_t pos) {
        uint32_t sp = 0;
        stack[0] = static_cast<int32_t>(pc0); stack[1] = 0; stack[2] = 0; sp = 1;
        while (sp) {
            --sp;
            const int32_t pc = stack[3 * sp];
            if (pc < 0) { caps[stack[3 * sp + 1]] = stack[3 * sp + 2]; continue; }   // (a TAG's branch is done: its boundary as it was)
            if (mark[pc] == gen) continue;
            mark[pc] = gen;
            const uint32_t w0 = code[2 * pc], op = w0 & 0xFFu, x = w0 >> 8, y = code[2 * pc + 1];
            if (op == 1u) {           // SPLIT: x first
                stack[3 * sp] = static_cast<int32_t>(y); ++sp;
                stack[3 * sp] = static_cast<int32_t>(x); ++sp;
            } else if (op == 2u) {    // JMP
                stack[3 * sp] = static_cast<int32_t>(x); ++sp;
            } else if (op == 3u) {    // TAG
                stack[3 * sp] = -1; stack[3 * sp + 1] = static_cast<int32_t>(x); stack[3 * sp + 2] = caps[x]; ++sp;
                caps[x] = pos;
                stack[3 * sp] = pc + 1; ++sp;
            } else if (op == 0u || op == 4u) {   // CHAR / MATCH: a thread
                int32_t* t = list + static_cast<size_t>(count) * W;
                t[0] = pc;
                for (uint32_t q = 1; q < W; ++q) t[q] = caps[q - 1];
                ++count;
            }
        }
    };
    add_thread(clist, ccount, 0u, 0);
    for (int64_t p = 0; p < len && ccount; ++p) {
        const uint32_t c = static_cast<uint32_t>(class_of(T, s[p]));
        ++gen;
        ncount = 0;
        for (uint32_t i = 0; i < ccount; ++i) {
            const int32_t* t = clist + static_cast<size_t>(i) * W;
            const uint32_t w0 = code[2 * t[0]];
            if ((w0 & 0xFFu) != 0u) continue;   // (a thread on MATCH with input left: over)
            if (!((T.pike_sets[8u * (w0 >> 8) + (c >> 5)] >> (c & 31u)) & 1u)) continue;
            for (uint32_t q = 1; q < W; ++q) caps[q - 1] = t[q];
            add_thread(nlist, ncount, static_cast<uint32_t>(t[0]) + 1u, static_cast<int32_t>(p + 1));
        }
        int32_t* sw = clist; clist = nlist; nlist = sw;
        ccount = ncount;
    }
    for (uint32_t i = 0; i < ccount; ++i) {
        const int32_t* t = clist + static_cast<size_t>(i) * W;
        if ((code[2 * t[0]] & 0xFFu) != 4u) continue;
        for (uint32_t q = 1; q < W; ++q) caps_out[q - 1] = t[q];
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------
// One line, tables read through L1/L2 (any table size, any line length).
// ---------------------------------------------------------------------------
// pike_slot: the lane's area in GxDev::pike_scratch (0xFFFFFFFF: its index in the grid -- the launchers keep the grids of a handle
// that has such extractions within GxDev::pike_blocks workgroups)
template <typename CH, typename MS>
__device__ void extract_line_global(const GxDev& T, const MS* __restrict__ m_next, const CH* __restrict__ s, int64_t len,
                                    uint64_t i, const LineOut& out, int32_t* __restrict__ state_out, int match_only, uint32_t pike_slot = 0xFFFFFFFFu) {
    const int ncls = T.ncls;
    const int slots = out.slots;
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
        if (match_only || !T.has_capture) { out.id(i, k); return; }
    }

    if (k < 0) {
        out.id(i, -1);
        for (int t = 0; t < slots; ++t) out.cap(i, t, -1);
        return;
    }
    if (T.pike_off && T.pike_off[k + 1] != T.pike_off[k]) {
        // ---- hot loop #2 for an extraction without a capture automaton: its program, run as it is ----
        const uint32_t lane_slot = pike_slot == 0xFFFFFFFFu ? blockIdx.x * blockDim.x + threadIdx.x : pike_slot;
        int32_t* scratch = T.pike_scratch + static_cast<size_t>(lane_slot) * T.pike_lane_ints;
        int32_t* got = scratch + T.pike_lane_ints - 2u * static_cast<uint32_t>(T.max_groups) - 2u;   // (the area's last ints)
        if (!pike_capture<CH>(T, k, s, len, scratch, got)) {
            out.id(i, capture_only ? -1 : -2 - k);
            for (int t = 0; t < slots; ++t) out.cap(i, t, -1);
            return;
        }
        const int ng = T.c_ngroups[k];
        for (int g = 0; g < ng; ++g) {
            int32_t pb = got[2 * g], pe = got[2 * g + 1];
            if (pb < 0 || pe < 0) { pb = -1; pe = -1; }
            out.cap(i, 2 * g, pb);
            out.cap(i, 2 * g + 1, pe);
        }
        for (int t = 2 * ng; t < slots; ++t) out.cap(i, t, -1);
        out.id(i, k);
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
        out.id(i, capture_only ? -1 : -2 - k);
        for (int t = 0; t < slots; ++t) out.cap(i, t, -1);
        return;
    }
    const int ng = T.c_ngroups[k];
    for (int g = 0; g < ng; ++g) {
        const uint16_t vb = T.fin_tags[f + 2 * g], ve = T.fin_tags[f + 2 * g + 1];
        int32_t pb = (vb == SRC_POS) ? static_cast<int32_t>(len) : (vb == SRC_NIL ? -1 : regs[vb]);
        int32_t pe = (ve == SRC_POS) ? static_cast<int32_t>(len) : (ve == SRC_NIL ? -1 : regs[ve]);
        if (pb < 0 || pe < 0) { pb = -1; pe = -1; }
        out.cap(i, 2 * g, pb);
        out.cap(i, 2 * g + 1, pe);
    }
    for (int t = 2 * ng; t < slots; ++t) out.cap(i, t, -1);
    out.id(i, k);
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
k_extract_generic(GxDev T, const CH* __restrict__ data, const OFF* __restrict__ off, uint64_t n, LineOut out,
                  int32_t* __restrict__ state_out, int match_only, const MS* __restrict__ m_next, int strip_eol) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = off[i], e = off[i + 1];
        int64_t len = static_cast<int64_t>(e - b);
        if (strip_eol) len = trim_eol(data + b, len);
        extract_line_global<CH, MS>(T, m_next, data + b, len, i, out, state_out, match_only);
    }
}

// ---- UTF-16 batches (gx_batch_opts.utf16: the code units of Java Strings) ----
// Log text is Latin-1 almost always, and then a code unit IS the byte the batch kernels walk.  k_narrow_units copies the
// units' low bytes into a byte buffer (16 bytes in, 8 out per lane) and flags the line of every unit above 0xFF; the byte
// kernels then run on the copy with the same offsets, and k_extract_flagged takes the flagged lines again, on the code
// units, with the per-line kernel's walk (classes of units above 0xFF by binary search).
template <typename OFF>
__global__ void __launch_bounds__(256)
k_narrow_units(const uint16_t* __restrict__ units, const OFF* __restrict__ off, uint64_t n, uint8_t* __restrict__ bytes, uint8_t* __restrict__ flags) {
    const uint64_t first = off[0], total = static_cast<uint64_t>(off[n]) - first;
    const uint16_t* src = units + first;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x * 8u;
    for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 8u; i < total; i += stride) {
        uint32_t wide = 0;
        if (total - i >= 8u) {
            // eight units in one (unaligned) 16-byte load, their low bytes in one aligned 8-byte store
            const u32x4 v = reinterpret_cast<const UnalignedWindow*>(src + i)->v;
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (d[q] & 0xFF00FF00u) wide |= ((d[q] & 0x0000FF00u) ? 1u : 0u) << (2 * q) | ((d[q] & 0xFF000000u) ? 2u : 0u) << (2 * q);
                const uint32_t two = (d[q] & 0xFFu) | ((d[q] >> 8) & 0xFF00u);
                if (q < 2) lo |= two << (16 * q); else hi |= two << (16 * (q - 2));
            }
            *reinterpret_cast<u32x2*>(bytes + i) = u32x2{lo, hi};
        } else {
            for (uint32_t q = 0; q < static_cast<uint32_t>(total - i); ++q) {
                const uint16_t u = src[i + q];
                bytes[i + q] = static_cast<uint8_t>(u);
                if (u > 0xFFu) wide |= 1u << q;
            }
        }
        while (wide) {   // (rare) the line of unit first + i + q: the last offset at or below it
            const uint32_t q = static_cast<uint32_t>(__ffs(static_cast<int>(wide)) - 1);
            wide &= wide - 1u;
            const uint64_t pos = first + i + q;
            uint64_t lo = 0, hi = n;   // off[lo] <= pos < off[hi]
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (static_cast<uint64_t>(off[mid]) <= pos) lo = mid; else hi = mid; }
            flags[lo] = 1;
        }
    }
}
template <typename OFF, typename MS>
__global__ void __launch_bounds__(256)
k_extract_flagged(GxDev T, const uint16_t* __restrict__ units, const OFF* __restrict__ off, uint64_t n, const uint8_t* __restrict__ flags,
                  LineOut out, int match_only, const MS* __restrict__ m_next, int strip_eol, const uint32_t* __restrict__ any_word, uint32_t seq) {
    if (any_word && __hip_atomic_load(any_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) return;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!flags[i]) continue;
        const uint64_t b = off[i], e = off[i + 1];
        int64_t len = static_cast<int64_t>(e - b);
        if (strip_eol) len = trim_eol(units + b, len);
        extract_line_global<uint16_t, MS>(T, m_next, units + b, len, i, out, nullptr, match_only);
    }
}

// Follow-up of the tile kernel: the lines it could not stage (one line longer than its staging area, rare) are
// taken here, one lane per line.  The tile kernel announces that there are any by storing the launch's sequence
// number into *flag; without it every wave leaves at once.  The predicate is the batch kernel's own: the tile kernel's
// make_round (the line, as it lies in memory, against the staging area: by_length 0), or the lane and hop slice kernels' "longer
// than the 16-bit positions hold" on the line WITHOUT its terminator (by_length 1) -- never a line the batch kernel has
// already answered (it would be counted twice in *overflow).
template <typename OFF, typename CH>
__global__ void __launch_bounds__(256)
k_extract_oversize(GxDev T, const CH* __restrict__ data, const OFF* __restrict__ off, uint64_t n, LineOut out, int match_only,
                   int strip_eol, const uint32_t* __restrict__ flag, uint32_t seq, uint32_t limit, int by_length, int32_t* __restrict__ state_out) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) return;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = off[i], e = off[i + 1];
        int64_t len = static_cast<int64_t>(e - b);
        if (!by_length) {
            // (code units: a staged chunk is 32 bytes of the buffer, and the skew counts in units)
            const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(data + b) & (sizeof(CH) == 2 ? 31u : 15u)) / static_cast<uint32_t>(sizeof(CH));
            if ((e - b) + skew + 48u <= limit) continue;  // the tile kernel staged this one
            if (strip_eol) len = trim_eol(data + b, len);
        } else {
            if (e - b <= limit) continue;                // (no need to look at its last bytes)
            if (strip_eol) len = trim_eol(data + b, len);
            if (len <= static_cast<int64_t>(limit)) continue;
        }
        if (T.m_next16) extract_line_global<CH, uint16_t>(T, T.m_next16, data + b, len, i, out, state_out, match_only);
        else extract_line_global<CH, uint32_t>(T, T.m_next32, data + b, len, i, out, state_out, match_only);
    }
}

LineOut line_out(const GxDev& dev, const GxBatch& b) {
    LineOut o;
    o.match_id = b.match_id;
    o.caps = b.caps;
    o.packed = b.packed;
    o.overflow = b.overflow;
    o.narrow = b.narrow;
    o.slots = 2 * dev.max_groups;
    return o;
}

template <typename CH, typename OFF>
hipError_t launch_generic_t(const GxDev& dev, const GxBatch& b, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    const int block = 256;
    uint64_t blocks = (b.n + block - 1) / block;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    if (dev.pike_off && blocks > dev.pike_blocks) blocks = dev.pike_blocks;   // (every lane that may run a program has its own thread lists)
    dim3 grid(static_cast<unsigned>(blocks));
    if (dev.m_next16)
        hipLaunchKernelGGL((k_extract_generic<CH, OFF, uint16_t>), grid, dim3(block), 0, stream, dev,
                           static_cast<const CH*>(b.data), static_cast<const OFF*>(b.offsets), b.n, line_out(dev, b),
                           b.state_out, b.match_only, dev.m_next16, b.strip_eol);
    else
        hipLaunchKernelGGL((k_extract_generic<CH, OFF, uint32_t>), grid, dim3(block), 0, stream, dev,
                           static_cast<const CH*>(b.data), static_cast<const OFF*>(b.offsets), b.n, line_out(dev, b),
                           b.state_out, b.match_only, dev.m_next32, b.strip_eol);
    return hipGetLastError();
}

// One line (gx_extract_one_utf16 / gx_match_one_utf16 / the capture-alone call): the call's latency is what counts.  The code units
// and the result words live in PINNED HOST memory that the device reads and writes directly -- no copy commands either side of the
// kernel -- so the wave first brings the line into LDS with one coalesced sweep (a walk straight out of host memory would cross the
// bus once per character), then lane 0 walks it there.
constexpr uint32_t ONE_LINE_MAX_UNITS = 16384;   // 32 KB of LDS; longer lines take the batch path's kernel on a device copy
template <typename MS>
__global__ void __launch_bounds__(64)
k_extract_one(GxDev T, const uint16_t* __restrict__ units, uint32_t len, LineOut out, int32_t* __restrict__ state_out, int match_only,
              const MS* __restrict__ m_next) {
    __shared__ uint16_t line[ONE_LINE_MAX_UNITS];
    const uint32_t pairs = (len + 1u) >> 1;   // (the buffer is padded to a multiple of 8 bytes)
    for (uint32_t q = threadIdx.x; q < pairs; q += 64u) reinterpret_cast<uint32_t*>(line)[q] = reinterpret_cast<const uint32_t*>(units)[q];
    __syncthreads();
    if (threadIdx.x == 0) extract_line_global<uint16_t, MS>(T, m_next, line, static_cast<int64_t>(len), 0, out, state_out, match_only, 256u * T.pike_blocks);
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

template <typename OFF, int TIER>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_extract_slices(GxDev T, GxLds L, const uint8_t* __restrict__ lds_image, const uint8_t* __restrict__ at_global,
                 const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n, LineOut out, int match_only, int strip_eol) {
    {
        extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];
        const uint4* src = reinterpret_cast<const uint4*>(lds_image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
    }
    __syncthreads();
    constexpr bool GT = TIER == TIER_L2 || TIER == TIER_RECG;
    const bool want_caps = match_only == 0 && T.has_capture;
    WalkTab W;  // the automaton this launch walks
    W.at = GT ? (want_caps || TIER == TIER_RECG ? at_global + L.c_base : at_global) : nullptr;
    W.row_bytes = L.row_bytes;
    W.ops_off = L.ops_off;
    W.ops = L.ops;
    W.acc_tab = L.acc_tab;
    W.ncls = L.ncls;
    W.indexed = L.rec_indexed;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t slice = L.stage + wave * L.stage_bytes;
    const uint32_t regs = L.regs + wave * L.regs_wave_bytes + 128u + lane * 2u;
    const uint8_t* data_end = data + static_cast<uint64_t>(off[n]);
    const uint32_t row0 = want_caps ? L.u_start : L.m_start, dead_row = want_caps ? L.u_dead : L.m_dead;
    const uint8_t* fin_g = GT ? at_global + L.fin_tags : nullptr;

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
    const uint32_t my = slice + lane * SLICE_ROW;

    for (;;) {
        // ---- lanes whose line is finished write its result -- not every time one finishes: lanes run in lock step,
        // so this block (and the refill below) costs every lane of the wave its full length whenever a single lane
        // enters it.  Finished lanes wait until a quarter of the wave is idle (or nothing is left to walk). ----
        const bool finished = has_line && (pos >= len || row == dead_row);
        const uint32_t idle = static_cast<uint32_t>(__popcll(__ballot(finished || !has_line)));
        const bool service = idle >= 16u || !__any(has_line && !finished);
        const bool done = service && finished;
        if (done) {
            const int32_t info = state_info<TIER>(W, row);
            if (!want_caps) out.id(i, info);
            else write_row<TIER>(out, i, info, L.fin_tags, fin_g, regs, len, T.max_groups);
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
                    if (T.m_next16) extract_line_global<uint8_t, uint16_t>(T, T.m_next16, data + o0, len64, i, out, nullptr, match_only);
                    else extract_line_global<uint8_t, uint32_t>(T, T.m_next32, data + o0, len64, i, out, nullptr, match_only);
                } else {
                    has_line = true;
                    len = static_cast<uint32_t>(len64);
                    pos = 0;
                    row = row0;
                    const uint32_t acc = state_acc<TIER>(W, row);
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
                if (src + 16 <= data_end) v = reinterpret_cast<const UnalignedWindow*>(src)->v;
                else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; ++b)
                        if (src + b < data_end) w[b >> 2] |= static_cast<uint32_t>(src[b]) << ((b & 3) * 8);
                    v = u32x4{w[0], w[1], w[2], w[3]};
                }
            }
            lds_st<u32x4>(slice + static_cast<uint32_t>(q) * SLICE_ROW + (lane & 3u) * 16u, v);
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
            const u32x4 wv = lds_ld<u32x4>(my + w * 16u);
            const uint4 w0 = make_uint4(wv.x, wv.y, wv.z, wv.w);
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
                        if (!masked) row = steps16<TIER, true, false, true>(w0, mask, W, row, rel, regs);
                        else row = steps16<TIER, true, true, true>(w0, mask, W, row, rel, regs);
                    } else {
                        if (!masked) row = steps16<TIER, true, false, false>(w0, mask, W, row, rel, regs);
                        else row = steps16<TIER, true, true, false>(w0, mask, W, row, rel, regs);
                    }
                } else {
                    if (!masked) row = steps16<TIER, false, false, false>(w0, mask, W, row, rel, regs);
                    else row = steps16<TIER, false, true, false>(w0, mask, W, row, rel, regs);
                }
                const uint32_t acc = state_acc<TIER>(W, row);
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

template <typename OFF, int TIER>
hipError_t launch_slices_t(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, dim3 grid, dim3 block,
                           const GxBatch& b, hipStream_t stream) {
    hipError_t e = allow_full_lds(&k_extract_slices<OFF, TIER>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_extract_slices<OFF, TIER>), grid, block, lds.total_bytes, stream, dev, lds, lds_image, at_global,
                       static_cast<const uint8_t*>(b.data), static_cast<const OFF*>(b.offsets), b.n, line_out(dev, b), b.match_only, b.strip_eol);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Hop slice kernel: the slice kernel's staging and lane refill, the hop tier's walk (gx_hop_dev.hpp)
// ---------------------------------------------------------------------------
// For long and uneven lines on definitions that have the hop tier's tables (64 extractions and more): a wave owns a
// contiguous range of lines, every lane stages the next 128 bytes of ITS line from its own position (mapped to class ids
// on the way into LDS, eight lanes per line fetching 16 bytes each), walks them with a run and a chain per iteration,
// and comes back for the next piece 24 bytes before the end of what is staged (a whole window and a whole chain are
// always there); a lane whose line has ended writes its result and takes the wave's next line.  A line's padding, fields
// and literals cost iterations, not bytes, so lanes of a wave stay roughly level however long their lines are.
#ifndef GX_HOP_LEAVE
#define GX_HOP_LEAVE 24u   // (64: a round's walk goes on until every lane has used up its piece)
#endif
#ifndef GX_HOP_SERVICE
#define GX_HOP_SERVICE 64u  // finished lanes write their results and take new lines once that many lanes have nothing to walk.  64: when
                            // none has -- a wave takes 64 lines, walks them in rounds until the last is done, writes 64 rows, takes the next
                            // 64.  Round 4 (no run test by the loaders): 8 / 16 / 24 / 32 idle lanes 0.954 / 0.877 / 0.853 / 0.854 ms per 2 M
                            // lines.  Round 5, ms per 3.8 M lines: 16 / 24 / 32 / 40 / 48 / 56: 1.219 / 1.170 / 1.125 / 1.072 / 1.068 / 1.060;
                            // with lines done a round or two earlier (tested kilobytes repeated, behind the walk): 44 / 48 / 52 / 56 / 60 /
                            // 64: 0.994 / 0.961 / 0.970 / 0.958 / 0.950 / 0.944 -- a service costs ~7 K cycles whatever it serves, and a
                            // round costs its instructions whatever its lanes do
#endif
constexpr uint64_t HOP_POOL_CHUNK = 64;   // lines per draw from the pool (the last quarter of a batch's lines)
constexpr uint32_t HOP_SLICE = GX_HOP_SLICE_BYTES, HOP_SLICE_ROW = GX_HOP_SLICE_BYTES + 16u, HOP_SLICE_KEEP = 24;

// 16 staged bytes at `src` (WIDE: the low bytes of 16 code units = 32 bytes of the buffer; the high bytes of the first `units` of
// them ORed into `high`).  GUARDED: never reads at or beyond data_end, byte by byte -- for the last lines of a buffer only (a lane
// says so: `near_end`); everything else is a global_load_dwordx4 (two for WIDE) at whatever alignment the position has.  (Until
// round 5 these went out as FLAT loads with a 16-byte fallback unrolled behind each of them: 184 flat_load_ubyte in the kernel, and
// a flat load counts as an LDS operation too.)
typedef __attribute__((address_space(1))) const UnalignedWindow GlobalWindow;
template <bool WIDE, bool GUARDED>
__device__ __forceinline__ u32x4 load_chunk16(const uint8_t* src, const uint8_t* data_end, uint32_t units, uint32_t& high) {
    u32x4 a = {0u, 0u, 0u, 0u}, b2 = {0u, 0u, 0u, 0u};   // (WIDE: units 0-7, 8-15)
    if (!GUARDED) {
        a = reinterpret_cast<GlobalWindow*>(reinterpret_cast<uintptr_t>(src))->v;
        if (WIDE) b2 = reinterpret_cast<GlobalWindow*>(reinterpret_cast<uintptr_t>(src + 16))->v;
    } else {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma nounroll
        for (int q = 0; q < (WIDE ? 32 : 16); ++q)
            if (src + q < data_end) w[q >> 2] |= static_cast<uint32_t>(src[q]) << ((q & 3) * 8);
        a = u32x4{w[0], w[1], w[2], w[3]}; b2 = u32x4{w[4], w[5], w[6], w[7]};
    }
    if (!WIDE) return a;
    // (only the units of the line count: the units behind its end are another line's)
    const uint32_t hi[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
    uint32_t acc = 0u;
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q) {
        uint32_t m = 0xFF00FF00u;
        if (2u * q + 1u >= units) m = 2u * q >= units ? 0u : 0x0000FF00u;
        acc |= hi[q] & m;
    }
    high |= acc;
    // the low bytes of two units per dword -> four staged bytes per pair of dwords (v_perm_b32: bytes 0, 2 of each)
    return u32x4{__builtin_amdgcn_perm(a.y, a.x, 0x06040200u), __builtin_amdgcn_perm(a.w, a.z, 0x06040200u),
                 __builtin_amdgcn_perm(b2.y, b2.x, 0x06040200u), __builtin_amdgcn_perm(b2.w, b2.z, 0x06040200u)};
}
// all 16 bytes in the run interval runinfo = lo | (0x7F - hi) << 8 (0x8000: no interval -- never)
__device__ __forceinline__ bool chunk_in_run(const u32x4& v, uint32_t runinfo) {
    const uint32_t lo4 = splat_byte0(runinfo), k4 = splat_byte1(runinfo);
    // (x | x - lo | x + k per dword, the top bits of all of them at once)
    const uint32_t f = (v.x | (v.x - lo4) | (v.x + k4)) | (v.y | (v.y - lo4) | (v.y + k4)) | (v.z | (v.z - lo4) | (v.z + k4)) | (v.w | (v.w - lo4) | (v.w + k4));
    return (f & HI_BITS) == 0u;
}
#ifndef GX_HOP_SPEC
#define GX_HOP_SPEC 1   // 0: no chunks are tested beyond the piece
#endif
#ifndef GX_HOP_SPEC_UNITS
#define GX_HOP_SPEC_UNITS 24   // units of 128 bytes behind its piece that a line has tested per round (8 / 16 / 24: 0.969 / 0.922 / 0.920 ms per 3.8 M lines)
#endif
#ifndef GX_HOP_SPEC_BATCH
#define GX_HOP_SPEC_BATCH 8u   // (2 / 4 / 8 load instructions in flight: 0.961 / 0.928 / 0.905 ms per 3.8 M lines)
#endif
#ifndef GX_HOP_SPEC_MIN
#define GX_HOP_SPEC_MIN 0u   // only lines with more than that many bytes behind their piece have them tested (the others: the next pieces' loaders)
#endif


// WIDE: `data` holds UTF-16 code units (offsets in units): a loading lane fetches 16 units = 32 bytes and stages their low bytes; a line
// with a unit above 0xFF is flagged (wide_flags[i], *wide_any) for the per-line walk on the units (k_extract_flagged), as the tile kernel's.
template <typename OFF, bool WIDE>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(1, 3)))   // (12 waves at most: plan_hop_slice_launch)
k_extract_hop_slices(GxDev T, GxLds L, const uint8_t* __restrict__ lds_image, const uint8_t* __restrict__ at_global,
                     const uint8_t* __restrict__ data, const OFF* __restrict__ off, uint64_t n, LineOut out, int match_only, int strip_eol,
                     uint32_t* __restrict__ oversize_flag, uint32_t seq, unsigned long long* __restrict__ stamps,
                     uint32_t* __restrict__ pool_ctr, uint32_t pool_base, uint64_t n_static,
                     uint8_t* __restrict__ wide_flags, uint32_t* __restrict__ wide_any) {
    {
        extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];
        const uint4* src = reinterpret_cast<const uint4*>(lds_image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
    }
    __syncthreads();
#ifdef GX_DEV
    // developer build: cycles per phase summed per wave (tools/hop_slice_phases.py): 0 results + handing out lines, 1 loads issued,
    // 2 waiting for them + LDS stores, 3 walk, 4 rounds
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
    unsigned long long dv[6] = {0, 0, 0, 0, 0, 0};   // lanes with a piece, lanes that went into the walk, lanes with a tested kilobyte, services, (walk iterations, lanes in them: K)
    const unsigned long long hs_begin = __builtin_amdgcn_s_memrealtime();
#define HS_STAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[slot] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define HS_STAMP(slot) do { } while (0)
#endif
    HopTab H;
    H.rows = at_global;
    H.hops = at_global + L.c_base;
    H.row_bytes = L.row_bytes;
    H.info_off = L.ncls * 4u;
    H.n_hot = L.rec_indexed;
    H.lrow_cols = (2u * L.ncls + 3u) & ~3u;
    H.sets = L.hop_sets;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t slice = L.stage + wave * L.stage_bytes;
    const uint32_t regs = L.regs + wave * L.regs_wave_bytes + 128u + lane * 2u;
    const uint8_t* data_end = data + (static_cast<uint64_t>(off[n]) << (WIDE ? 1 : 0));
    // (match only: the tables are the match automaton's, a state's info word is its first accepting extraction)
    const uint32_t row0 = match_only ? L.m_start : L.u_start, dead_row = match_only ? L.m_dead : L.u_dead;
    const uint8_t* fin_g = L.at != 0u ? nullptr : at_global + L.fin_tags;
    const uint32_t fin_lds = L.at;

    // A wave's own range of lines (of the first n_static), then chunks of HOP_POOL_CHUNK lines out of the rest, drawn from a counter
    // in global memory: lines of 50-2000 bytes make equal shares of LINES unequal shares of bytes -- the last wave of BASELINE
    // configs[4] ended 8.5 % behind the median one (1 413 against 1 302 us; tools/hop_slice_phases.py).  The counter is never reset:
    // chunk = ticket - pool_base, and every wave draws until its ticket is past the last chunk -- once: the host knows what the counter
    // reads when the launch is over (hop_slices_tickets).
    const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * L.nwaves;
    const uint64_t per_wave = (n_static + nwaves - 1) / nwaves;
    const uint64_t wid = static_cast<uint64_t>(blockIdx.x) * L.nwaves + wave;
    const uint64_t range_lo = min(n_static, wid * per_wave);
    uint64_t range_hi = min(n_static, range_lo + per_wave);
    uint64_t next = range_lo;   // first line of the range not yet handed to a lane (wave-uniform)
    const uint64_t pool_chunks = (n - n_static + HOP_POOL_CHUNK - 1) / HOP_POOL_CHUNK;
    bool pool_done = pool_ctr == nullptr;

    bool has_line = false;
    uint64_t i = 0, o0 = 0;
    uint32_t len = 0, pos = 0, row = row0;
    const uint32_t my = slice + lane * HOP_SLICE_ROW;
    bool line_wide = false;   // WIDE: the line holds a unit above 0xFF
    const uint64_t total_units = static_cast<uint64_t>(off[n]);
    // the offsets of the lines [win_base, win_base + 64]: lane l holds those of line win_base + l
    uint64_t win_base = ~0ull;
    OFF w_lo = 0, w_hi = 0;
    auto refill = [&](uint64_t base) {
        win_base = base;
        const uint64_t a = min(base + lane, n);
        w_lo = off[a];
        w_hi = off[min(a + 1u, n)];
    };
    refill(range_lo);
    FinAhead F;               // the final record of the state a finished line ended in, asked for behind the walk
    bool fin_here = false;
    uint32_t near_end = 0u;   // the line ends within 16 units of the buffer's end: its loaders read byte by byte
    HopKept K;                // the record the lane holds while it stays in its state (kept across rounds)
    const bool all_hot = L.rec_indexed >= L.sort_chunk;

    for (;;) {
        // ---- finished lanes write their results once a quarter of the wave is idle (k_extract_slices says why) ----
        const bool finished = has_line && (pos >= len || row == dead_row);
        const uint32_t idle = static_cast<uint32_t>(__popcll(__ballot(finished || !has_line)));
        const bool service = idle >= GX_HOP_SERVICE || !__any(has_line && !finished);
        if (service && finished) {
            if (!match_only) lds_st<uint16_t>(regs - 128u, static_cast<uint16_t>(len));   // (the dummy column, free now: the tag "the line's length" names it)
            if (!match_only && L.fin_state_off != 0u) {
                // (the row out of the final record of the state: one read, no info word first -- asked for when the line ended, behind
                // the walk; a line that ended where it began is read now)
                const uint8_t* recp = at_global + L.fin_state_off + static_cast<uint64_t>(row) * L.fin_state_rec;
                if (!fin_here) F.load(recp, T.max_groups);
                write_row_rec(out, i, recp, F, regs, len, T.max_groups, L.fin_unset);
                fin_here = false;
            } else {
                const int32_t hot_info = static_cast<int16_t>(lds_ld<uint16_t>(L.acc_tab + 2u * min(row, H.n_hot - 1u)));
                int32_t info = hot_info >= 0 && !match_only ? hot_info * 16 : hot_info;
                if (row >= H.n_hot) info = *reinterpret_cast<const int32_t*>(H.rows + (static_cast<uint64_t>(row) * H.row_bytes + H.info_off));
                if (match_only) out.id(i, info);
                else write_row<TIER_HOP>(out, i, info, fin_lds, fin_g, regs, len, T.max_groups, L.fin_unset);
            }
            if (WIDE) {   // (a flagged line's result is the per-line walk's to write again)
                wide_flags[i] = line_wide ? 1 : 0;
                if (line_wide) __hip_atomic_store(wide_any, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            has_line = false;
        }
#ifdef GX_DEV
        if (service) ++dv[3];
#endif
        HS_STAMP(5);
        // ---- the range is used up: the next chunk of the pool ----
        if (service && next >= range_hi && !pool_done) {
            uint32_t t = 0u;
            if (lane == 0u) t = __hip_atomic_fetch_add(pool_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - pool_base;
            t = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
            if (t < pool_chunks) {
                next = n_static + static_cast<uint64_t>(t) * HOP_POOL_CHUNK;
                range_hi = min(n, next + HOP_POOL_CHUNK);
            } else pool_done = true;
        }
        // ---- free lanes take the next lines of the range, in lane order ----
        const uint64_t free_mask = __ballot(!has_line);
        if (service && free_mask && next < range_hi) {
            const uint32_t rank = static_cast<uint32_t>(__popcll(free_mask & ((1ull << lane) - 1ull)));
            const uint64_t cand = next + rank;
            // (the offsets of the wave's next 64 lines are on their way since its last hand-out, two per lane: a new line's first piece
            // was a second trip to memory behind the one for its offsets)
            const uint32_t n_free = static_cast<uint32_t>(__popcll(free_mask));
            const bool in_win = next >= win_base && next - win_base + n_free <= 64u;   // (wave-uniform)
            OFF o_lo = 0, o_hi = 0;
            if (in_win) {
                const int src = static_cast<int>(static_cast<uint32_t>(next - win_base) + rank) & 63;
                o_lo = __shfl(w_lo, src);
                o_hi = __shfl(w_hi, src);
            }
            if (!has_line && cand < range_hi) {
                i = cand;
                if (!in_win) { o_lo = off[i]; o_hi = off[i + 1]; }
                o0 = o_lo;
                int64_t len64 = static_cast<int64_t>(static_cast<uint64_t>(o_hi) - o0);
                if (strip_eol) len64 = WIDE ? trim_eol(reinterpret_cast<const uint16_t*>(data) + o0, len64) : trim_eol(data + o0, len64);
                line_wide = false;
                if (WIDE && len64 > 65535) wide_flags[i] = 0;   // (the follow-up launch walks its units)
                if (len64 > 65535) {
                    // positions are 16-bit in the register block: such a line is left to the follow-up launch of the per-line
                    // kernel (not walked here: its 96 capture registers would give every lane of this kernel a scratch frame)
                    __hip_atomic_store(oversize_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    has_line = true;
                    len = static_cast<uint32_t>(len64);
                    pos = 0;
                    row = row0;
                    near_end = static_cast<uint64_t>(o_hi) + 16u > total_units ? 0x10000u : 0u;
                }
            }
            next = min(range_hi, next + static_cast<uint64_t>(n_free));
            refill(next);
        }
        if (!__any(has_line)) {
            if (next >= range_hi && pool_done) break;
            continue;  // (only lines for the follow-up launch were handed out, or the range is used up: hand out more / draw a chunk)
        }
        HS_STAMP(0);
        // ---- stage the next piece of every lane's line, from the lane's own position: lane l fetches 16 bytes (l & 7) of the
        // line of lane (l >> 3) + 8 r ----
        const bool walking = has_line && pos < len && row != dead_row;
        // (HOP_SLICE / 16 lanes per line: 8 for 128 bytes)
        constexpr uint32_t LPL = HOP_SLICE / 16u, LINES_PER_LOAD = 64u / LPL;
        static_assert(LPL == 8u, "the loaders' run bits are a byte per line");
        // The run interval of the state the lane is in, for the LOADERS: a lane in a long value (half of configs[4]'s bytes are the
        // padded last value; 28 of a line's 43 walk iterations were the next 16 bytes of the same run) has its piece tested where it
        // is loaded -- 16 bytes per loader, all lanes busy, no dependent LDS round trips -- and begins its walk behind the chunks
        // that lie in the run.  0x8000: no interval (every byte fails).  The record: the one the lane holds, or the hot one in LDS.
        uint32_t runinfo = 0x8000u;
        if (walking) {
            if (!all_hot && row == K.kept) runinfo = K.k0.x & 0xFFFFu;
            else if (row < H.n_hot) runinfo = lds_ld<uint32_t>(__umul24(row, HOP_REC_B) + HOP_LDS_AT) & 0xFFFFu;
        }
        // (what a loading lane has to know of the line it loads for -- where its piece begins, how many bytes are left of it and
        // the run interval -- goes through the piece buffer itself, which holds nothing at this point: one 16-byte store per lane
        // and one broadcast read per load, instead of four shuffles per load)
        const uint32_t left_now = walking ? len - pos : 0u;
        {
            const uint8_t* mine = data + ((o0 + pos) << (WIDE ? 1 : 0));
            const uint64_t mv = reinterpret_cast<uint64_t>(mine);
            lds_st<u32x4>(slice + lane * 16u, u32x4{static_cast<uint32_t>(mv), static_cast<uint32_t>(mv >> 32), left_now, runinfo | near_end});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4 who[LPL];
#pragma unroll
        for (uint32_t r = 0; r < LPL; ++r) who[r] = lds_ld<u32x4>(slice + (lane / LPL + LINES_PER_LOAD * r) * 16u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (every lane has read before anything else is stored there)
        // (Round 5: one dword of the 128-byte line that the lane's NEXT piece will begin to need, asked for with this piece's loads -- a
        // round ahead, so that those loads find it in L2: 0.829 against 0.814 ms per 3.8 M lines of configs[4], one device.  Not kept.)
        u32x4 pv[LPL];
        uint32_t high_or[LPL];   // WIDE: the high bytes of the units a lane loaded, ORed
#pragma unroll
        for (uint32_t r = 0; r < LPL; ++r) {
            const uint32_t at_byte = (lane % LPL) * 16u;   // (in staged bytes = code units)
            pv[r] = u32x4{0u, 0u, 0u, 0u};
            high_or[r] = 0u;
            const bool mine = at_byte < who[r].z;
            const uint8_t* src = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(who[r].y) << 32 | who[r].x) + (at_byte << (WIDE ? 1 : 0));
            if (__builtin_amdgcn_ballot_w64(mine && (who[r].w & 0x10000u) != 0u) == 0ull) {   // (wave-uniform: no line of this instruction ends at the buffer's end)
                if (mine) pv[r] = load_chunk16<WIDE, false>(src, data_end, who[r].z - at_byte, high_or[r]);
            } else if (mine) pv[r] = load_chunk16<WIDE, true>(src, data_end, who[r].z - at_byte, high_or[r]);
        }
        // ---- A lane that used up its piece inside a run (the loaders found every chunk in the run, or the walk ran to the piece's last
        // byte) has the bytes BEHIND the piece tested, not staged: loaders of 16 bytes, a ballot, and the lane moves behind the chunks
        // that lie in the run -- a padded value of a kilobyte is one round of loads and no walk iteration (it was 64 iterations over ten
        // rounds).  How this part got its shape (ms per 3.8 M lines of configs[4], one device per comparison): without it 1.304 against
        // 1.168; one load instruction per such line, sent out before the walk and looked at behind it -- 0 / 4 / 6 lines ahead -- 1.167 /
        // 1.157 / 1.168; sent out WITH the piece's loads for lanes guessed to be in a run: 1.022 / 1.032 / 1.052 / 1.124 for 0 / 2 / 4 /
        // 6 guesses; two short remainders per instruction 1.045 against 1.059; only lines with more than 128 / 256 / 384 / 640 bytes
        // left: 1.074 / 1.107 / 1.119 / 1.153 against 1.036; repeated while everything tested lay in the run, 1 / 2 / 3 times: 1.055 /
        // 1.010 / 0.987; behind the walk instead of before it: 0.992 against 0.997; in units of all lines at once (below): 0.920 against
        // 0.936; 2 / 4 / 8 / 12 load instructions in flight: 0.961 / 0.928 / 0.905 / 0.967. ----
        // `chunk`: which 16 bytes behind the piece this lane tests (a line of its own: 0 .. 63; a line that shares the instruction with
        // another: 0 .. 31)
        auto spec_load = [&](const u32x4& d, uint32_t chunk, bool valid, uint32_t& high) -> u32x4 {
            const uint32_t left2 = d.z;
            int32_t at = static_cast<int32_t>(chunk * 16u);
            const bool mine = valid && static_cast<uint32_t>(at) < left2;
            if (static_cast<uint32_t>(at) + 16u > left2) at = static_cast<int32_t>(left2) - 16;
            high = 1u;
            if (!mine) return u32x4{0u, 0u, 0u, 0u};
            const uint8_t* base = reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(d.y) << 32 | d.x);
            const uint8_t* src = WIDE ? base + 2 * static_cast<int64_t>(at) : base + at;
            high = 0u;
            return load_chunk16<WIDE, false>(src, data_end, 16u, high);   // (inside the line: the window was moved back from its end)
        };
        // which chunks lie in their line's run (whole chunks inside the line), and -- WIDE -- which lines hold a unit above 0xFF: eight
        // loaders per line and instruction -> one byte of the instruction's ballot per line; the ballots go through the piece buffer's
        // first 128 bytes (read again below, before the pieces are stored)
#pragma unroll
        for (uint32_t r = 0; r < LPL; ++r) {
            const uint32_t at_byte = (lane % LPL) * 16u;
            const bool full = at_byte + 16u <= who[r].z && chunk_in_run(pv[r], who[r].w) && (!WIDE || high_or[r] == 0u);
            const uint64_t bal = __builtin_amdgcn_ballot_w64(full);
            if (lane == 0u) lds_st<u32x2>(slice + 8u * r, u32x2{static_cast<uint32_t>(bal), static_cast<uint32_t>(bal >> 32)});
            if (WIDE) {
                const uint64_t wbal = __builtin_amdgcn_ballot_w64(high_or[r] != 0u);
                if (lane == 0u) lds_st<u32x2>(slice + 64u + 8u * r, u32x2{static_cast<uint32_t>(wbal), static_cast<uint32_t>(wbal >> 32)});
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t run_bits = lds_ld<uint8_t>(slice + lane);   // bit c: chunk c of this lane's piece lies in its run
        if (WIDE) {
            if (lds_ld<uint8_t>(slice + 64u + lane) != 0u && walking) line_wide = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HS_STAMP(1);
#pragma unroll
        for (uint32_t r = 0; r < LPL; ++r)
            lds_st<u32x4>(slice + (lane / LPL + LINES_PER_LOAD * r) * HOP_SLICE_ROW + (lane % LPL) * 16u, pv[r]);
        // the bytes of the piece that the run covers: whole leading chunks
        const uint32_t skip = walking ? 16u * static_cast<uint32_t>(__builtin_ctz(~run_bits | 0x100u)) : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        HS_STAMP(2);
        bool used_up = false;
        // ---- walk the piece ----
        {
            const uint32_t left = left_now;                          // bytes of the line from pos on
            const bool whole = left <= HOP_SLICE;                   // the line ends inside the piece
            const uint32_t e = my + (whole ? left : HOP_SLICE);
            const uint32_t limit = whole ? e : e - HOP_SLICE_KEEP;
            uint32_t p = my + skip;                                 // (behind the chunks the loaders found in the run: possibly at e, beyond the limit)
            const uint32_t e_chain = whole ? e : 0xFFFFFFF0u;
            // the round ends when GX_HOP_LEAVE of the lanes that went into it have used up their pieces (or their lines).  BASELINE
            // configs[4], 2 M lines, ms per launch (captures / match only), one device: 64 (the last lane) 0.955 / 0.712, 48 0.914 / 0.666,
            // 32 0.892 / 0.660, 16 0.907 / 0.670; then with 32: results and new lines at 8 idle lanes 0.954, 16 0.877, 24 0.853, 32 0.854;
            // 24 and 24: 0.849 / 0.627 (tools/hop_stats.py replays the rounds: 58 % of the lanes have something to walk in an iteration
            // of a round that waits for its last lane, 85 % with 24).  Round 5, with the loaders' run test: 16 / 24 / 32 / 40: no difference.
            const uint32_t went_in = static_cast<uint32_t>(__popcll(__ballot(p < limit)));
#ifdef GX_DEV
            dv[0] += static_cast<unsigned long long>(__popcll(__ballot(walking)));
            dv[1] += went_in;
#endif
            const uint32_t leave_at = went_in > GX_HOP_LEAVE ? went_in - GX_HOP_LEAVE : 0u;
            if (match_only) row = all_hot ? walk_hop_span<true, false>(H, p, e, limit, e_chain, my - pos, row, dead_row, regs, leave_at, K)
                                          : walk_hop_span<false, false>(H, p, e, limit, e_chain, my - pos, row, dead_row, regs, leave_at, K);
            else row = all_hot ? walk_hop_span<true, true>(H, p, e, limit, e_chain, my - pos, row, dead_row, regs, leave_at, K)
                               : walk_hop_span<false, true>(H, p, e, limit, e_chain, my - pos, row, dead_row, regs, leave_at, K);
            pos += p - my;
            used_up = walking && !whole && p == e;   // (every staged byte consumed: only a run does that -- the loaders' or the walk's)
        }
        HS_STAMP(3);
        uint32_t extra = 0u;
        // (behind the walk since round 5's end: a lane that REACHES its long value in this piece -- the walk ran to the piece's last byte in
        // a run -- has its next kilobyte tested in this round, not in the next one: a line is done a round earlier)
        uint32_t runinfo2 = 0x8000u;
        if (used_up) {
            if (!all_hot && row == K.kept) runinfo2 = K.k0.x & 0xFFFFu;
            else if (row < H.n_hot) runinfo2 = lds_ld<uint32_t>(__umul24(row, HOP_REC_B) + HOP_LDS_AT) & 0xFFFFu;
        }
        const bool longrun = GX_HOP_SPEC && used_up && runinfo2 != 0x8000u && left_now > HOP_SLICE + GX_HOP_SPEC_MIN;
#ifdef GX_DEV
        dv[2] += static_cast<unsigned long long>(__popcll(__ballot(longrun)));
#endif
        // The bytes behind the pieces, of all such lines at once, in UNITS of 128 bytes (up to GX_HOP_SPEC_UNITS per line and round):
        // a load instruction tests eight units -- eight loaders each -- of whichever lines they belong to, so its 64 lanes are all busy
        // (one instruction per line left 60 % of them idle: half of the remainders are shorter than 400 bytes).  Through the piece
        // buffer, which holds nothing any more: owner[u] = the lane whose line unit u belongs to, the lanes' descriptors, a byte of
        // chunk bits per unit.
        {
            constexpr uint32_t UNIT = 128u, SB = GX_HOP_SPEC_BATCH;   // (that many load instructions in flight)
            const uint32_t rest0 = longrun ? left_now - HOP_SLICE : 0u;
            const uint32_t units = longrun ? min((rest0 + UNIT - 1u) / UNIT, static_cast<uint32_t>(GX_HOP_SPEC_UNITS)) : 0u;
            uint32_t incl = units;
#pragma unroll
            for (uint32_t dlt = 1u; dlt < 64u; dlt <<= 1) {
                const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), dlt));
                if (lane >= dlt) incl += up;
            }
            const uint32_t first = incl - units;
            const uint32_t n_units = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
            if (n_units != 0u) {
                const uint32_t OWN = slice, BITS = slice + 2048u, DESC = slice + 4096u;
                if (longrun) {
                    const uint8_t* mine = data + ((o0 + pos) << (WIDE ? 1 : 0));   // (pos is behind the piece by now)
                    const uint64_t mv = reinterpret_cast<uint64_t>(mine);
                    lds_st<u32x4>(DESC + lane * 16u, u32x4{static_cast<uint32_t>(mv), static_cast<uint32_t>(mv >> 32), rest0, runinfo2 | first << 16});
                }
                for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(k < units) != 0ull; ++k)
                    if (k < units) lds_st<uint8_t>(OWN + first + k, static_cast<uint8_t>(lane));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t n_ins = (n_units + 7u) >> 3;
                for (uint32_t i0 = 0; i0 < n_ins; i0 += SB) {
                    u32x4 d[SB], sv[SB];
                    uint32_t sh[SB];
#pragma unroll
                    for (uint32_t u = 0; u < SB; ++u) {
                        const uint32_t unit = min(8u * (i0 + u) + (lane >> 3), n_units - 1u);
                        d[u] = lds_ld<u32x4>(DESC + static_cast<uint32_t>(lds_ld<uint8_t>(OWN + unit)) * 16u);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < SB; ++u) {
                        const uint32_t unit = 8u * (i0 + u) + (lane >> 3);
                        const uint32_t chunk = (min(unit, n_units - 1u) - (d[u].w >> 16)) * 8u + (lane & 7u);   // of the line
                        sv[u] = spec_load(d[u], chunk, unit < n_units, sh[u]);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < SB; ++u) {
                        const uint64_t bal = __builtin_amdgcn_ballot_w64(sh[u] == 0u && chunk_in_run(sv[u], d[u].w));
                        if (lane == 0u && i0 + u < n_ins) lds_st<u32x2>(BITS + 8u * (i0 + u), u32x2{static_cast<uint32_t>(bal), static_cast<uint32_t>(bal >> 32)});
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // a line's leading units that lie in the run, and the chunks of the first one that does not
                uint32_t got = 0u;
                bool alive = longrun;
                for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(alive && k < units) != 0ull; ++k) {
                    const bool look = alive && k < units;
                    const uint32_t bits = look ? lds_ld<uint8_t>(BITS + first + k) : 0xFFu;
                    if (look) {
                        got += 16u * static_cast<uint32_t>(__builtin_ctz(~bits | 0x100u));
                        alive = bits == 0xFFu;
                    }
                }
                extra = min(got, rest0);
            }
        }
        pos += extra;
        if (!match_only && L.fin_state_off != 0u) {
            if (has_line && !fin_here && (pos >= len || row == dead_row)) {
                F.load(at_global + L.fin_state_off + static_cast<uint64_t>(row) * L.fin_state_rec, T.max_groups);
                fin_here = true;
            }
        }
        // the slice buffer is rewritten by the next iteration
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        HS_STAMP(2);   // (the tested kilobytes count with the waiting)
#ifdef GX_DEV
        ++ph[4];
#endif
    }
#ifdef GX_DEV
    if (stamps && lane == 0) {
        unsigned long long* st = stamps + 16ull * (static_cast<uint64_t>(blockIdx.x) * L.nwaves + wave);
        for (int q = 0; q < 6; ++q) st[q] = ph[q];
        st[6] = hs_begin;
        st[7] = __builtin_amdgcn_s_memrealtime();
        for (int q = 0; q < 4; ++q) st[8 + q] = dv[q];
        st[12] = K.dev_iters;
        st[13] = K.dev_lane_iters;
    }
#endif
}

}  // namespace

// lds: a layout from plan_hop_slice_launch (gx_api.cpp): the hop tier's tables, per wave a register block and a [64][144]-byte piece buffer
namespace {
uint64_t hop_slices_blocks(uint64_t n, uint32_t nwaves, int num_cus) {
    uint64_t blocks = static_cast<uint64_t>(num_cus);
    const uint64_t need = (n + 256ull * nwaves - 1) / (256ull * nwaves);
    return blocks > need ? need : blocks;
}
uint64_t hop_slices_static(uint64_t n) { return n - (n / 4) / HOP_POOL_CHUNK * HOP_POOL_CHUNK; }   // the lines outside the pool
}  // namespace

// what a launch with b.chunk_ctr set adds to that counter: a draw per chunk of the pool and the one draw past it of every wave
uint32_t hop_slices_tickets(uint64_t n, uint32_t nwaves, int num_cus) {
    if (n == 0) return 0u;
    const uint64_t chunks = (n - hop_slices_static(n) + HOP_POOL_CHUNK - 1) / HOP_POOL_CHUNK;
    return static_cast<uint32_t>(chunks + hop_slices_blocks(n, nwaves, num_cus) * nwaves);
}

hipError_t launch_extract_hop_slices(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                     const GxBatch& b, hipStream_t stream, unsigned long long* stamps) {
    if (b.n == 0) return hipSuccess;
    const uint64_t blocks = hop_slices_blocks(b.n, lds.nwaves, num_cus);
    const uint64_t n_static = b.chunk_ctr ? hop_slices_static(b.n) : b.n;
    const dim3 grid(static_cast<unsigned>(blocks)), block(lds.nwaves * 64);
    auto go = [&](auto kernel, auto offsets) -> hipError_t {
        hipError_t e = allow_full_lds(kernel);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, block, lds.total_bytes, stream, dev, lds, lds_image, at_global, static_cast<const uint8_t*>(b.data), offsets, b.n, line_out(dev, b),
                           (b.match_only != 0 || !dev.has_capture) ? 1 : 0, b.strip_eol, b.oversize_flag, b.seq, stamps, b.chunk_ctr, b.chunk_base, n_static,
                           b.wide_flags, b.wide_any);
        return hipGetLastError();
    };
    if (b.wide) {   // UTF-16 code units (b.wide_flags / b.wide_any set by the caller: gx_api.cpp)
        if (!b.wide_flags || !b.wide_any) return hipErrorInvalidValue;
        if (b.offsets64) return go(&k_extract_hop_slices<uint64_t, true>, static_cast<const uint64_t*>(b.offsets));
        return go(&k_extract_hop_slices<uint32_t, true>, static_cast<const uint32_t*>(b.offsets));
    }
    if (b.offsets64) return go(&k_extract_hop_slices<uint64_t, false>, static_cast<const uint64_t*>(b.offsets));
    return go(&k_extract_hop_slices<uint32_t, false>, static_cast<const uint32_t*>(b.offsets));
}

// b: the UTF-16 batch (data = code units, offsets in units).  bytes: room for the batch's units as bytes, addressed like the
// units (bytes[offsets[i]] is line i's first); flags: n bytes, zeroed here.
hipError_t launch_narrow_units(const GxBatch& b, uint8_t* bytes_at_first_unit, uint8_t* flags, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(flags, 0, b.n, stream);
    if (e != hipSuccess) return e;
    const dim3 grid(256u * 16u), block(256);
    if (b.offsets64) hipLaunchKernelGGL((k_narrow_units<uint64_t>), grid, block, 0, stream, static_cast<const uint16_t*>(b.data), static_cast<const uint64_t*>(b.offsets), b.n, bytes_at_first_unit, flags);
    else hipLaunchKernelGGL((k_narrow_units<uint32_t>), grid, block, 0, stream, static_cast<const uint16_t*>(b.data), static_cast<const uint32_t*>(b.offsets), b.n, bytes_at_first_unit, flags);
    return hipGetLastError();
}
hipError_t launch_extract_flagged(const GxDev& dev, const GxBatch& b, const uint8_t* flags, hipStream_t stream, const uint32_t* any_word) {
    if (b.n == 0) return hipSuccess;
    const dim3 grid(dev.pike_off ? dev.pike_blocks : 256u * 4u), block(256);
    const uint16_t* units = static_cast<const uint16_t*>(b.data);
#define GX_FLAGGED(OFF, MS, NEXT) hipLaunchKernelGGL((k_extract_flagged<OFF, MS>), grid, block, 0, stream, dev, units, static_cast<const OFF*>(b.offsets), b.n, flags, line_out(dev, b), b.match_only, NEXT, b.strip_eol, any_word, b.seq)
    if (b.offsets64) { if (dev.m_next16) GX_FLAGGED(uint64_t, uint16_t, dev.m_next16); else GX_FLAGGED(uint64_t, uint32_t, dev.m_next32); }
    else { if (dev.m_next16) GX_FLAGGED(uint32_t, uint16_t, dev.m_next16); else GX_FLAGGED(uint32_t, uint32_t, dev.m_next32); }
#undef GX_FLAGGED
    return hipGetLastError();
}

hipError_t launch_extract_one(const GxDev& dev, const uint16_t* units, uint32_t len, const GxBatch& b, hipStream_t stream) {
    if (len > ONE_LINE_MAX_UNITS) return hipErrorInvalidValue;
    if (dev.m_next16)
        hipLaunchKernelGGL((k_extract_one<uint16_t>), dim3(1), dim3(64), 0, stream, dev, units, len, line_out(dev, b), b.state_out, b.match_only, dev.m_next16);
    else
        hipLaunchKernelGGL((k_extract_one<uint32_t>), dim3(1), dim3(64), 0, stream, dev, units, len, line_out(dev, b), b.state_out, b.match_only, dev.m_next32);
    return hipGetLastError();
}

hipError_t launch_extract_generic(const GxDev& dev, const GxBatch& b, hipStream_t stream) {
    if (b.wide) {
        return b.offsets64 ? launch_generic_t<uint16_t, uint64_t>(dev, b, stream) : launch_generic_t<uint16_t, uint32_t>(dev, b, stream);
    }
    return b.offsets64 ? launch_generic_t<uint8_t, uint64_t>(dev, b, stream) : launch_generic_t<uint8_t, uint32_t>(dev, b, stream);
}

hipError_t launch_extract_oversize(const GxDev& dev, const GxBatch& b, uint32_t limit, int by_length, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    // a small grid: without the flag every wave leaves at once; with it, the lines in question are few and long
    const dim3 grid(dev.pike_off ? dev.pike_blocks : 256u), block(256);
#define GX_OVERSIZE(OFF, CH) hipLaunchKernelGGL((k_extract_oversize<OFF, CH>), grid, block, 0, stream, dev, static_cast<const CH*>(b.data), \
                           static_cast<const OFF*>(b.offsets), b.n, line_out(dev, b), b.match_only, b.strip_eol, b.oversize_flag, b.seq, limit, by_length, b.state_out)
    if (b.wide) { if (b.offsets64) GX_OVERSIZE(uint64_t, uint16_t); else GX_OVERSIZE(uint32_t, uint16_t); }
    else { if (b.offsets64) GX_OVERSIZE(uint64_t, uint8_t); else GX_OVERSIZE(uint32_t, uint8_t); }
#undef GX_OVERSIZE
    return hipGetLastError();
}

// Slice kernel (see k_extract_slices): lds.stage_bytes = 64 * 80, lds.nwaves <= 16.
hipError_t launch_extract_slices(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                 const GxBatch& b, hipStream_t stream) {
    if (b.n == 0) return hipSuccess;
    // every wave owns a contiguous range of lines; at least 256 lines per wave so that lanes can be refilled
    uint64_t blocks = static_cast<uint64_t>(num_cus);
    const uint64_t need = (b.n + 256ull * lds.nwaves - 1) / (256ull * lds.nwaves);
    if (blocks > need) blocks = need;
    dim3 grid(static_cast<unsigned>(blocks)), block(lds.nwaves * 64);
    if (at_global && lds.tier == 3) {
        if (b.offsets64) return launch_slices_t<uint64_t, TIER_RECG>(dev, lds, lds_image, at_global, grid, block, b, stream);
        return launch_slices_t<uint32_t, TIER_RECG>(dev, lds, lds_image, at_global, grid, block, b, stream);
    }
    if (at_global) {
        if (b.offsets64) return launch_slices_t<uint64_t, TIER_L2>(dev, lds, lds_image, at_global, grid, block, b, stream);
        return launch_slices_t<uint32_t, TIER_L2>(dev, lds, lds_image, at_global, grid, block, b, stream);
    }
    if (lds.tier == 2) {
        if (b.offsets64) return launch_slices_t<uint64_t, TIER_REC>(dev, lds, lds_image, nullptr, grid, block, b, stream);
        return launch_slices_t<uint32_t, TIER_REC>(dev, lds, lds_image, nullptr, grid, block, b, stream);
    }
    if (b.offsets64) return launch_slices_t<uint64_t, TIER_LDS>(dev, lds, lds_image, nullptr, grid, block, b, stream);
    return launch_slices_t<uint32_t, TIER_LDS>(dev, lds, lds_image, nullptr, grid, block, b, stream);
}

}  // namespace gx
