// gx_device.hpp -- device-side view of the compiled tables + kernel launch entry points.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace gx {

// All pointers are device pointers into ONE allocation (the uploaded table image).
struct GxDev {
    // class maps
    const uint8_t* cls256;       // [256]
    const uint16_t* hi_lo;       // [n_hi] ascending lower bounds, hi_lo[0] == 256
    const uint16_t* hi_cls;      // [n_hi]
    int32_t n_hi;
    int32_t ncls;
    // match automaton (total; m_dead absorbing)
    const uint16_t* m_next16;    // [m_states * ncls] when m_states <= 65536, else null
    const uint32_t* m_next32;    // [m_states * ncls] otherwise
    const int32_t* m_accept_first;  // [m_states]
    int32_t m_states;
    int32_t m_dead;
    // capture automata, all extractions concatenated
    const uint32_t* c_trans;     // rule k at c_trans + c_trans_off[k], [n_states_k * ncls]
    const uint32_t* c_trans_off; // [n_rules]
    const int32_t* c_fin;        // rule k at c_fin + c_fin_off[k], [n_states_k]
    const uint32_t* c_fin_off;   // [n_rules]
    const int32_t* c_ngroups;    // [n_rules]
    const uint32_t* ops_off;     // [n_oplists + 1] in (dst,src) pairs
    const uint16_t* ops;         // pairs
    const uint16_t* fin_tags;
    int32_t n_rules;
    int32_t max_groups;
    int32_t max_regs;
    int32_t has_capture;
    // extractions without a capture automaton (gx_compile.hpp: Tables::pike_*): their programs, run as they are.  pike_off null: none.
    const uint32_t* pike_off;    // [n_rules + 1]
    const uint32_t* pike_code;   // two words per instruction
    const uint32_t* pike_sets;   // eight words per set
    int32_t* pike_scratch;       // [256 pike_blocks + 1][pike_lane_ints]: the thread lists of the lanes that run one (the last: the one-line calls)
    uint32_t pike_lane_ints;
    uint32_t pike_blocks;        // the per-line kernels of a handle that has such extractions run this many workgroups of 256 lanes at most
};
constexpr uint32_t GX_PIKE_BLOCKS = 64;                     // ... 64 when the thread lists of 16 K lanes stay within GX_PIKE_SCRATCH_BYTES, else fewer
constexpr uint64_t GX_PIKE_SCRATCH_BYTES = 256ull << 20;

// Layout of the LDS-resident table image used by the tile kernel: byte offsets
// from the start of dynamic LDS.  The image [0, table_bytes) is built on the
// host, kept in HBM next to the other tables and copied into LDS by every
// workgroup's prologue.
struct GxLds {
    uint32_t cmap;        // u16[272] byte -> class * 4 (the byte offset of the class's column in a row), entry 256 = the
                          // identity column; always at offset 0 (the kernel indexes LDS by 2 * byte value)
    uint32_t ncls;        // number of classes = index of the identity column
    uint32_t at;          // automaton rows, u32[rows][ncls + 3]: one row per state of the match automaton and of
                          // the fused automaton (or of every extraction's capture automaton).  Columns
                          // 0..ncls-1: successor | capture program << 16, where the successor is the LDS byte
                          // address of its row (LDS tier; always < 65536) or its state index (L2 tier); column
                          // ncls: identity (self, no program); column ncls+1: self-loop byte interval as
                          // lo | (0x7F - hi) << 8 (0x8000: none), bit 16 set when the state also loops on every byte
                          // of the hot interval (hot_lo4 / hot_k4 below); column ncls+2: info (match automaton: first
                          // accepting extraction or -1; capture automaton: byte offset of the state's final
                          // record, or -1 / -2-k)
    uint32_t row_bytes;   // (ncls + 3) * 4
    uint32_t c_base;      // L2 tier: byte offset of the capture rows inside the global row image (0 in the LDS tier)
    uint32_t m_start;     // row (LDS address / state index) of the match automaton's start state
    uint32_t m_dead;      // row of its absorbing dead state
    uint32_t c_rule;      // u32[n_rules * 2]: row of the rule's start state, group count
    uint32_t u_start;     // row of the fused automaton's start state, or 0xFFFFFFFF when absent
    uint32_t u_dead;      // row of its dead state
    uint32_t ops_off;     // u32[n_oplists + 1]
    uint32_t ops;         // u16 pairs
    uint32_t fin_tags;    // final records (gx_walk.hpp: line_result): LDS address, or byte offset in the global row image (L2 tier)
    uint32_t table_bytes; // size of the image, multiple of 16
    uint32_t simple_ops;  // 1: every capture program is one "register := position"; the program field is then
                          // (register + 1) * 128 = byte offset of the register's column in the wave's register
                          // block (0 = the dummy column = no program) and steps are branch-free;
                          // 0: program field 0 = none, 0x8000 | register = single set, else op-list index
    uint32_t regs;        // u16[nwaves][1 + max_regs][64]; column 0 is a write-only dummy
    uint32_t regs_wave_bytes;
    uint32_t bitmap;      // tile kernel: u64[nwaves][18], bit c of a wave's map = "all 16 bytes of staged chunk c lie in
                          // the hot interval" (written when the tile is staged, read by the walk to jump over runs)
    uint32_t counter;     // tile kernel: u32, the workgroup's next tile (gx_tile.hip)
    uint32_t stage;       // u8[nwaves][stage_bytes]
    uint32_t stage_bytes; // multiple of 16
    uint32_t nwaves;
    uint32_t total_bytes; // dynamic LDS size to launch with
    // Hot interval: the widest self-loop byte interval of the automata (typically \S+ or .*), in the form the SWAR
    // range test consumes (lo and 0x7F - hi in every byte; hot_k4 = 0x80808080: none, no chunk ever qualifies).
    uint32_t hot_lo4, hot_k4;
    // Record tier (tier == 2; gx_api.cpp: records_from_dense): states are 8-byte range records at LDS address rec,
    // a state is the index of its first record, cmap holds class ids (renumbered), acc_tab[first class of the self
    // range] is the state's self-loop interval word.
    uint32_t tier;        // 0: dense rows in LDS, 1: dense rows in global memory, 2: records in LDS
    uint32_t rec, acc_tab;
    uint32_t rec_indexed; // states with an index >= this keep one record per class (state index + class), behind all the others
    uint32_t sort_lds;    // lane kernel, length-sorted mode: LDS address of u16 perm[sort_chunk] + u32 hist[64] + u32 cursor[64]
    uint32_t sort_chunk;  // ... lines per chunk (0: tiles in input order)
    uint32_t hop_sets;    // hop tier: LDS address of the loop sets (gx_hop.cpp: u8 lo[4], u8 k[4] per entry, entry 0 = none)
    uint32_t fin_unset;   // hop tier: byte offset, from a wave's dummy column, of the column a lane fills with 0xFFFF before it reads its result (the one before it: the line's length); the final records' tags name columns
    uint32_t fin_state_off, fin_state_rec;   // hop tier: the final records by state in the global image (byte offset, 0: none; bytes per record)
};
constexpr uint32_t GX_STEAL_MAX = 3072;         // workgroups of a tile-kernel launch at most (256 CUs x 12)
constexpr uint32_t GX_STEAL_STRIDE = 32;        // u32 words between two workgroups' counters: a cache line each (atomics on ONE line
                                                // take their turns at 11 ns apiece, whichever words of it they want)
constexpr uint32_t GX_BITMAP_WAVE_BYTES = 144;  // 16 x u64 (1024 chunks = 16 KB of staging) + one word read ahead

struct GxBatch {
    const void* data;      // uint8_t (Latin-1) or uint16_t (UTF-16) code units
    const void* offsets;   // uint32_t or uint64_t [n + 1], in code units
    uint64_t n;
    int32_t* match_id;     // [n]
    int32_t* caps;         // [n * 2 * max_groups] or null when match_only
    int32_t* state_out;    // optional [n]: final match-automaton state (for PolyMatcher.match)
    int32_t wide;          // 1: data is uint16_t
    int32_t offsets64;
    int32_t match_only;
    int32_t strip_eol;     // 1: every line carries its terminator ("\n", "\r\n" or "\r"), to be ignored
    // compact result rows (gx_batch_opts.compact_results): per line u16[1 + 2 * max_groups] = int16 match id, then the
    // capture offsets (0xFFFF = unset) -- written instead of match_id / caps when `packed` is set
    uint16_t* packed;
    unsigned long long* overflow;  // with `packed`: += number of offsets above 65534 (stored saturated)
    // narrow rows (compact_results = 2): `packed` holds u8[1 + 2 * max_groups] per line instead -- int8 match id, offsets with
    // 0xFF = unset, an offset above 254 stored as 254 and counted in *overflow
    int32_t narrow;
    // lines the tile kernel cannot stage (longer than its staging area) are left to a follow-up launch of the
    // per-line kernel: the tile kernel stores `seq` into *oversize_flag when it meets one
    uint32_t* oversize_flag;
    uint32_t seq;
    // lane kernel, length-sorted mode: the launch's chunk counter (never reset: chunk = ticket - chunk_base; a launch draws
    // one ticket per chunk and one more per workgroup)
    uint32_t* chunk_ctr;
    uint32_t chunk_base;
    // the host knows that no line of the batch is beyond what the chosen kernel stages (the one-line calls: the host has the line;
    // batches: the caller's promise, gx_batch_opts.max_line_bytes): no follow-up launch of the per-line kernel behind the batch
    // kernel.  oversize_flag then points to a word in pinned HOST memory: a kernel that meets such a line after all says so there.
    uint32_t no_followup;
    uint32_t max_line_bytes;   // the promise itself (0: none)
    uint32_t caller_no_sync;   // host side only: the caller asked for no synchronisation (gx_batch_opts.no_sync): a path that needs one refuses
    // tile kernel: the workgroups' tile counters, u32[2][GX_STEAL_MAX * GX_STEAL_STRIDE] of the launch's stream slot (every GX_STEAL_STRIDE-th word is a counter).  A launch draws from row
    // steal_parity and zeroes the other row, which the stream's next launch draws from.
    uint32_t* steal;
    uint32_t steal_parity;
    // tile kernel on UTF-16 code units (wide = 1): flags[n] (1: the line holds a unit above 0xFF; launch_extract_flagged takes it
    // again) and the word that receives `seq` when there is any such line
    uint8_t* wide_flags;
    uint32_t* wide_any;
};

// More than 64 KiB of dynamic LDS needs the attribute, once per kernel (= per instantiation of this template) and
// per device.
template <typename K>
inline hipError_t allow_full_lds(K kernel) {
    static bool prepared[64] = {};  // (a benign race: setting the attribute twice is harmless)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && prepared[dev]) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (e == hipSuccess && dev >= 0 && dev < 64) prepared[dev] = true;
    return e;
}

// Generic kernel: any table size, any line length, bytes or UTF-16.
hipError_t launch_extract_generic(const GxDev& dev, const GxBatch& b, hipStream_t stream);

// UTF-16 batches: the units' low bytes -> `bytes_at_first_unit` (room for offsets[n] - offsets[0] bytes), flags[i] = 1 for a line
// with a unit above 0xFF; launch_extract_flagged takes those lines again on the code units (per-line walk).
hipError_t launch_narrow_units(const GxBatch& b, uint8_t* bytes_at_first_unit, uint8_t* flags, hipStream_t stream);
// any_word != nullptr: every wave leaves at once unless *any_word == b.seq (the tile kernel's "there was such a line")
hipError_t launch_extract_flagged(const GxDev& dev, const GxBatch& b, const uint8_t* flags, hipStream_t stream, const uint32_t* any_word = nullptr);

// Tile kernel (gx_tile.hip): 64-line tiles staged through LDS with coalesced loads, self-loop runs skipped by SWAR
// tests and a per-chunk bitmap.  Byte input only.  `lds_image` is the device copy of the table image.
// at_global != nullptr selects the L2 tier: automaton rows are read from that global-memory table (row =
// state index, GxLds::row_bytes per row) instead of from LDS.  A line longer than the staging area is not
// processed: the kernel stores b.seq into *b.oversize_flag and launch_extract_oversize (same stream, right after)
// takes those lines with the per-line kernel.  dev_stamps: developer builds (-DGX_DEV) only, else ignored.
hipError_t launch_extract_tile(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                               const GxBatch& b, hipStream_t stream, unsigned long long* dev_stamps);
// by_length 0: the lines for which (bytes in memory) + (address & 15) + 48 > limit (the tile kernel's staging predicate, limit =
// GxLds::stage_bytes); 1: the lines longer than `limit` bytes without their terminator (lane and hop slice kernels: 65 535, or
// 65 534 with compact rows under the lane kernel).
hipError_t launch_extract_oversize(const GxDev& dev, const GxBatch& b, uint32_t limit, int by_length, hipStream_t stream);
// Lane kernel (gx_lanes.hip): tables in global memory (GxLds::tier 1 or 3), every lane keeps its own line in registers;
// lds.nwaves waves per workgroup; lds.regs_wave_bytes = the wave's LDS area (register block of lds.stage_bytes bytes + result rows).
// Lines longer than 65 535 bytes are left to launch_extract_oversize (stage_bytes = 65 535 + 48).
// (with lds.sort_chunk set the launch draws lanes_sorted_tickets(...) tickets from *b.chunk_ctr)
uint32_t lanes_sorted_tickets(uint64_t n, uint32_t chunk_lines, int num_cus);
hipError_t launch_extract_lanes(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                const GxBatch& b, hipStream_t stream, unsigned long long* dev_stamps);
// Slice kernel: the same tables, lines staged 64 bytes at a time (GxLds::stage_bytes = 64 * 80); fused automaton or match only.
hipError_t launch_extract_slices(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                 const GxBatch& b, hipStream_t stream);

// Hop slice kernel: the hop tier's tables (GxLds::tier 4), lines staged GX_HOP_SLICE_BYTES at a time from each lane's own
// position (GxLds::stage_bytes = 64 * (GX_HOP_SLICE_BYTES + 16) + 16); captures only.
#ifndef GX_HOP_SLICE_BYTES
#define GX_HOP_SLICE_BYTES 128u   // (a power of two times 16, at most 1024: one piece is loaded by 64 / (bytes / 16) ... lanes per line)
#endif
hipError_t launch_extract_hop_slices(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                     const GxBatch& b, hipStream_t stream, unsigned long long* stamps = nullptr);
// (with b.chunk_ctr set the launch draws hop_slices_tickets(...) tickets from it: the pool of chunks its waves share)
uint32_t hop_slices_tickets(uint64_t n, uint32_t nwaves, int num_cus);

// The resident one-line service (gx_service.hip): one wave, the dense-rows image in LDS, requests through `mailbox` and answers
// through `answer` (both pinned host memory, device addresses), `state` (pinned): 1 resident, 2 gone.  mode 0: match automaton
// alone, 1: fused automaton with "register := position" programs.
hipError_t launch_one_service(int mode, const GxLds& lds, const uint8_t* lds_image, const uint32_t* mailbox, int32_t* answer, uint32_t* state,
                              uint32_t last_seq, int max_groups, unsigned long long idle_ticks, unsigned long long life_ticks, hipStream_t stream);
constexpr uint32_t GX_SERVICE_MAX_BYTES = 56u + 16u * 60u;   // the longest line a request holds (1 016 bytes)

// Line ingestion (gx_ingest.hip): raw bytes -> CSR offsets of readLine()-style lines, terminators included.
// `workspace` holds split_workspace_bytes(size) bytes; *d_n_lines receives the device address of the line count.
size_t split_workspace_bytes(uint64_t size, bool with_flags = false);
// d_max_line (optional): *d_max_line receives the device address of the longest line's length, terminator included.
// esc_bits (optional, 32-bit offsets, no line flags): 2 048 u16 per 32 KiB of text (whole blocks: ((size + 32767) / 32768) * 4096 bytes), a bit per byte of the text that takes ONE more byte inside a JSON string, and
// the word behind the longest line (d_max_line[1]) != 0 when some byte takes five more (a control character): launch_jsonl_sizes.
hipError_t launch_split_lines(const uint8_t* data, uint64_t size, void* offsets, int offsets64, uint64_t cap_lines, uint8_t* flags,
                              void* workspace, uint64_t** d_n_lines, hipStream_t stream, uint64_t** d_max_line = nullptr,
                              uint16_t* esc_bits = nullptr, int passthrough = 0);


// Result materialisation (gx_jsonl.hip): per-extraction JSON templates on the device.  A template is a list of
// segments; segment s = literal bytes lits[lit_off[s] .. +lit_len[s]) followed by capture group group[s] (-1: none).
struct GxJsonl {
    const uint32_t* seg_off;    // [n_rules + 1]
    const uint32_t* lit_off;    // [n_segs]
    const uint32_t* lit_len;    // [n_segs]
    const int32_t* group;       // [n_segs]
    const uint32_t* fixed_len;  // [n_rules] sum of the template's literal lengths
    const uint8_t* lits;
    uint32_t lits_bytes, n_rules, n_segs;
};
size_t jsonl_workspace_bytes(uint64_t n);
// one line out of / into pinned host memory (gx_kernels.hip: k_extract_one); hipErrorInvalidValue for a line of more than 16 384 code units
hipError_t launch_extract_one(const GxDev& dev, const uint16_t* units, uint32_t len, const GxBatch& b, hipStream_t stream);
// mean_in / mean_out: mean bytes per line of input and of output text; they size the LDS staging of the tile kernels
// esc_bits (optional): launch_split_lines' bits of this batch's text, when its hard word is 0: the sizes come from the bits, not the text
hipError_t launch_jsonl_sizes(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, uint32_t mean_in, uint64_t* line_out_off,
                              void* workspace, hipStream_t stream, const uint32_t* esc_bits = nullptr);
hipError_t launch_count_outcomes(const int32_t* match_id, uint64_t n, unsigned long long* d_counts, hipStream_t stream);
hipError_t launch_pack_results(const int32_t* match_id, const int32_t* caps, uint64_t n, int slots, uint16_t* packed,
                               unsigned long long* d_overflow, hipStream_t stream);
hipError_t launch_unpack_results(const uint16_t* packed, uint64_t n, int slots, int32_t* match_id, int32_t* caps, hipStream_t stream);
hipError_t launch_unpack_results8(const uint8_t* rows, uint64_t n, int slots, int32_t* match_id, int32_t* caps, hipStream_t stream);
hipError_t launch_jsonl_write(const GxJsonl& tm, const GxBatch& b, int slots, int passthrough, uint32_t mean_in, uint32_t mean_out,
                              const uint64_t* line_out_off, uint8_t* out, void* workspace, hipStream_t stream);

}  // namespace gx
