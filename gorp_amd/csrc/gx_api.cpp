// gx_api.cpp -- C ABI of libgorp_hip.so (declared in include/gorp_hip.h).
// Host logic only: compile tables, upload them once, launch kernels, move
// results.  There is no CPU execution path for the hot loops in this library:
// every extract/match entry point runs the HIP kernels or fails with GX_E_DEVICE.
#include <hip/hip_runtime.h>

#include <array>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <atomic>

#include "gx_compile.hpp"
#include "gx_device.hpp"
#include "gx_dsl.hpp"
#include "gx_hop.hpp"

using namespace gx;

namespace {

thread_local std::string g_last_error;
// gx_create_on_devices: the handle whose table images (already in its device's memory) this thread's upload() copies from, device
// to device, instead of from host memory
thread_local const gx_handle* g_peer_src = nullptr;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define GX_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) throw GxError(GX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct Image {
    std::vector<uint8_t> bytes;
    template <typename V> size_t put(const V* p, size_t count) {
        while (bytes.size() % 16) bytes.push_back(0);
        size_t at = bytes.size();
        const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
        bytes.insert(bytes.end(), b, b + count * sizeof(V));
        return at;
    }
    template <typename V> size_t put(const std::vector<V>& v) { return put(v.data(), v.size()); }
};

}  // namespace

struct gx_handle {
    Tables T;
    std::vector<uint8_t> blob;
    bool on_device = false;
    int device = 0;
    void* dimage = nullptr;
    size_t image_bytes = 0;
    GxDev dev{};
    int max_regs = 0;
    // tile kernel: LDS tier (automaton rows in LDS) or L2 tier (rows in global memory, l2_image)
    bool tile_ok = false;
    bool tile_global = false;
    GxLds lds{};                 // table part of the layout; staging is sized per batch
    std::vector<uint8_t> lds_image;
    void* d_lds_image = nullptr;
    // record tier with a fused automaton: the image above holds the fused automaton alone, this one the match
    // automaton alone (PolyMatcher.match batches) -- the two together would not leave LDS for the waves
    bool has_mo = false;
    GxLds lds_mo{};
    std::vector<uint8_t> lds_image_mo;
    void* d_lds_image_mo = nullptr;
    std::vector<uint8_t> l2_image;
    void* d_l2_image = nullptr;
    // hop tier (gx_hop.hpp): a further image of the fused automaton for capture batches of definitions whose dense rows do
    // not fit LDS -- hop records (hot ones in LDS) over dense rows in global memory, walked by the tile kernel
    bool hop_ok = false;
    GxLds lds_hop{}, lds_hop_small{};
    HopImage hop;
    void* d_lds_image_hop = nullptr;
    void* d_lds_image_hop_small = nullptr;
    bool hop_mo_ok = false;             // the same for the match automaton alone (match-only batches)
    GxLds lds_hop_mo{}, lds_hop_mo_small{};
    HopImage hop_mo;
    void* d_lds_image_hop_mo = nullptr;
    void* d_lds_image_hop_mo_small = nullptr;
    void* d_hop_mo_global = nullptr;
    void* d_hop_global = nullptr;
    int num_cus = 256;
    std::vector<dsl::Extraction> meta;  // names / extractor names / append (from definition text or gx_set_extraction_meta)
    void* one_dev = nullptr;     // scratch of the one-String entry points (device) ...
    void* one_host = nullptr;    // ... and its pinned host mirror
    size_t one_cap = 0;
    std::vector<std::vector<std::pair<std::string, std::string>>> append_entries;  // per extraction: (key, value JSON), lazily
    struct JsonlImage { void* d = nullptr; GxJsonl dev{}; };
    std::map<std::string, JsonlImage> jsonl;  // device templates per id_as ("0" = none, "1" + id_as)
    std::mutex mu;  // serialises host-pointer batches that share nothing else
    // Host-pointer batches (what a JNI caller hands over) go through a small pipeline: the batch is cut into chunks of
    // whole lines, and HOST_WORKERS threads, each with its own stream and persistent device buffers, take alternate
    // chunks -- copy in, kernels, copy out -- so that one chunk's copy-in overlaps another's kernels and copy-out on
    // the bus' two directions.  (Threads, not just streams: a copy from pageable memory holds its host thread.)
    static const int HOST_WORKERS = 4;
    struct HostSlot {
        hipStream_t stream = nullptr;
        void* d_bytes = nullptr; size_t cap_bytes = 0;
        void* d_off = nullptr; size_t cap_off = 0;
        void* d_res = nullptr; size_t cap_res = 0;      // match ids, or compact rows
        void* d_caps = nullptr; size_t cap_caps = 0;
        void* d_states = nullptr; size_t cap_states = 0;
        unsigned long long* d_over = nullptr;
    } host_slot[HOST_WORKERS];
    uint32_t create_flags = 0;   // GX_CREATE_* given at creation (kernel choice)
    // Asynchronous batches that give no line_bytes_hint: the mean line length of the previous such batch, read back
    // without a synchronisation (two offsets copied to pinned memory, picked up by the next call once its event is done).
    uint32_t learned_hint = 0;
    uint64_t* hint_probe = nullptr;   // pinned: offsets[0], offsets[n] of the batch being probed
    hipEvent_t hint_event = nullptr;
    bool hint_pending = false;
    uint64_t hint_n = 0;
    bool hint_off64 = false;
    std::mutex hint_mu;
    // Tile-kernel launches that are in flight share nothing but these slots: one word each, into which a launch
    // stores its sequence number when it meets a line it cannot stage (gx_device.hpp: GxBatch::oversize_flag).
    // A slot is reused only after the follow-up kernel of its previous user has run (event).
    // A slot belongs to ONE STREAM for the life of the handle (launches of a stream run in order, so the word is free again
    // when the stream's next launch begins: nothing to wait for, nothing to record); the last slot is shared by the streams
    // that come after N_SLOTS - 1 others and is handed over with an event.
    static const int N_SLOTS = 32;
    uint32_t* d_slots = nullptr;          // [N_SLOTS] oversize flags, then [N_SLOTS] chunk counters of the lane kernel, then [N_SLOTS] "a line
                                          // of this UTF-16 batch holds a unit above 0xFF" words
    hipStream_t slot_stream[N_SLOTS] = {};
    bool slot_taken[N_SLOTS] = {};
    // Batches that promise their longest line (gx_batch_opts.max_line_bytes) have no follow-up launch; their flag word is in
    // pinned host memory (one per slot; the device writes it only if the promise is broken), so that the host can see it.
    uint32_t* h_broken = nullptr;         // [N_SLOTS], pinned + mapped
    uint32_t* d_broken = nullptr;         // the device's address of the same words
    uint32_t* d_steal[N_SLOTS] = {};      // per slot, at its first tile launch: [2][GX_STEAL_MAX * GX_STEAL_STRIDE], the tile kernel's workgroup counters (GxBatch::steal)
    uint32_t steal_parity[N_SLOTS] = {};  // the row the slot's next tile-kernel launch draws from
    uint32_t promise_seq[N_SLOTS] = {};   // the sequence number of the slot's last launch under a promise (0: none)
    uint32_t broken_seen[N_SLOTS] = {};   // the slot's pinned word as the host last saw it: ANY other value is a promise that broke since --
                                          // whichever of the stream's batches it was, however many have been enqueued behind it
    std::atomic<uint64_t> promises_broken{0};
    // the resident one-line service (gx_service.hip; GX_CREATE_RESIDENT_ONE)
    struct Service {
        bool enabled = false;
        int mode = 0;
        GxLds L{};
        hipStream_t stream = nullptr;
        uint32_t* host = nullptr;      // pinned block: mailbox [17 x 16 dwords] | answer [2 + 2 G dwords, padded] | state
        uint32_t* dev = nullptr;       // the device's address of the same block
        uint32_t seq = 0;
        bool started = false;
        uint64_t launches = 0;
    } svc;
    int32_t* d_pike_scratch = nullptr;    // thread lists of the lanes that run an extraction's program as it is (GxDev::pike_scratch)
    // ... ONE set of them per handle, a lane's area named by its place in the grid: two per-line kernels of the handle must not run at
    // once (the host pipeline's four streams, a caller's streams, the one-String calls beside a batch).  Every launch of such a
    // handle waits for the one before it, on whatever stream that was (PikeGate).
    std::recursive_mutex pike_mu;
    hipEvent_t pike_event = nullptr;
    bool pike_event_set = false;
    int pike_depth = 0;
    int hop_reason = 4;                   // why capture batches have no hop tables (gx_stat(h, 26); 0: they have)
    hipStream_t multi_stream = nullptr;   // gx_extract_batch_multi_device: the stream of shards that bring none
    hipStream_t gather_stream = nullptr;  // gx_gather_rows: this handle's rows leave for the root's device on it (a copy queue of its own: seven peers, seven links)
    hipEvent_t gather_event = nullptr;    // ... "the shard's kernel is done", recorded on the kernel's stream
    size_t peer_image_bytes = 0;          // table bytes that came from another device's copy (gx_create_on_devices; gx_stat(h, 30))
    std::atomic<int> last_kernel{0};      // GX_KERNEL_* of the most recent batch launch (gx_stat(h, 25))
    // device scratch of gx_results_to_jsonl / gx_text_to_jsonl (sizes, split points, line offsets), kept between calls and grown as
    // batches ask: a hipMalloc + hipFree pair per call cost more than the scan kernels.  Used under `mu` only, and every call that
    // uses it ends with a stream synchronisation.
    void* scratch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // stream-ordered memory of the UTF-16 batch path (the narrowed copy of a batch): a pool of the handle's own that keeps what a
    // batch frees for the next one (the device's default pool gives everything back at the next synchronisation: an allocation of
    // gigabytes per call, 0.6 of that path's 2.4 ms per 10 M lines)
    hipMemPool_t pool = nullptr;
    uint32_t chunk_tickets[N_SLOTS] = {};  // what each chunk counter will read when the next launch on its slot begins
    hipEvent_t shared_event = nullptr;    // the shared slot's "previous user is done"
    bool shared_used = false;
    uint32_t next_seq = 1;
    std::mutex slot_mu;
#ifdef GX_DEV
    unsigned long long* dev_stamps = nullptr;  // developer build: device buffer for the tile kernel's phase cycle counts
#endif
};

namespace {

const uint32_t LDS_BYTES = 163840;     // 160 KiB per CU on gfx950
const uint32_t LDS_TABLE_BUDGET = 96 * 1024;

// Longest run of ASCII byte values on which `loops(b)` holds, as lo | (0x7F - hi) << 8 (0x8000 = none): the
// form the kernel's SWAR range test consumes.  Of equally long runs the later one wins: for \w that is a-z rather
// than A-Z, and log text is mostly lower case.
template <typename F> uint16_t self_loop_interval(F loops) {
    int best_lo = 0, best_len = 0, run_lo = 0, run = 0;
    for (int b = 0; b < 128; ++b) {
        if (loops(b)) { if (run == 0) run_lo = b; ++run; if (run >= best_len) { best_len = run; best_lo = run_lo; } }
        else run = 0;
    }
    if (best_len < 4) return 0x8000;
    return static_cast<uint16_t>(best_lo | ((0x7F - (best_lo + best_len - 1)) << 8));
}

// ---- record tier ------------------------------------------------------------------------------------------
// A state's dense row, as a handful of class ranges.  Classes are renumbered first so that the sets the rows use
// (the classes of one successor: "digits", "\w", "not a blank", single literals) become contiguous id ranges
// wherever one ordering can serve them all (greedy partition refinement, heaviest sets first); a set that stays
// split simply takes several ranges.  A state is then
//     [header: 8 bytes, only when its info word is not -1]
//     record 0:  self range (plain self-loop, no program) + one exit range with its successor and program
//     record 1:  one more exit range (record 0 says that it exists)
// or, with three ranges and more besides the self range (the branching nodes of a trie of literals), one record per
// class, behind all the others: the walk reads record [state + class].  Every class in no range leads to the dead
// state, which is record 0 of the image for all its automata (a dead state has no successors and accepts nothing, so
// they are all alike).  A record is two dwords:
//     w0 = self_lo | self_span << 8 | exit_lo << 16 | exit_span << 24      (class c is inside iff c - lo <= span, unsigned;
//                                                                           lo = 255, span = 0: empty)
//     w1 = successor (record index, 16 bits) | program << 16 (8 bits) | flags << 24 (0x80: a second record follows,
//          2: a header precedes)
// A state's index is the index of its record 0; the self-loop byte interval (GxLds rows' ACC word) is looked up by the
// self range's first class in a small table.  The class map has 32-bit entries, class | class * 8 << 16 (the walk's
// range tests take the low byte, its address arithmetic the upper half).  The builder checks its own output against
// the dense rows, class by class, before it is used.
constexpr uint32_t REC_MORE = 0x80u << 24, REC_HDR = 2u << 24, REC_IDC = 254u, REC_EMPTY = 255u, REC_AT = 1088u;

bool records_from_dense(gx_handle* h, const std::vector<uint32_t>& at, size_t rows, uint32_t cols, const std::vector<uint32_t>& dead_of_row,
                        const std::vector<std::array<uint64_t, 2>>& loop_set, bool simple, Image& img, GxLds& L, std::vector<uint32_t>& c_rule,
                        std::vector<uint32_t>* items_global) {
    const Tables& T = h->T;
    const int ncls = T.ncls;
    const uint32_t ACC = ncls + 1, INFO = ncls + 2;
    auto has = [](const ClassSet& s, int c) { return (s[c >> 6] >> (c & 63)) & 1ull; };
    // the groups of every row: successor entry -> classes (the dead default is not a group)
    struct Group { uint32_t entry; ClassSet set; };
    std::vector<std::vector<Group>> groups(rows);
    std::map<ClassSet, uint64_t> weight;
    for (size_t r = 0; r < rows; ++r) {
        std::map<uint32_t, ClassSet> by_entry;
        for (int c = 0; c < ncls; ++c) {
            const uint32_t e = at[r * cols + c];
            if (e == dead_of_row[r]) continue;
            by_entry[e][c >> 6] |= 1ull << (c & 63);
        }
        for (auto& g : by_entry) {
            // the plain self-loop first: it takes item 0's self slot
            if (g.first == static_cast<uint32_t>(r)) groups[r].insert(groups[r].begin(), Group{g.first, g.second});
            else groups[r].push_back(Group{g.first, g.second});
            ++weight[g.second];
        }
    }
    // class order: as many of the sets as possible become id ranges (gx_hop.cpp: order_classes)
    const std::vector<int> new_id = order_classes(weight, ncls);
    // ranges (in new ids) of a class set
    auto ranges_of = [&](const ClassSet& S) {
        std::vector<char> in(ncls, 0);
        for (int c = 0; c < ncls; ++c) if (has(S, c)) in[new_id[c]] = 1;
        std::vector<std::pair<int, int>> out;
        for (int i = 0; i < ncls; ++i) if (in[i]) { int j = i; while (j + 1 < ncls && in[j + 1]) ++j; out.push_back({i, j}); i = j; }
        return out;
    };
    // pass 1: items per row -> indexes
    struct Slot { int lo, hi; uint32_t entry; };
    std::vector<std::vector<Slot>> exits(rows);   // everything but the first range of the plain self-loop
    std::vector<std::pair<int, int>> self0(rows, {-1, -1});
    std::vector<uint32_t> index_of(rows);
    size_t n_items = 0;
    L.rec_indexed = 0xFFFFFFFFu;  // (first index of the class-indexed states; none)
    // which exit shares item 0 with the self range decides how often a lane needs a second record: the one that
    // printable text takes most likely first (a field's blank before some control character's odd successor)
    std::vector<int> printable(ncls, 0), bytes_of(ncls, 0);
    for (int b = 0; b < 256; ++b) { ++bytes_of[T.cls256[b]]; if (b >= 0x20 && b < 0x7F) ++printable[T.cls256[b]]; }
    auto likelihood = [&](const ClassSet& S) {
        long p = 0, n = 0;
        for (int c = 0; c < ncls; ++c) if (has(S, c)) { p += printable[c]; n += bytes_of[c]; }
        return p * 1000 + n;
    };
    for (size_t r = 0; r < rows; ++r) {
        std::stable_sort(groups[r].begin(), groups[r].end(), [&](const Group& a, const Group& b) {
            const bool sa = a.entry == static_cast<uint32_t>(r), sb = b.entry == static_cast<uint32_t>(r);
            if (sa != sb) return sa;  // the plain self-loop stays first
            return likelihood(a.set) > likelihood(b.set);
        });
        for (auto& g : groups[r]) {
            auto rs = ranges_of(g.set);
            size_t from = 0;
            if (g.entry == static_cast<uint32_t>(r) && self0[r].first < 0) { self0[r] = rs[0]; from = 1; }
            for (size_t q = from; q < rs.size(); ++q) exits[r].push_back(Slot{rs[q].first, rs[q].second, g.entry});
        }
    }
    // States with three or more records (the branching nodes of a trie of literals, the start state) are laid out
    // INDEXED BY CLASS instead: one record per class, behind all the others, so that the walk finds the class's
    // successor with its first read (state index + class; a state's index says which layout it has).
    std::vector<char> indexed(rows, 0);
    // the dead states: every class leads back to the state itself (no group above), nothing accepted; they share record 0
    auto pure_dead = [&](size_t r) { return dead_of_row[r] == r && groups[r].empty() && at[r * cols + INFO] == 0xFFFFFFFFu; };
    for (size_t r = 0; r < rows; ++r)
        if (dead_of_row[r] >= rows || !pure_dead(dead_of_row[r])) return false;  // (every automaton the compiler emits has one)
    n_items = 1;
    for (int pass = 0; pass < 2; ++pass)
        for (size_t r = 0; r < rows; ++r) {
            if (pure_dead(r)) { index_of[r] = 0; continue; }
            const bool ix = exits[r].size() >= 3;
            indexed[r] = ix;
            if (ix != (pass == 1)) continue;
            const bool hdr = at[r * cols + INFO] != 0xFFFFFFFFu;
            if (pass == 1 && L.rec_indexed == 0xFFFFFFFFu) L.rec_indexed = static_cast<uint32_t>(n_items);
            if (hdr) ++n_items;
            index_of[r] = static_cast<uint32_t>(n_items);
            n_items += ix ? static_cast<size_t>(ncls) : std::max<size_t>(1, exits[r].size());
        }
    const bool wide = items_global != nullptr;  // records in global memory: 22-bit successors
    if (n_items > (wide ? 0x3FFFFFu : 65535u)) return false;
    auto pack_w1 = [&](uint32_t target, uint32_t op, bool more, bool hdr) {
        return wide ? (target | op << 22 | (hdr ? 1u << 30 : 0u) | (more ? 1u << 31 : 0u)) : (target | op << 16 | (hdr ? REC_HDR : 0u) | (more ? REC_MORE : 0u));
    };
#ifdef GX_DEV
    if (getenv("GX_REC_STATS")) {
        size_t multi = 0, self_split = 0, self_states = 0;
        std::map<size_t, size_t> hist;
        for (size_t r = 0; r < rows; ++r) {
            ++hist[std::max<size_t>(1, exits[r].size())];
            if (exits[r].size() > 1) ++multi;
            if (self0[r].first >= 0) { ++self_states; for (auto& e : exits[r]) if (e.entry == static_cast<uint32_t>(r)) { ++self_split; break; } }
        }
        fprintf(stderr, "records: rows %zu items %zu multi-item states %zu self-loop states %zu of which split %zu\n", rows, n_items, multi, self_states, self_split);
        for (auto& hh : hist) fprintf(stderr, "  %zu items: %zu states\n", hh.first, hh.second);
    }
#endif
    // pass 2: emit
    std::vector<uint32_t> items(2 * n_items + 2, 0);  // (+ one: the second-record pass reads record [state + 1] of every state)
    items[0] = REC_EMPTY | (REC_EMPTY << 16);  // record 0: the dead state
    for (size_t r = 0; r < rows; ++r) {
        if (pure_dead(r)) continue;
        const uint32_t info = at[r * cols + INFO];
        const bool hdr = info != 0xFFFFFFFFu;
        if (hdr) { items[2 * (index_of[r] - 1)] = 0; items[2 * (index_of[r] - 1) + 1] = info; }
        auto op_field = [&](uint32_t entry, uint32_t& op) {  // the program as the records carry it
            op = entry >> 16;
            if (simple) op /= 128u;       // register + 1
            else if (op & 0x8000u) { if ((op & 0x7FFFu) > 127u) return false; op = 0x80u | (op & 0x7Fu); }
            else if (op > 127u) return false;
            return op <= 255u;
        };
        if (indexed[r]) {
            for (int c = 0; c < ncls; ++c) {  // record new_id[c]: class c's successor as a one-class exit (none: dead)
                const uint32_t id = static_cast<uint32_t>(new_id[c]), e = at[r * cols + c];
                uint32_t w0 = REC_EMPTY | (REC_EMPTY << 16), target = 0, op = 0;
                if (e != dead_of_row[r]) {
                    if (!op_field(e, op)) return false;
                    w0 = REC_EMPTY | (id << 16);
                    target = index_of[e & 0xFFFFu];
                }
                const uint32_t w1 = pack_w1(target, op, false, id == 0 && hdr);
                items[2 * (index_of[r] + id)] = w0;
                items[2 * (index_of[r] + id) + 1] = w1;
            }
            continue;
        }
        const size_t n = std::max<size_t>(1, exits[r].size());
        for (size_t q = 0; q < n; ++q) {
            uint32_t w0 = REC_EMPTY | (REC_EMPTY << 16), target = 0, op = 0;
            if (q == 0 && self0[r].first >= 0) w0 = (w0 & 0xFFFF0000u) | self0[r].first | (static_cast<uint32_t>(self0[r].second - self0[r].first) << 8);
            if (q < exits[r].size()) {
                const Slot& e = exits[r][q];
                w0 = (w0 & 0xFFFFu) | (static_cast<uint32_t>(e.lo) << 16) | (static_cast<uint32_t>(e.hi - e.lo) << 24);
                if (!op_field(e.entry, op)) return false;
                target = index_of[e.entry & 0xFFFFu];
            }
            const uint32_t w1 = pack_w1(target, op, q + 1 < n, q == 0 && hdr);
            items[2 * (index_of[r] + q)] = w0;
            items[2 * (index_of[r] + q) + 1] = w1;
        }
    }
    // self-loop interval words by the first class of the self range; states that share it must agree (else the
    // narrowest claim that is true for all of them: none)
    std::vector<uint32_t> acc_tab(ncls + 1, 0x8000u);
    {
        std::vector<std::vector<size_t>> by_lo(ncls);
        for (size_t r = 0; r < rows; ++r) if (self0[r].first >= 0) by_lo[self0[r].first].push_back(r);
        auto covers = [&](size_t r, int lo, int hi) {
            for (int b = lo; b <= hi; ++b) if (!(loop_set[r][b >> 6] >> (b & 63) & 1ull)) return false;
            return true;
        };
        for (int lo_c = 0; lo_c < ncls; ++lo_c) {
            uint32_t best = 0x8000u;
            int best_w = 0;
            for (size_t r : by_lo[lo_c]) {
                const uint32_t a = at[r * cols + ACC];
                if ((a & 0xFFFFu) == 0x8000u) continue;
                const int lo = static_cast<int>(a & 0xFFu), hi = 0x7F - static_cast<int>((a >> 8) & 0xFFu);
                bool all = true, hot = true;
                for (size_t q : by_lo[lo_c]) { all = all && covers(q, lo, hi); hot = hot && (at[q * cols + ACC] & 0x10000u); }
                if (all && hi - lo + 1 > best_w) { best_w = hi - lo + 1; best = (a & 0xFFFFu) | (hot ? 0x10000u : 0u); }
            }
            acc_tab[lo_c] = best;
        }
    }
    // check: every (row, class) decodes to the dense entry
    for (size_t r = 0; r < rows; ++r)
        for (int c = 0; c < ncls; ++c) {
            const uint32_t id = static_cast<uint32_t>(new_id[c]);
            uint32_t next = 0, op = 0;
            for (uint32_t q = index_of[r] + (index_of[r] >= L.rec_indexed ? id : 0u);; ++q) {
                const uint32_t w0 = items[2 * q], w1 = items[2 * q + 1];
                if (id - ((w0 >> 16) & 0xFFu) <= (w0 >> 24)) { next = wide ? (w1 & 0x3FFFFFu) : (w1 & 0xFFFFu); op = wide ? ((w1 >> 22) & 0xFFu) : ((w1 >> 16) & 0xFFu); break; }
                if (id - (w0 & 0xFFu) <= ((w0 >> 8) & 0xFFu)) { next = index_of[r]; break; }
                if (!(w1 & (wide ? 1u << 31 : REC_MORE))) break;
            }
            const uint32_t e = at[r * cols + c];
            uint32_t want_op = e >> 16;
            if (simple) want_op /= 128u; else if (want_op & 0x8000u) want_op = 0x80u | (want_op & 0x7Fu);
            if (next != index_of[e & 0xFFFFu] || op != want_op) throw GxError(GX_E_ARG, "internal: record tier does not reproduce the dense rows");
        }
    // image: class map (new ids; entry 256 = the identity class of masked bytes), interval table, items
    std::vector<uint32_t> cmap(REC_AT / 4, REC_IDC);
    for (int b = 0; b < 256; ++b) { const uint32_t id = static_cast<uint32_t>(new_id[T.cls256[b]]); cmap[b] = id | (id * 8u) << 16; }
    L.cmap = static_cast<uint32_t>(img.put(cmap));
    if (items_global) { L.rec = 0; items_global->swap(items); }  // records in global memory (tier 3)
    else { L.rec = static_cast<uint32_t>(img.put(items)); if (L.rec != REC_AT) throw GxError(GX_E_ARG, "internal: record image layout"); }
    L.acc_tab = static_cast<uint32_t>(img.put(acc_tab));
    L.at = 0;
    if (L.m_dead < rows) { L.m_start = index_of[L.m_start]; L.m_dead = index_of[L.m_dead]; }  // (absent from a capture-only image)
    if (L.u_start != 0xFFFFFFFFu) { L.u_start = index_of[L.u_start]; L.u_dead = index_of[L.u_dead]; }
    else for (size_t k = 0; k + 1 < c_rule.size(); k += 2) {
        c_rule[k] = index_of[c_rule[k]];  // per-extraction capture automata: the start states
    }
    return true;
}

// Build the table image of the tile kernel (layout: GxLds).
// tier 0: LDS tier, everything in one LDS-resident image, dense rows addressed by byte offset.
// tier 1: L2 tier, the dense rows go to a separate global-memory image (h->l2_image: match rows at 0, capture rows
//         at GxLds::c_base) and are addressed by state index; LDS keeps only the byte->class map, the per-extraction
//         start rows and the capture programs.
// tier 2: record tier, for automata whose dense rows do not fit LDS but whose states are sparse (a literal chain
//         link has one live class, a field state one self range and one exit): every state becomes a few 8-byte
//         range records in LDS (records_from_dense below; ~8.5 bytes per state instead of 4 * classes).
// tier 3: the same records in global memory (h->l2_image), where 64-512 KB of them live in the vector L1 / L2 caches.
bool records_from_dense(gx_handle* h, const std::vector<uint32_t>& at, size_t rows, uint32_t cols, const std::vector<uint32_t>& dead_of_row,
                        const std::vector<std::array<uint64_t, 2>>& loop_set, bool simple, Image& img, GxLds& L, std::vector<uint32_t>& c_rule,
                        std::vector<uint32_t>* items_global);

// part 0: match automaton + capture automata in one image; 1: capture side alone (fused automaton); 2: match alone.
bool build_tile_image(gx_handle* h, int tier, int part = 0) {
    const Tables& T = h->T;
    const bool global = tier == 1;
    const bool in_global = tier == 1 || tier == 3;  // automaton tables and final records in h->l2_image
    if (part != 2 && tier == 3) h->l2_image.clear();
    if (part != 2) {
        h->tile_ok = false;
        h->tile_global = in_global;
        h->has_mo = false;
    }
    if (T.n_rules > 32767) return false;
    const uint32_t cols = static_cast<uint32_t>(T.ncls) + 3u;
    const uint32_t RS = cols * 4u;
    // with the fused automaton present the per-extraction capture rows are not needed on the device
    // (GX_CREATE_NO_FUSED forces the two-pass layout, which otherwise only very large definitions get)
    const bool fused = T.union_ok && !(h->create_flags & GX_CREATE_NO_FUSED);
    if (part == 1 && !(fused && T.has_capture)) return false;  // the two-pass layout needs both sides
    const size_t m_rows = part == 1 ? 0 : static_cast<size_t>(T.m_states);
    size_t c_rows = 0;
    if (part == 2) c_rows = 0;
    else if (fused) c_rows = T.uni.n_states;
    else for (auto& r : T.rules) c_rows += r.n_states;
    const size_t rows = m_rows + c_rows;
    if (T.ncls > 252) return false;  // keeps the column offsets of a row (class * 4, + 3 extra columns) below 1024
    const uint32_t AT = 544;         // LDS tier: the rows follow the class map (u16[256] + the identity entry, padded)
    if (tier >= 2) {
        // a dense entry carries its successor's row in 16 bits; records in LDS have 16-bit successors too (in global memory: 22)
        if (rows > (tier == 3 ? 65536u : 65000u) || T.ncls > 250) return false;  // class ids 254 / 255 are reserved
    } else if (!global) {
        if (AT + rows * RS > 65536u) return false;  // successors are 16-bit LDS addresses
        if (rows * RS + T.ops_off.size() * 4 + T.ops.size() * 2 + 1024 > LDS_TABLE_BUDGET) return false;  // (+ the final records, below)
    } else {
        if (m_rows > 65536u || c_rows > 65536u) return false;  // state indexes are 16-bit per automaton table
        if (T.n_rules * 8 + T.ops_off.size() * 4 + T.ops.size() * 2 + 1024 > LDS_TABLE_BUDGET) return false;
    }
    // a state's successor field: LDS tier = LDS address of the row, L2 tier = state index
    const uint32_t UNIT = tier ? 1u : RS;
    const uint32_t ORG = tier ? 0u : AT;
    std::vector<uint32_t> dead_of_row(rows, 0);  // tier 2: the row every unlisted class of a row leads to

    Image img;
    GxLds L{};
    std::vector<uint16_t> cmap(272, static_cast<uint16_t>(T.ncls * 4));  // entry 256: the identity column, for bytes outside the line
    for (int b = 0; b < 256; ++b) cmap[b] = static_cast<uint16_t>(T.cls256[b] * 4);  // byte offset of the class's column
    L.cmap = static_cast<uint32_t>(img.put(cmap));  // offset 0
    L.ncls = static_cast<uint32_t>(T.ncls);
    L.row_bytes = RS;
    std::vector<uint32_t> at(rows * cols, 0);
    const uint32_t IDC = T.ncls, ACC = T.ncls + 1, INFO = T.ncls + 2;
    // per row: the ASCII bytes the state loops on (with no capture program), for the choice of the hot interval
    std::vector<std::array<uint64_t, 2>> loop_set(rows, std::array<uint64_t, 2>{0, 0});
    auto loop_interval = [&](size_t row_index, auto loops) {
        for (int b = 0; b < 128; ++b) if (loops(b)) loop_set[row_index][b >> 6] |= 1ull << (b & 63);
        return self_loop_interval(loops);
    };
    // match automaton rows
    for (int s = 0; s < static_cast<int>(m_rows); ++s) {
        uint32_t* row = &at[static_cast<size_t>(s) * cols];
        for (int c = 0; c < T.ncls; ++c) row[c] = ORG + T.m_next[static_cast<size_t>(s) * T.ncls + c] * UNIT;
        row[IDC] = ORG + static_cast<uint32_t>(s) * UNIT;
        row[ACC] = loop_interval(static_cast<size_t>(s), [&](int b) { return T.m_next[static_cast<size_t>(s) * T.ncls + T.cls256[b]] == static_cast<uint32_t>(s); });
        row[INFO] = static_cast<uint32_t>(T.m_accept_first[s]);
        dead_of_row[s] = static_cast<uint32_t>(T.m_dead);
    }
    L.m_start = ORG;
    L.m_dead = ORG + static_cast<uint32_t>(T.m_dead) * UNIT;
    L.c_base = global ? static_cast<uint32_t>(m_rows * RS) : 0u;
    // capture automata rows: the fused automaton, or one automaton per extraction
    std::vector<uint32_t> c_rule;
    size_t base_row = m_rows;
    bool too_many_programs = false;
    // are all capture programs of the tables we ship "one register := position"?
    auto is_single_set = [&](uint32_t op) {
        const uint32_t b = T.ops_off[op], e = T.ops_off[op + 1];
        return e - b == 1 && T.ops[2 * b + 1] == GX_SRC_POS && T.ops[2 * b] < 500;
    };
    bool simple = true;
    auto scan_simple = [&](const RuleTables& r) {
        for (uint32_t w : r.trans) if ((w >> 16) && !is_single_set(w >> 16)) simple = false;
    };
    if (part == 2) simple = true;
    else if (fused) scan_simple(T.uni);
    else for (auto& r : T.rules) scan_simple(r);
    L.simple_ops = simple ? 1u : 0u;
    // Final records, one per distinct (final tag list, extraction): u16 [begin tag, end tag] x max_groups padded to a
    // multiple of four groups (16 bytes), then the extraction and padding to the next 16 bytes; the tags as
    // line_result (gx_walk.hpp) wants them: 0 = unset, 1 = the line length, else the byte offset of the register's
    // column from the dummy column.  Record 0 = "no groups" for the lines that match nothing.  A row's info word is
    // the byte offset of its record.
    const size_t tag_slots = 8 * static_cast<size_t>((T.max_groups + 3) / 4), rec_len = tag_slots + 8;
    std::vector<uint16_t> fin_rec(rec_len, 0);
    fin_rec[tag_slots] = 0xFFFFu;
    std::map<std::pair<int32_t, int32_t>, uint32_t> rec_of;
    auto fin_record = [&](int32_t f, int32_t k_or_minus1) -> uint32_t {  // k < 0: the list starts with the extraction
        auto it = rec_of.find({f, k_or_minus1});
        if (it != rec_of.end()) return it->second;
        const int32_t k = k_or_minus1 >= 0 ? k_or_minus1 : static_cast<int32_t>(T.fin_tags[f]);
        const size_t t0 = k_or_minus1 >= 0 ? f : f + 1;
        const size_t at = fin_rec.size();
        fin_rec.resize(at + rec_len, 0);
        for (int g = 0; g < T.rules[k].n_groups; ++g)
            for (int e = 0; e < 2; ++e) {
                const uint16_t v = T.fin_tags[t0 + 2 * g + e];
                fin_rec[at + 2 * g + e] = v == GX_SRC_NIL ? 0 : v == GX_SRC_POS ? 1 : static_cast<uint16_t>((v + 1u) * 128u);
            }
        fin_rec[at + tag_slots] = static_cast<uint16_t>(k);
        rec_of[{f, k_or_minus1}] = static_cast<uint32_t>(at * 2);
        return static_cast<uint32_t>(at * 2);
    };
    int rule_being_emitted = -1;  // per-extraction capture automata: the rule; fused automaton: -1
    auto emit_rows = [&](const RuleTables& r) {
        // LDS tier: LDS addresses; L2 tier: state indexes within the capture rows; record tier: indexes over all rows
        const uint32_t base = ORG + static_cast<uint32_t>(global ? base_row - m_rows : base_row) * UNIT;
        for (int s = 0; s < r.n_states; ++s) {
            uint32_t* row = &at[(base_row + s) * cols];
            dead_of_row[base_row + s] = base + static_cast<uint32_t>(r.dead);
            for (int c = 0; c < T.ncls; ++c) {
                const uint32_t w = r.trans[static_cast<size_t>(s) * T.ncls + c];
                uint32_t op = w >> 16;
                if (simple) {
                    // byte offset of the register's column in the wave's register block: 0 = the dummy column
                    // ("no program"), (r + 1) * 128 = register r
                    op = op ? (T.ops[2 * T.ops_off[op]] + 1u) * 128u : 0u;
                } else if (op) {
                    // the common capture program "one register := position" is folded into the entry as 0x8000 | register
                    if (is_single_set(op)) op = 0x8000u | T.ops[2 * T.ops_off[op]];
                    else if (op >= 0x8000u) too_many_programs = true;
                }
                row[c] = (base + (w & 0xFFFFu) * UNIT) | (op << 16);
            }
            row[IDC] = base + static_cast<uint32_t>(s) * UNIT;
            row[ACC] = loop_interval(base_row + s,
                [&](int b) { return r.trans[static_cast<size_t>(s) * T.ncls + T.cls256[b]] == static_cast<uint32_t>(s); });
            row[INFO] = r.fin[s] >= 0 ? fin_record(r.fin[s], rule_being_emitted) : static_cast<uint32_t>(r.fin[s]);
        }
        base_row += r.n_states;
        return base;
    };
    L.u_start = 0xFFFFFFFFu;
    L.u_dead = 0xFFFFFFFFu;
    if (part == 2) {
        // match automaton alone
    } else if (fused) {
        const uint32_t base = emit_rows(T.uni);
        L.u_start = base;
        L.u_dead = base + static_cast<uint32_t>(T.uni.dead) * UNIT;
        for (auto& r : T.rules) { c_rule.push_back(0); c_rule.push_back(static_cast<uint32_t>(r.n_groups)); }
    } else {
        for (auto& r : T.rules) {
            rule_being_emitted = static_cast<int>(&r - &T.rules[0]);
            const uint32_t base = emit_rows(r);
            c_rule.push_back(base);
            c_rule.push_back(static_cast<uint32_t>(r.n_groups));
        }
    }
    if (too_many_programs) return false;  // too many distinct general programs for the 15-bit program field
    if (fin_rec.size() * 2 > 0xFFFFFFu) return false;
    if (tier == 0 && rows * RS + T.ops_off.size() * 4 + T.ops.size() * 2 + fin_rec.size() * 2 + 1024 > LDS_TABLE_BUDGET) return false;
    // Hot interval: among the self-loop intervals of all rows, the one that promises the longest skips -- width
    // squared (only runs of several 16-byte chunks pay off) times the number of states that loop on all of it.
    // Those states get bit 16 of their interval column; the tile kernel marks the staged chunks that lie inside the
    // interval and lets such a state jump over runs of them (gx_tile.hip).
    {
        auto covers = [&](size_t r, int lo, int hi) {
            for (int b = lo; b <= hi; ++b) if (!(loop_set[r][b >> 6] >> (b & 63) & 1ull)) return false;
            return true;
        };
        std::map<uint32_t, int> candidates;
        for (size_t r = 0; r < rows; ++r) if (at[r * cols + ACC] != 0x8000u) candidates[at[r * cols + ACC]] = 0;
        double best = 0;
        int best_lo = 0, best_hi = -1;
        for (auto& c : candidates) {
            const int lo = static_cast<int>(c.first & 0xFFu), hi = 0x7F - static_cast<int>((c.first >> 8) & 0xFFu);
            if (hi - lo + 1 < 16) continue;  // a run of such bytes seldom fills whole chunks
            int states = 0;
            for (size_t r = 0; r < rows; ++r) if (covers(r, lo, hi)) ++states;
            const double score = static_cast<double>(hi - lo + 1) * (hi - lo + 1) * states;
            if (score > best) { best = score; best_lo = lo; best_hi = hi; }
        }
        L.hot_lo4 = 0;
        L.hot_k4 = 0x80808080u;
        if (best_hi >= best_lo) {
            L.hot_lo4 = static_cast<uint32_t>(best_lo) * 0x01010101u;
            L.hot_k4 = static_cast<uint32_t>(0x7F - best_hi) * 0x01010101u;
            for (size_t r = 0; r < rows; ++r) if (covers(r, best_lo, best_hi)) at[r * cols + ACC] |= 0x10000u;
        }
    }
    if (c_rule.empty()) { c_rule.push_back(0); c_rule.push_back(0); }
    // tier 3: one global image per handle: [capture-side records + final records][match-only records]; GxLds::rec_g /
    // fin_tags are byte offsets into it
    const size_t g_base = h->l2_image.size();
    if (tier >= 2) {
        img.bytes.clear();  // the record tiers have their own class map (classes renumbered so that sets become ranges)
        std::vector<uint32_t> items_g;
        if (!records_from_dense(h, at, rows, cols, dead_of_row, loop_set, simple, img, L, c_rule, tier == 3 ? &items_g : nullptr)) return false;
        if (tier == 2 && img.bytes.size() + T.ops_off.size() * 4 + T.ops.size() * 2 + fin_rec.size() * 2 + 1024 > LDS_TABLE_BUDGET) return false;
        if (tier == 3) {
            L.c_base = static_cast<uint32_t>(g_base);  // (byte offset of this image's records in the global image)
            const uint8_t* ib = reinterpret_cast<const uint8_t*>(items_g.data());
            h->l2_image.insert(h->l2_image.end(), ib, ib + items_g.size() * 4);
            while (h->l2_image.size() % 16) h->l2_image.push_back(0);
            L.fin_tags = static_cast<uint32_t>(h->l2_image.size());
            const uint8_t* fr = reinterpret_cast<const uint8_t*>(fin_rec.data());
            h->l2_image.insert(h->l2_image.end(), fr, fr + fin_rec.size() * 2);
            while (h->l2_image.size() % 16) h->l2_image.push_back(0);
        }
    } else if (global) {
        L.at = 0;
        h->l2_image.assign(reinterpret_cast<const uint8_t*>(at.data()), reinterpret_cast<const uint8_t*>(at.data() + at.size()));
        while (h->l2_image.size() % 16) h->l2_image.push_back(0);
        L.fin_tags = static_cast<uint32_t>(h->l2_image.size());  // L2 tier: the final records follow the rows in global memory
        const uint8_t* fr = reinterpret_cast<const uint8_t*>(fin_rec.data());
        h->l2_image.insert(h->l2_image.end(), fr, fr + fin_rec.size() * 2);
    } else {
        L.at = static_cast<uint32_t>(img.put(at));
        if (L.at != AT) throw GxError(GX_E_ARG, "internal: LDS table image layout");
    }
    L.c_rule = static_cast<uint32_t>(img.put(c_rule));
    L.ops_off = static_cast<uint32_t>(img.put(T.ops_off));
    std::vector<uint16_t> ops = T.ops;
    if (ops.empty()) ops.push_back(0);
    L.ops = static_cast<uint32_t>(img.put(ops));
    if (!in_global) L.fin_tags = static_cast<uint32_t>(img.put(fin_rec));
    L.tier = static_cast<uint32_t>(tier);
    while (img.bytes.size() % 16) img.bytes.push_back(0);
    L.table_bytes = static_cast<uint32_t>(img.bytes.size());
    int max_regs = 0;
    for (auto& r : T.rules) max_regs = std::max(max_regs, r.n_regs);
    if (fused) max_regs = T.uni.n_regs;
    if (part == 2) max_regs = 0;
    L.regs_wave_bytes = static_cast<uint32_t>(((max_regs + 1) * 64 * 2 + 15) & ~15);  // + the dummy column
    if (part == 2) {
        h->lds_mo = L;
        h->lds_image_mo.swap(img.bytes);
        h->has_mo = true;
        return true;
    }
    h->lds = L;
    h->lds_image.swap(img.bytes);
    h->tile_ok = true;
    return true;
}

// Complete the layout for one batch: staging sized for 64 lines of the hinted length.
// wide: the kernel variant that reads UTF-16 code units (two prefetch registers per staged chunk: 13 KB of staging and 8 waves at most)
bool plan_tile_layout(GxLds L, uint32_t line_bytes_hint, GxLds* out, bool wide = false);
bool plan_tile_launch(const gx_handle* h, uint32_t line_bytes_hint, GxLds* out, bool match_only = false, bool wide = false) {
    if (!h->tile_ok) return false;
    return plan_tile_layout(match_only && h->has_mo ? h->lds_mo : h->lds, line_bytes_hint, out, wide);
}
// the hop tier's layout: the same kernel, its own tables
bool plan_hop_launch(const gx_handle* h, uint32_t line_bytes_hint, GxLds* out, bool match_only = false, bool wide = false) {
    if (match_only) return h->hop_mo_ok && plan_tile_layout(h->lds_hop_mo, line_bytes_hint, out, wide);
    return h->hop_ok && plan_tile_layout(h->lds_hop, line_bytes_hint, out, wide);
}
bool plan_tile_layout(GxLds L, uint32_t line_bytes_hint, GxLds* out, bool wide) {
    if (line_bytes_hint == 0) line_bytes_hint = 200;
    if (line_bytes_hint > 2000) line_bytes_hint = 2000;
    L.stage_bytes = (64u * line_bytes_hint + 64u + 15u) & ~15u;  // + slack: the walk reads ahead of the line
    if (L.stage_bytes > 16384u) L.stage_bytes = 16384u;  // the kernel prefetches a tile into <= 64 VGPRs per lane;
                                                          // longer lines go in several rounds or to the per-line kernel
    if (wide && L.stage_bytes > 13u * 1024u) L.stage_bytes = 13u * 1024u;   // (groups of longer lines go in several rounds)
    const uint32_t bitmap_bytes = L.tier == 4 ? 0u : GX_BITMAP_WAVE_BYTES;   // (the hop tier has no chunk bitmap)
    const uint32_t fixed = L.regs_wave_bytes + bitmap_bytes;
    if (L.table_bytes + 4 * (L.stage_bytes + fixed) > LDS_BYTES) return false;
    uint32_t nw = (LDS_BYTES - L.table_bytes) / (L.stage_bytes + fixed);
    if (nw > 12) nw = 12;  // 768 threads: leaves 170 VGPRs per lane for the prefetch registers
    if ((wide || L.stage_bytes > 13u * 1024u) && nw > 8) nw = 8;  // the 16 KB variant prefetches 64 VGPRs, the UTF-16 variant 104: 2 waves per SIMD
#ifdef GX_DEV
    if (getenv("GX_DEV_NWAVES") && static_cast<uint32_t>(atoi(getenv("GX_DEV_NWAVES"))) < nw) nw = static_cast<uint32_t>(atoi(getenv("GX_DEV_NWAVES")));
#endif
    // Whatever LDS is left goes to the staging areas, up to what the kernel variant's prefetch registers hold: a hint
    // that is a few bytes short (mean length, lines of 201 bytes announced as 200) then still stages whole groups.
    {
        const uint32_t kch = (L.stage_bytes + 1023u) / 1024u;
        const uint32_t cap = (wide ? 13u : kch <= 4 ? 4u : kch <= 8 ? 8u : kch <= 13 ? 13u : 16u) * 1024u;
        uint32_t room = ((LDS_BYTES - L.table_bytes - 32u) / nw - fixed) & ~15u;
        if (room > cap) room = cap;
        if (room > L.stage_bytes) L.stage_bytes = room;
    }
    L.nwaves = nw;
    L.regs = L.table_bytes;
    L.bitmap = L.regs + nw * L.regs_wave_bytes;
    L.counter = L.bitmap + nw * bitmap_bytes;
    L.stage = (L.counter + 16u + 15u) & ~15u;
    L.total_bytes = L.stage + nw * L.stage_bytes;
    if (L.total_bytes > LDS_BYTES) { L.stage_bytes -= 16u; L.total_bytes = L.stage + nw * L.stage_bytes; }
    *out = L;
    return true;
}

// Layout for the lane kernel (gx_lanes.hip): per wave the register block and the area its result rows go through.
bool plan_lanes_launch(const gx_handle* h, GxLds* out, bool match_only, bool compact, bool sorted = false, uint64_t n = 0) {
    if (!h->tile_ok) return false;
    GxLds L = match_only && h->has_mo ? h->lds_mo : h->lds;
    if (!match_only && h->T.has_capture && L.u_start == 0xFFFFFFFFu) return false;  // walks the fused automaton
    const uint32_t slots = 2u * static_cast<uint32_t>(h->T.max_groups);
    const uint32_t rows = match_only || !compact ? 0u : 64u * (2u + 2u * slots);  // (dense rows are stored lane by lane)
    L.stage_bytes = L.regs_wave_bytes;                       // (the register block)
    L.regs_wave_bytes = (L.regs_wave_bytes + rows + 16u + 15u) & ~15u;
    // length-sorted tiles (uneven lines, tables in global memory): a chunk's line order, 2 bytes per line, + two 64-entry tables
    L.sort_chunk = 0;
    L.sort_lds = 0;
    uint32_t sort_bytes = 0;
    if (sorted) {
        // (with the tables in LDS too the index array takes the place of a wave or two)
        for (uint32_t ch = L.tier == 2 ? 2048u : 8192u; ch >= 2048u; ch >>= 1)  // (records in LDS: 2 048 lines cost one wave, 8 192 two)
            if (L.table_bytes + 32u + 2u * ch + 512u + (L.tier == 2 ? 14u : 8u) * L.regs_wave_bytes <= LDS_BYTES) { L.sort_chunk = ch; sort_bytes = 2u * ch + 512u; break; }
    }
    // Chunks are what the workgroups share the batch in, and a chunk ends with a barrier, so its waves want several tiles
    // each (measured: 1 M lines of 50-2000 bytes on the LDS records: chunks of 8 192 lines 0.82 ms -- 122 chunks for 256
    // CUs --, 2 048 lines 0.52 ms, 1 024 lines 0.95 ms -- one tile per wave and chunk; configs[4], 2 M lines: 8 192 1.75 ms,
    // 1 984 1.82 ms).  Small batches take smaller chunks, down to 2 048 lines.
    if (L.sort_chunk && n)
        while (L.sort_chunk > 2048u && n / L.sort_chunk < static_cast<uint64_t>(h->num_cus > 0 ? h->num_cus : 256) / 2) L.sort_chunk >>= 1;
    if (L.table_bytes + 32u + sort_bytes + 4u * L.regs_wave_bytes > LDS_BYTES) return false;
    L.nwaves = std::min<uint32_t>(16u, (LDS_BYTES - L.table_bytes - 32u - sort_bytes) / L.regs_wave_bytes);
    L.sort_lds = L.table_bytes;  // (behind the tables)
    L.regs = L.table_bytes + sort_bytes;
    L.bitmap = 0;
    L.counter = L.regs + L.nwaves * L.regs_wave_bytes;
    L.stage = 0;
    L.total_bytes = L.counter + 16u;
    if (L.total_bytes > LDS_BYTES) return false;
    *out = L;
    return true;
}

// Layout for the slice kernel: a 64 x 80-byte slice buffer per wave, up to 16 waves.
bool plan_slice_launch(const gx_handle* h, GxLds* out, bool match_only = false) {
    if (!h->tile_ok) return false;
    if (h->T.has_capture && h->lds.u_start == 0xFFFFFFFFu) return false;  // the slice kernel walks the fused automaton
    GxLds L = match_only && h->has_mo ? h->lds_mo : h->lds;
    L.stage_bytes = 64u * 80u;
    const uint32_t per_wave = L.stage_bytes + L.regs_wave_bytes;
    if (L.table_bytes + per_wave > LDS_BYTES) return false;
    uint32_t nw = (LDS_BYTES - L.table_bytes) / per_wave;
    if (nw > 16) nw = 16;
    L.nwaves = nw;
    L.regs = L.table_bytes;
    L.bitmap = 0;
    L.stage = L.regs + nw * L.regs_wave_bytes;
    L.total_bytes = L.stage + nw * L.stage_bytes;
    *out = L;
    return true;
}

// Layout for the hop slice kernel: the hop tier's tables, per wave a register block and a [64][144]-byte piece buffer.
bool plan_hop_slice_launch(const gx_handle* h, GxLds* out, bool match_only = false) {
    if (!(match_only ? h->hop_mo_ok : h->hop_ok)) return false;
    GxLds L = match_only ? h->lds_hop_mo_small : h->lds_hop_small;
    L.stage_bytes = 64u * (GX_HOP_SLICE_BYTES + 16u) + 48u;  // (+ 48: a window read at a row's last bytes runs a few bytes past it)
    const uint32_t per_wave = L.stage_bytes + L.regs_wave_bytes;
    if (L.table_bytes + 4u * per_wave > LDS_BYTES) return false;
    uint32_t nw = (LDS_BYTES - L.table_bytes) / per_wave;
    if (nw > 12) nw = 12;   // (the kernel holds the loads of eight tested lines across its walk: three waves per SIMD by registers)
    L.nwaves = nw;
    L.regs = L.table_bytes;
    L.bitmap = 0;
    L.stage = L.regs + nw * L.regs_wave_bytes;
    L.total_bytes = L.stage + nw * L.stage_bytes;
    *out = L;
    return true;
}

// kernel choice: automaton rows in LDS when they fit, else sparse range records in LDS, else dense rows in global
// memory (L2), else the per-line kernel alone.  Host work only (also done for host-only handles, where it is a check
// of the builders and feeds gx_stat).
void choose_tile_image(gx_handle* h) {
    // (a definition with an extraction that has no capture automaton: the per-line kernel alone -- it is the one that runs programs)
    const bool no_tiles = (h->create_flags & GX_CREATE_NO_TILES) != 0 || h->T.has_pike(), force_l2 = (h->create_flags & GX_CREATE_TIER_L2) != 0;
    const bool force_rec = (h->create_flags & GX_CREATE_TIER_RECORDS) != 0;
    auto records = [&](int tier) {
        if (build_tile_image(h, tier)) return true;
        // the fused automaton alone, and a second image with the match automaton alone for match-only batches
        return build_tile_image(h, tier, 1) && build_tile_image(h, tier, 2);
    };
    const bool force_recg = (h->create_flags & GX_CREATE_TIER_RECORDS_GLOBAL) != 0;
    bool ok = false;
    if (!no_tiles) {
        if (force_l2) ok = build_tile_image(h, 1);
        else if (force_recg) ok = records(3) || build_tile_image(h, 1);
        else if (force_rec) ok = records(2) || records(3) || build_tile_image(h, 1);
        else {
            // Measured on the 64-extraction definition of BASELINE configs[2] (10 M x 200-byte lines), captures / match only:
            // range records in LDS walked by the lane kernel 1.9 / 1.4 ms; dense rows in global memory (L2) under the tile
            // kernel 3.0 / 3.0 ms; records in LDS under the tile kernel 3.9 / 1.9 ms; records in global memory 4.0 ms.
            // Hence: dense rows in LDS when they fit; else records in LDS when they fit beside at least 8 waves of the
            // lane kernel; else dense rows in global memory for the capture side and, when they fit, LDS records for
            // match-only batches.
            ok = build_tile_image(h, 0);
            if (!ok) {
                GxLds L;
                ok = records(2) && plan_lanes_launch(h, &L, false, true) && L.nwaves >= 8u;
                if (!ok) {
                    ok = build_tile_image(h, 1);
                    if (ok) (void)build_tile_image(h, 2, 2);
                }
            }
        }
    }
    if (!ok) h->tile_ok = false;
    // The hop tier beside it, for capture batches: whenever the dense rows do not fit LDS (or on request).  Its hot records
    // may take what LDS leaves beside eight waves' staging areas of 200-byte lines.
    h->hop_ok = false;
    const bool forced = force_l2 || force_rec || force_recg;  // (a caller that names a tier gets that tier's kernels)
    const bool want_hop = (h->create_flags & GX_CREATE_TIER_HOP) != 0 || (ok && !forced && (h->lds.tier != 0 || h->tile_global));
    uint32_t hot_budget = 48u * 1024u;
#ifdef GX_DEV
    if (getenv("GX_DEV_HOT_BUDGET")) hot_budget = static_cast<uint32_t>(atoi(getenv("GX_DEV_HOT_BUDGET")));
#endif
    uint32_t small_budget = 12u * 1024u;   // the hop slice kernel's share of LDS for hot records
#ifdef GX_DEV
    if (getenv("GX_DEV_SMALL_BUDGET")) small_budget = static_cast<uint32_t>(atoi(getenv("GX_DEV_SMALL_BUDGET")));
#endif
    // layouts of one hop image: the tile kernel's (full) and the hop slice kernel's (small: fewer hot records, more waves)
    auto layouts = [&](const HopImage& I, GxLds* full, GxLds* small) {
        GxLds L{};
        L.ncls = I.ncls;
        L.row_bytes = I.row_bytes;
        L.c_base = I.hops_off;
        L.m_start = I.match_automaton ? I.start : 0u;
        L.m_dead = I.match_automaton ? I.dead : 0u;
        L.u_start = I.match_automaton ? 0xFFFFFFFFu : I.start;
        L.u_dead = I.match_automaton ? 0xFFFFFFFFu : I.dead;
        L.fin_tags = I.fin_off;
        L.fin_state_off = I.fin_state_off;
        L.fin_state_rec = I.fin_state_rec;
        L.simple_ops = 1;
        L.tier = 4;
        L.rec = HOP_AT;
        L.sort_chunk = I.n_reachable_hot;  // (hop tier: the states well-formed lines reach; rec_indexed of them are in LDS)
        L.hot_lo4 = 0;
        L.hot_k4 = 0x80808080u;
        L.regs_wave_bytes = static_cast<uint32_t>(((I.n_regs + 1) * 64 * 2 + 15) & ~15u);   // (the dummy column, then the registers)
        L.fin_unset = I.col_unset;
        for (int q = 0; q < 2; ++q) {
            const HopLds& P = q ? I.small : I.full;
            L.table_bytes = static_cast<uint32_t>(P.bytes.size());
            L.rec_indexed = P.n_hot;
            L.acc_tab = P.info_lds;   // int16 info words of the hot states
            L.at = P.fin_lds;         // final records in LDS (0: in the global image at fin_tags)
            L.hop_sets = P.sets_lds;  // the loop sets (the walk's second chance)
            *(q ? small : full) = L;
        }
    };
    h->hop_reason = 4;   // not built: the dense rows fit LDS (or the caller named another tier)
    if (!no_tiles && want_hop && h->T.has_capture && !(h->create_flags & GX_CREATE_NO_FUSED)) {
        if (build_hop_image(h->T, false, hot_budget, small_budget, h->hop)) {
            layouts(h->hop, &h->lds_hop, &h->lds_hop_small);
            GxLds P;
            h->hop_ok = plan_tile_layout(h->lds_hop, 200, &P);
            h->hop_reason = h->hop_ok ? 0 : 5;   // (5: the tables leave no room in LDS for a wave)
        } else h->hop_reason = h->hop.refused;
    } else if (want_hop && !no_tiles) h->hop_reason = 1;
    // ... and of the match automaton alone, for match-only batches (PolyMatcher.match over a batch)
    h->hop_mo_ok = false;
    if (!no_tiles && want_hop && build_hop_image(h->T, true, hot_budget, small_budget, h->hop_mo)) {
        layouts(h->hop_mo, &h->lds_hop_mo, &h->lds_hop_mo_small);
        GxLds P;
        h->hop_mo_ok = plan_tile_layout(h->lds_hop_mo, 200, &P);
    }
}

// One table image into the handle's device: from host memory, or -- gx_create_on_devices -- from the same image on the device of
// the handle that was built first: a copy between devices (over xGMI where the devices are peers; the runtime stages it otherwise).
static void put_image(gx_handle* h, void* dst, const void* host, size_t bytes, const void* peer) {
    if (bytes == 0) return;
    if (peer && g_peer_src && hipMemcpyPeer(dst, h->device, peer, g_peer_src->device, bytes) == hipSuccess) {
        h->peer_image_bytes += bytes;
        return;
    }
    GX_HIP(hipMemcpy(dst, host, bytes, hipMemcpyHostToDevice));
}

void upload(gx_handle* h) {
    const Tables& T = h->T;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw GxError(GX_E_DEVICE, "no HIP device available (libgorp_hip needs a gfx950 GPU; there is no CPU fallback)");
    GX_HIP(hipGetDevice(&h->device));

    Image img;
    const size_t o_cls = img.put(T.cls256, 256);
    const size_t o_hilo = img.put(T.hi_lo);
    const size_t o_hicls = img.put(T.hi_cls);
    size_t o_next16 = 0, o_next32 = 0;
    const bool small = T.m_states <= 65536;
    if (small) {
        std::vector<uint16_t> n16(T.m_next.begin(), T.m_next.end());
        o_next16 = img.put(n16);
    } else o_next32 = img.put(T.m_next);
    const size_t o_acc = img.put(T.m_accept_first);
    std::vector<uint32_t> trans_all, trans_off, fin_off;
    std::vector<int32_t> fin_all, ngroups;
    int max_regs = 0;
    for (auto& r : T.rules) {
        trans_off.push_back(static_cast<uint32_t>(trans_all.size()));
        trans_all.insert(trans_all.end(), r.trans.begin(), r.trans.end());
        fin_off.push_back(static_cast<uint32_t>(fin_all.size()));
        fin_all.insert(fin_all.end(), r.fin.begin(), r.fin.end());
        ngroups.push_back(r.n_groups);
        max_regs = std::max(max_regs, r.n_regs);
    }
    if (trans_all.empty()) { trans_all.push_back(0); trans_off.push_back(0); fin_all.push_back(-1); fin_off.push_back(0); ngroups.push_back(0); }
    h->max_regs = max_regs;
    if (max_regs > 96) throw GxError(GX_E_LIMIT, "capture automaton needs more than 96 registers");
    const size_t o_trans = img.put(trans_all);
    const size_t o_troff = img.put(trans_off);
    const size_t o_fin = img.put(fin_all);
    const size_t o_finoff = img.put(fin_off);
    const size_t o_ng = img.put(ngroups);
    const size_t o_opsoff = img.put(T.ops_off);
    std::vector<uint16_t> ops = T.ops;
    if (ops.empty()) ops.push_back(0);
    const size_t o_ops = img.put(ops);
    std::vector<uint16_t> fin_tags = T.fin_tags;
    if (fin_tags.empty()) fin_tags.push_back(0);
    const size_t o_fintags = img.put(fin_tags);
    // extractions without a capture automaton: their programs (gx_compile.hpp: Tables::pike_*)
    size_t o_pike_off = 0, o_pike_code = 0, o_pike_sets = 0;
    uint32_t pike_lane_ints = 0, pike_blocks = 0;
    if (T.has_pike()) {
        o_pike_off = img.put(T.pike_off);
        o_pike_code = img.put(T.pike_code);
        o_pike_sets = img.put(T.pike_sets);
        for (size_t k = 0; k < T.rules.size(); ++k) {
            const uint32_t ni = T.pike_off[k + 1] - T.pike_off[k], W = 1u + 2u * static_cast<uint32_t>(T.rules[k].n_groups);
            if (ni) pike_lane_ints = std::max(pike_lane_ints, 2u * ni * W + 3u * (2u * ni + 2u) + ni + 2u * static_cast<uint32_t>(T.rules[k].n_groups));
        }
        pike_lane_ints += 2u * static_cast<uint32_t>(T.max_groups) + 2u;
        // as many workgroups of 256 lanes as the scratch budget holds thread lists for (64 at most, one at least)
        pike_blocks = static_cast<uint32_t>(std::min<uint64_t>(GX_PIKE_BLOCKS, GX_PIKE_SCRATCH_BYTES / (static_cast<uint64_t>(pike_lane_ints) * 4u * 256u)));
        if (pike_blocks == 0) pike_blocks = 1;
        if (static_cast<uint64_t>(pike_lane_ints) * 4u * (256u * pike_blocks + 1u) > (1ull << 30))
            throw GxError(GX_E_LIMIT, "capture program too large to run as it is (the thread lists of one workgroup beyond 1 GiB)");
    }

    GX_HIP(hipMalloc(&h->dimage, img.bytes.size()));
    h->image_bytes = img.bytes.size();
    put_image(h, h->dimage, img.bytes.data(), img.bytes.size(), g_peer_src ? g_peer_src->dimage : nullptr);
    const uint8_t* base = static_cast<const uint8_t*>(h->dimage);
    GxDev& d = h->dev;
    d.cls256 = base + o_cls;
    d.hi_lo = reinterpret_cast<const uint16_t*>(base + o_hilo);
    d.hi_cls = reinterpret_cast<const uint16_t*>(base + o_hicls);
    d.n_hi = static_cast<int32_t>(T.hi_lo.size());
    d.ncls = T.ncls;
    d.m_next16 = small ? reinterpret_cast<const uint16_t*>(base + o_next16) : nullptr;
    d.m_next32 = small ? nullptr : reinterpret_cast<const uint32_t*>(base + o_next32);
    d.m_accept_first = reinterpret_cast<const int32_t*>(base + o_acc);
    d.m_states = T.m_states;
    d.m_dead = T.m_dead;
    d.c_trans = reinterpret_cast<const uint32_t*>(base + o_trans);
    d.c_trans_off = reinterpret_cast<const uint32_t*>(base + o_troff);
    d.c_fin = reinterpret_cast<const int32_t*>(base + o_fin);
    d.c_fin_off = reinterpret_cast<const uint32_t*>(base + o_finoff);
    d.c_ngroups = reinterpret_cast<const int32_t*>(base + o_ng);
    d.ops_off = reinterpret_cast<const uint32_t*>(base + o_opsoff);
    d.ops = reinterpret_cast<const uint16_t*>(base + o_ops);
    d.fin_tags = reinterpret_cast<const uint16_t*>(base + o_fintags);
    d.n_rules = T.n_rules;
    d.max_groups = T.max_groups;
    d.max_regs = max_regs;
    d.has_capture = T.has_capture ? 1 : 0;
    if (T.has_pike()) {
        d.pike_off = reinterpret_cast<const uint32_t*>(base + o_pike_off);
        d.pike_code = reinterpret_cast<const uint32_t*>(base + o_pike_code);
        d.pike_sets = reinterpret_cast<const uint32_t*>(base + o_pike_sets);
        d.pike_lane_ints = pike_lane_ints;
        d.pike_blocks = pike_blocks;
        GX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_pike_scratch), static_cast<size_t>(pike_lane_ints) * 4u * (256u * pike_blocks + 1u)));
        d.pike_scratch = h->d_pike_scratch;
    }

    choose_tile_image(h);
    if (h->tile_ok) {
        GX_HIP(hipMalloc(&h->d_lds_image, h->lds_image.size()));
        put_image(h, h->d_lds_image, h->lds_image.data(), h->lds_image.size(), g_peer_src ? g_peer_src->d_lds_image : nullptr);
        if (h->has_mo) {
            GX_HIP(hipMalloc(&h->d_lds_image_mo, h->lds_image_mo.size()));
            put_image(h, h->d_lds_image_mo, h->lds_image_mo.data(), h->lds_image_mo.size(), g_peer_src ? g_peer_src->d_lds_image_mo : nullptr);
        }
        if (h->tile_global) {
            GX_HIP(hipMalloc(&h->d_l2_image, h->l2_image.size()));
            put_image(h, h->d_l2_image, h->l2_image.data(), h->l2_image.size(), g_peer_src ? g_peer_src->d_l2_image : nullptr);
        }
    }
    if (h->tile_ok || h->hop_ok || h->hop_mo_ok) {
        if (h->hop_ok) {
            GX_HIP(hipMalloc(&h->d_lds_image_hop, h->hop.full.bytes.size()));
            put_image(h, h->d_lds_image_hop, h->hop.full.bytes.data(), h->hop.full.bytes.size(), g_peer_src ? g_peer_src->d_lds_image_hop : nullptr);
            GX_HIP(hipMalloc(&h->d_lds_image_hop_small, h->hop.small.bytes.size()));
            put_image(h, h->d_lds_image_hop_small, h->hop.small.bytes.data(), h->hop.small.bytes.size(), g_peer_src ? g_peer_src->d_lds_image_hop_small : nullptr);
            GX_HIP(hipMalloc(&h->d_hop_global, h->hop.global.size()));
            put_image(h, h->d_hop_global, h->hop.global.data(), h->hop.global.size(), g_peer_src ? g_peer_src->d_hop_global : nullptr);
        }
        if (h->hop_mo_ok) {
            GX_HIP(hipMalloc(&h->d_lds_image_hop_mo, h->hop_mo.full.bytes.size()));
            put_image(h, h->d_lds_image_hop_mo, h->hop_mo.full.bytes.data(), h->hop_mo.full.bytes.size(), g_peer_src ? g_peer_src->d_lds_image_hop_mo : nullptr);
            GX_HIP(hipMalloc(&h->d_lds_image_hop_mo_small, h->hop_mo.small.bytes.size()));
            put_image(h, h->d_lds_image_hop_mo_small, h->hop_mo.small.bytes.data(), h->hop_mo.small.bytes.size(), g_peer_src ? g_peer_src->d_lds_image_hop_mo_small : nullptr);
            GX_HIP(hipMalloc(&h->d_hop_mo_global, h->hop_mo.global.size()));
            put_image(h, h->d_hop_mo_global, h->hop_mo.global.data(), h->hop_mo.global.size(), g_peer_src ? g_peer_src->d_hop_mo_global : nullptr);
        }
        GX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_slots), 3 * gx_handle::N_SLOTS * sizeof(uint32_t)));
        GX_HIP(hipMemset(h->d_slots, 0, 3 * gx_handle::N_SLOTS * sizeof(uint32_t)));
        GX_HIP(hipEventCreateWithFlags(&h->shared_event, hipEventDisableTiming));

        GX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_broken), gx_handle::N_SLOTS * sizeof(uint32_t), hipHostMallocMapped));
        for (int q = 0; q < gx_handle::N_SLOTS; ++q) h->h_broken[q] = 0;
        GX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_broken), h->h_broken, 0));
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->num_cus = cus;
        // the resident one-line service: dense rows in LDS, and either no captures or the fused automaton with simple programs
        if ((h->create_flags & GX_CREATE_RESIDENT_ONE) && h->tile_ok && !h->tile_global && h->lds.tier == 0 && h->T.max_groups <= 32 &&
            (!h->T.has_capture || (h->lds.u_start != 0xFFFFFFFFu && h->lds.simple_ops))) {
            gx_handle::Service& sv = h->svc;
            sv.mode = h->T.has_capture ? 1 : 0;
            GxLds L = h->lds;
            L.nwaves = 1;
            L.stage_bytes = ((GX_SERVICE_MAX_BYTES + 8u + 64u + 15u) & ~15u) + 272u;   // (+ the answer's words: gx_service.hip)
            L.regs = L.table_bytes;
            L.bitmap = L.regs + L.regs_wave_bytes;
            L.counter = L.bitmap + GX_BITMAP_WAVE_BYTES;
            L.stage = (L.counter + 16u + 15u) & ~15u;
            L.total_bytes = L.stage + L.stage_bytes;
            if (L.total_bytes <= LDS_BYTES) {
                sv.L = L;
                const size_t dwords = 17 * 16 + 80 + 16;
                GX_HIP(hipHostMalloc(reinterpret_cast<void**>(&sv.host), dwords * 4, hipHostMallocMapped));
                memset(sv.host, 0, dwords * 4);
                GX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&sv.dev), sv.host, 0));
                GX_HIP(hipStreamCreateWithFlags(&sv.stream, hipStreamNonBlocking));
                sv.enabled = true;
            }
        }
    }
    h->on_device = true;
}

// One batch on the device: tile kernel (LDS tier when the tables fit LDS, else L2 tier), slice kernel for long lines,
// per-line kernel otherwise.  kernel: gx_batch_opts.kernel (0 = choose).
// What a launch leaves for its caller: where a broken max_line_bytes promise would show, and how to make good for it.
struct Launched {
    int slot = -1;
    uint32_t seq = 0;
    bool promised = false;     // no follow-up launch: the caller said that no line is longer than the kernel takes
    uint32_t limit = 0;        // launch_extract_oversize's arguments for this launch
    int by_length = 0;
};

// The launch's flag word (and the lane kernel's chunk counter): the slot of its stream.  Call under h->slot_mu.
struct SlotUse { int slot; bool shared; };
SlotUse take_slot(gx_handle* h, GxBatch& b, hipStream_t stream) {
    b.seq = h->next_seq++;
    if (h->next_seq == 0) h->next_seq = 1;
    int slot = -1;
    for (int q = 0; q < gx_handle::N_SLOTS - 1 && slot < 0; ++q)
        if (h->slot_taken[q] && h->slot_stream[q] == stream) slot = q;
    for (int q = 0; q < gx_handle::N_SLOTS - 1 && slot < 0; ++q)
        if (!h->slot_taken[q]) { h->slot_taken[q] = true; h->slot_stream[q] = stream; slot = q; }
    const bool shared = slot < 0;
    if (shared) {
        slot = gx_handle::N_SLOTS - 1;
        if (h->shared_used) GX_HIP(hipStreamWaitEvent(stream, h->shared_event, 0));
        h->shared_used = true;
    }
    // a batch of this stream that promised its longest line, ran without a follow-up launch and met a longer line after all
    // (no_sync batches: nobody has looked yet)
    // (the word holds the sequence number of the LAST launch that broke its promise; until round 4 it was compared with the newest
    // promise alone, and a kernel that reached its long line after the next batch had been enqueued was never noticed)
    const uint32_t word = __atomic_load_n(&h->h_broken[slot], __ATOMIC_RELAXED);
    if (word != h->broken_seen[slot]) {
        h->broken_seen[slot] = word;
        h->promise_seq[slot] = 0;
        h->promises_broken.fetch_add(1);
        throw GxError(GX_E_ARG, "an earlier no_sync batch on this stream held a line longer than its gx_batch_opts.max_line_bytes: that line was not "
                                "processed (its result row is unwritten); this batch was not launched");
    }
    b.oversize_flag = h->d_slots + slot;
    if (!h->d_steal[slot]) {   // (the slot's first launch: 768 KB, zeroed once -- every launch leaves the next one's row zeroed)
        const size_t bytes = 2 * static_cast<size_t>(GX_STEAL_MAX) * GX_STEAL_STRIDE * sizeof(uint32_t);
        GX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_steal[slot]), bytes));
        GX_HIP(hipMemsetAsync(h->d_steal[slot], 0, bytes, stream));   // (on the launch's own stream: in order before its kernel; the shared slot's later users wait for its event)
    }
    b.steal = h->d_steal[slot];
    b.steal_parity = h->steal_parity[slot];
    return SlotUse{slot, shared};
}
// The follow-up launch for the lines the batch kernel leaves (longer than `limit`: see launch_extract_oversize) -- unless the
// caller promised that there are none (b.max_line_bytes within `fits`), or the host knows (b.no_followup: the one-line calls).
// Call before the batch kernel is launched; returns what to launch after it.
bool plan_followup(gx_handle* h, GxBatch& b, const SlotUse& u, uint32_t fits, Launched* out) {
    if (out) { out->slot = u.slot; out->seq = b.seq; }
    if (b.no_followup) return false;
    if (b.max_line_bytes != 0 && b.max_line_bytes <= fits) {
        b.no_followup = 1;
        b.oversize_flag = h->d_broken + u.slot;
        h->promise_seq[u.slot] = b.seq;
        if (out) out->promised = true;
        return false;
    }
    h->promise_seq[u.slot] = 0;
    return true;
}
// Has a batch of this stream broken its promise since anybody looked?  (For the calls that wait for their batches themselves:
// the word is final once the stream is idle.)  Marks it seen.
bool promise_broken_since(gx_handle* h, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(h->slot_mu);
    int slot = gx_handle::N_SLOTS - 1;
    for (int q = 0; q < gx_handle::N_SLOTS - 1; ++q)
        if (h->slot_taken[q] && h->slot_stream[q] == stream) { slot = q; break; }
    if (!h->h_broken) return false;
    const uint32_t word = __atomic_load_n(&h->h_broken[slot], __ATOMIC_RELAXED);
    if (word == h->broken_seen[slot]) return false;
    h->broken_seen[slot] = word;
    h->promise_seq[slot] = 0;
    h->promises_broken.fetch_add(1);
    return true;
}
void done_slot(gx_handle* h, const SlotUse& u, hipStream_t stream) {
    if (u.shared) GX_HIP(hipEventRecord(h->shared_event, stream));
}

// Stream-ordered memory out of the handle's own pool (gx_handle::pool) for the length of a scope: freed on the stream when the scope
// ends -- by a return or by an exception (take_slot and plan_followup throw).
struct PoolBuffer {
    void* p = nullptr;
    hipStream_t s;
    PoolBuffer(gx_handle* h, size_t bytes, hipStream_t stream) : s(stream) {
        {
            std::lock_guard<std::mutex> pool_lock(h->slot_mu);   // (device-pointer batches of several threads come here without h->mu)
            if (!h->pool) {
                hipMemPoolProps props{};
                props.allocType = hipMemAllocationTypePinned;
                props.handleTypes = hipMemHandleTypeNone;
                props.location.type = hipMemLocationTypeDevice;
                props.location.id = h->device;
                GX_HIP(hipMemPoolCreate(&h->pool, &props));
                uint64_t keep = ~0ull;
                GX_HIP(hipMemPoolSetAttribute(h->pool, hipMemPoolAttrReleaseThreshold, &keep));
            }
        }
        GX_HIP(hipMallocFromPoolAsync(&p, bytes, h->pool, stream));
    }
    ~PoolBuffer() { if (p) (void)hipFreeAsync(p, s); }
    PoolBuffer(const PoolBuffer&) = delete;
    PoolBuffer& operator=(const PoolBuffer&) = delete;
};

// Launches of a handle that runs an extraction's program as it is take their turns across streams: the thread lists are the handle's.
struct PikeGate {
    gx_handle* h;
    hipStream_t s;
    bool on;
    PikeGate(gx_handle* h_, hipStream_t s_) : h(h_), s(s_), on(h_->T.has_pike()) {
        if (!on) return;
        h->pike_mu.lock();
        if (h->pike_depth++ == 0 && h->pike_event_set) (void)hipStreamWaitEvent(s, h->pike_event, 0);
    }
    ~PikeGate() {
        if (!on) return;
        if (--h->pike_depth == 0) {
            if (!h->pike_event) (void)hipEventCreateWithFlags(&h->pike_event, hipEventDisableTiming);
            if (h->pike_event && hipEventRecord(h->pike_event, s) == hipSuccess) h->pike_event_set = true;
        }
        h->pike_mu.unlock();
    }
    PikeGate(const PikeGate&) = delete;
    PikeGate& operator=(const PikeGate&) = delete;
};

void launch_batch(gx_handle* h, GxBatch b, uint32_t line_bytes_hint, uint32_t kernel, hipStream_t stream, bool uneven = false, Launched* launched = nullptr) {
    PikeGate pike_gate(h, stream);
    GxLds L;
    if (b.wide && !b.state_out && b.match_only >= 0 && b.n > 0 && line_bytes_hint <= 255u && !uneven &&
        (kernel == GX_KERNEL_AUTO || kernel == GX_KERNEL_TILES || kernel == GX_KERNEL_HOPS)) {
        // UTF-16 code units, lines of ordinary length, tables that are dense rows in LDS or hop tables: the tile kernel reads the
        // units itself -- their low bytes are what it stages -- and flags the lines that hold a unit above 0xFF for the per-line
        // walk (k_extract_flagged, which leaves at once when there is none).  No copy of the batch, no synchronisation.
        const bool mo = b.match_only != 0 || !h->T.has_capture;
        const bool have_hop = mo ? h->hop_mo_ok : h->hop_ok;
        const bool hop = have_hop && kernel != GX_KERNEL_TILES;
        const uint32_t image_tier = mo && h->has_mo ? h->lds_mo.tier : h->lds.tier;
        bool direct = hop ? plan_hop_launch(h, line_bytes_hint, &L, mo, true)
                          : (kernel != GX_KERNEL_HOPS && h->tile_ok && !h->tile_global && image_tier == 0 && plan_tile_launch(h, line_bytes_hint, &L, mo, true));
        if (direct && hop && !mo && !(h->lds_hop.u_start != 0xFFFFFFFFu)) direct = false;
        if (direct) {
            PoolBuffer flags_buf(h, b.n + 64, stream);   // (given back to the pool when this scope ends, whatever ends it)
            void* flags = flags_buf.p;
            hipError_t e = hipSuccess;
            {
                std::lock_guard<std::mutex> lock(h->slot_mu);
                const SlotUse u = take_slot(h, b, stream);
                const bool followup = plan_followup(h, b, u, L.stage_bytes >= 63u ? L.stage_bytes - 63u : 0u, launched);
                if (launched) { launched->limit = L.stage_bytes; launched->by_length = 0; }
                b.wide_flags = static_cast<uint8_t*>(flags);
                b.wide_any = h->d_slots + 2 * gx_handle::N_SLOTS + u.slot;
                const uint8_t* image = static_cast<const uint8_t*>(hop ? (mo ? h->d_lds_image_hop_mo : h->d_lds_image_hop) : (mo && h->has_mo ? h->d_lds_image_mo : h->d_lds_image));
                const uint8_t* at_global = hop ? static_cast<const uint8_t*>(mo ? h->d_hop_mo_global : h->d_hop_global) : nullptr;
                h->last_kernel = hop ? GX_KERNEL_HOPS : GX_KERNEL_TILES;
                e = launch_extract_tile(h->dev, L, image, at_global, h->num_cus, b, stream, nullptr);
                if (e == hipSuccess) {
                    h->steal_parity[u.slot] ^= 1u;
                    e = launch_extract_flagged(h->dev, b, static_cast<const uint8_t*>(flags), stream, b.wide_any);
                }
                if (e == hipSuccess && followup) e = launch_extract_oversize(h->dev, b, L.stage_bytes, 0, stream);
                if (e == hipSuccess) done_slot(h, u, stream);
            }
            GX_HIP(e);
            return;
        }
    }
    if (b.wide && !b.state_out && b.match_only >= 0 && b.n > 0 && (line_bytes_hint > 255u || uneven || kernel == GX_KERNEL_HOP_SLICES) &&
        (kernel == GX_KERNEL_AUTO || kernel == GX_KERNEL_HOP_SLICES)) {
        // UTF-16 code units, long or uneven lines, hop tables: the hop slice kernel reads the units itself (a loading lane fetches 16
        // units and stages their low bytes) and flags the lines that hold a unit above 0xFF, as the tile kernel above.
        const bool mo = b.match_only != 0 || !h->T.has_capture;
        if ((mo ? h->hop_mo_ok : h->hop_ok) && plan_hop_slice_launch(h, &L, mo)) {
            PoolBuffer flags_buf(h, b.n + 64, stream);   // (given back to the pool when this scope ends, whatever ends it)
            void* flags = flags_buf.p;
            hipError_t e = hipSuccess;
            {
                std::lock_guard<std::mutex> lock(h->slot_mu);
                const SlotUse u = take_slot(h, b, stream);
                const bool followup = plan_followup(h, b, u, 65535u, launched);
                if (launched) { launched->limit = 65535u; launched->by_length = 1; }
                b.wide_flags = static_cast<uint8_t*>(flags);
                b.wide_any = h->d_slots + 2 * gx_handle::N_SLOTS + u.slot;
                b.chunk_ctr = h->d_slots + gx_handle::N_SLOTS + u.slot;
                b.chunk_base = h->chunk_tickets[u.slot];
                h->last_kernel = GX_KERNEL_HOP_SLICES;
                e = launch_extract_hop_slices(h->dev, L, static_cast<const uint8_t*>(mo ? h->d_lds_image_hop_mo_small : h->d_lds_image_hop_small),
                                              static_cast<const uint8_t*>(mo ? h->d_hop_mo_global : h->d_hop_global), h->num_cus, b, stream);
                if (e == hipSuccess) {
                    h->chunk_tickets[u.slot] += hop_slices_tickets(b.n, L.nwaves, h->num_cus);
                    e = launch_extract_flagged(h->dev, b, static_cast<const uint8_t*>(flags), stream, b.wide_any);
                }
                if (e == hipSuccess && followup) e = launch_extract_oversize(h->dev, b, 65535u, 1, stream);
                if (e == hipSuccess) done_slot(h, u, stream);
            }
            GX_HIP(e);
            return;
        }
    }
    if (b.wide && !b.state_out && b.match_only >= 0 && kernel != GX_KERNEL_PER_LINE && b.n > 0) {
        // UTF-16 code units: their low bytes through the byte kernels, then the lines that hold a unit above 0xFF again through
        // the per-line walk (gx_kernels.hip: k_narrow_units).  The copy is n units long -- the one thing this path has to
        // know on the host, so it reads the two ends of the offsets (a small synchronous copy) -- and lives in
        // stream-ordered memory for the length of the call.
        if (b.caller_no_sync)
            throw GxError(GX_E_ARG, "gx_batch_opts.utf16 with no_sync: this batch would take the narrowed copy of its code units (tables other than dense rows "
                                    "in LDS or hop tables, or a kernel named in gx_batch_opts.kernel), which is sized by a read of the offsets on the host; "
                                    "call it without no_sync");
        const size_t off_w = b.offsets64 ? 8 : 4;
        uint64_t first = 0, last = 0;
        GX_HIP(hipMemcpyAsync(&first, b.offsets, off_w, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipMemcpyAsync(&last, static_cast<const uint8_t*>(b.offsets) + b.n * off_w, off_w, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        const uint64_t units = last >= first ? last - first : 0;
        PoolBuffer tmp_buf(h, units + b.n + 64, stream);
        uint8_t* bytes = static_cast<uint8_t*>(tmp_buf.p);
        uint8_t* flags = bytes + ((units + 15) & ~15ull);
        hipError_t e = launch_narrow_units(b, bytes, flags, stream);
        if (e == hipSuccess) {
            GxBatch nb = b;
            nb.wide = 0;
            nb.max_line_bytes = 0;     // (the copy does not outlive this call: its follow-up launch always runs)
            nb.data = bytes - first;   // (addressed like the units: line i at data + offsets[i])
            launch_batch(h, nb, line_bytes_hint, kernel, stream, uneven, launched);
            e = launch_extract_flagged(h->dev, b, flags, stream);
        }
        GX_HIP(e);
        return;
    }
    // (gx_match_batch wants the product-DFA state a line ends in: the tile kernel on dense rows -- a row is a state -- gives it for
    // match-only batches; every other kernel and table keeps only the first accepting extraction)
    const bool want_states = b.state_out != nullptr;
    const bool batchable = !b.wide && (!want_states || b.match_only == 1) && b.match_only >= 0 && kernel != GX_KERNEL_PER_LINE;
    // Which kernel (gx_batch_opts.kernel 0), by the tables and the mean line length.  Measured, one device (ms; captures /
    // match only):
    //   README definition (dense rows in LDS), 2 M lines of 50-2000 bytes: tiles 0.39, slices 0.73, lanes on sorted tiles 0.75;
    //     400 k lines of 50-20000 bytes (mean 3.4 KB): tiles 61.7 (lines beyond the staging area go one by one), slices 1.74, lanes 3.4
    //   512 extractions, 2 M lines of 50-2000 bytes (configs[4]), dense rows in L2: tiles 4.4 / 4.0, slices 2.39 / 2.20,
    //     lanes 2.19 / 2.07, lanes on length-sorted tiles 1.77 / 1.56 (on range records in global memory 2.18 / 1.81)
    //   64 extractions, 1 M such lines, records in LDS: slices 1.04 / 0.74, lanes 1.41 / 1.22, lanes on sorted tiles 0.52 / 0.47;
    //     200 k lines of 50-20000 bytes: slices 5.1, lanes 9.4
    //   64 extractions, 10 M lines of 200 bytes (configs[2]): records in LDS + lanes 1.52 / 1.18, dense rows in L2 + tiles 3.0 / 3.0
    // Hence: a mean above 1 KB -> slice kernel (64 bytes of every line at a time, a lane takes its next line as soon as it is
    // done); dense rows in LDS -> tile kernel; records in LDS -> lane kernel; anything else -> tile kernel, or above 255 bytes
    // the lane kernel; the lane kernel on tiles of lines of similar length above 255 bytes (gx_lanes.hip, SORTED).
    const bool long_lines = line_bytes_hint > 255u, very_long = line_bytes_hint > 1024u;
    const bool sorted = long_lines || uneven;  // tiles of lines of similar length (the lane kernel's SORTED mode)
    const bool mo = b.match_only != 0 || !h->T.has_capture;
    const uint8_t* image = static_cast<const uint8_t*>(mo && h->has_mo ? h->d_lds_image_mo : h->d_lds_image);
    const uint32_t image_tier = mo && h->has_mo ? h->lds_mo.tier : h->lds.tier;
    const uint8_t* at_global = image_tier == 1 || image_tier == 3 ? static_cast<const uint8_t*>(h->d_l2_image) : nullptr;
    // hop tier: capture batches of definitions whose dense rows do not fit LDS, lines of ordinary length and evenness (the
    // tile kernel wants a tile's lines to be neighbours in memory and about as long as each other)
    // ... and for long or uneven lines the hop slice kernel: a piece of every lane's own line at a time, lanes refilled
    const bool have_hop = (mo ? h->hop_mo_ok : h->hop_ok) && !want_states;
    const uint8_t* hop_image = static_cast<const uint8_t*>(mo ? h->d_lds_image_hop_mo : h->d_lds_image_hop);
    const uint8_t* hop_image_small = static_cast<const uint8_t*>(mo ? h->d_lds_image_hop_mo_small : h->d_lds_image_hop_small);
    const uint8_t* hop_global = static_cast<const uint8_t*>(mo ? h->d_hop_mo_global : h->d_hop_global);
    const bool hop_slices = have_hop && !b.wide && (kernel == GX_KERNEL_HOP_SLICES || (kernel == GX_KERNEL_AUTO && (long_lines || uneven)));
    if (batchable && hop_slices && plan_hop_slice_launch(h, &L, mo)) {
        std::lock_guard<std::mutex> lock(h->slot_mu);
        const SlotUse u = take_slot(h, b, stream);
        const bool followup = plan_followup(h, b, u, 65535u, launched);
        if (launched) { launched->limit = 65535u; launched->by_length = 1; }
        h->last_kernel = GX_KERNEL_HOP_SLICES;
        unsigned long long* stamps = nullptr;
#ifdef GX_DEV
        stamps = h->dev_stamps;
#endif
        // (the pool of chunks the launch's waves share at its end: the slot's chunk counter, as the lane kernel's sorted tiles)
        b.chunk_ctr = h->d_slots + gx_handle::N_SLOTS + u.slot;
        b.chunk_base = h->chunk_tickets[u.slot];
        GX_HIP(launch_extract_hop_slices(h->dev, L, hop_image_small, hop_global, h->num_cus, b, stream, stamps));
        h->chunk_tickets[u.slot] += hop_slices_tickets(b.n, L.nwaves, h->num_cus);
        if (followup) GX_HIP(launch_extract_oversize(h->dev, b, 65535u, 1, stream));   // (lines beyond the 16-bit positions, if the kernel met any)
        done_slot(h, u, stream);
        return;
    }
    const bool hops = have_hop && !b.wide && (kernel == GX_KERNEL_HOPS || (kernel == GX_KERNEL_AUTO && !long_lines && !uneven));
    if (batchable && hops && plan_hop_launch(h, line_bytes_hint, &L, mo)) {
        std::lock_guard<std::mutex> lock(h->slot_mu);
        const SlotUse u = take_slot(h, b, stream);
        // (a line fits a wave's staging area when its bytes + the 15 its address may add + the walk's look-ahead do)
        const bool followup = plan_followup(h, b, u, L.stage_bytes >= 63u ? L.stage_bytes - 63u : 0u, launched);
        if (launched) { launched->limit = L.stage_bytes; launched->by_length = 0; }
        unsigned long long* stamps = nullptr;
#ifdef GX_DEV
        stamps = h->dev_stamps;
#endif
        h->last_kernel = GX_KERNEL_HOPS;
        GX_HIP(launch_extract_tile(h->dev, L, hop_image, hop_global, h->num_cus, b, stream, stamps));
        h->steal_parity[u.slot] ^= 1u;
        if (followup) GX_HIP(launch_extract_oversize(h->dev, b, L.stage_bytes, 0, stream));
        done_slot(h, u, stream);
        return;
    }
    const bool slices = (kernel == GX_KERNEL_SLICES || (kernel == GX_KERNEL_AUTO && very_long)) && !want_states;
    if (batchable && slices && plan_slice_launch(h, &L, mo)) {
        h->last_kernel = GX_KERNEL_SLICES;
        GX_HIP(launch_extract_slices(h->dev, L, image, at_global, h->num_cus, b, stream));
        return;
    }
    // records in LDS: the lane kernel (every lane keeps its own line in registers, 16 waves share the tables)
    const bool lanes = (kernel == GX_KERNEL_LANES || (kernel == GX_KERNEL_AUTO && (image_tier == 2 || (image_tier != 0 && long_lines)))) && !want_states;
    // (long lines: tiles of lines of similar length, see gx_lanes.hip; where LDS has no room for that, the slice kernel)
    bool lanes_ok = batchable && lanes && plan_lanes_launch(h, &L, mo, b.packed != nullptr, sorted, b.n);
    if (lanes_ok && kernel == GX_KERNEL_AUTO && long_lines && L.sort_chunk == 0) {
        GxLds S;
        if (plan_slice_launch(h, &S, mo)) {
            h->last_kernel = GX_KERNEL_SLICES;
            GX_HIP(launch_extract_slices(h->dev, S, image, at_global, h->num_cus, b, stream));
            return;
        }
    }
    if (lanes_ok) {
        std::lock_guard<std::mutex> lock(h->slot_mu);
        const SlotUse u = take_slot(h, b, stream);
        const int slot = u.slot;
        // (the lines the lane kernel leaves: longer than its 16-bit positions -- with compact rows, than the 65 534 they can hold)
        const uint32_t lanes_limit = b.packed ? 65534u : 65535u;
        const bool followup = plan_followup(h, b, u, lanes_limit, launched);
        if (launched) { launched->limit = lanes_limit; launched->by_length = 1; }
        if (L.sort_chunk) {
            b.chunk_ctr = h->d_slots + gx_handle::N_SLOTS + slot;
            b.chunk_base = h->chunk_tickets[slot];
        }
        unsigned long long* stamps = nullptr;
#ifdef GX_DEV
        stamps = h->dev_stamps;
#endif
        h->last_kernel = GX_KERNEL_LANES;
        GX_HIP(launch_extract_lanes(h->dev, L, image, at_global, h->num_cus, b, stream, stamps));
        if (L.sort_chunk) h->chunk_tickets[slot] += lanes_sorted_tickets(b.n, L.sort_chunk, h->num_cus);  // (what the launch will draw)
        if (followup) GX_HIP(launch_extract_oversize(h->dev, b, lanes_limit, 1, stream));
        done_slot(h, u, stream);
        return;
    }
    if (batchable && (!want_states || image_tier <= 1u) && plan_tile_launch(h, line_bytes_hint, &L, mo)) {
        // a slot for the "lines I could not stage" word of this launch, free again once its follow-up kernel has run
        // (submission of tile launches is serialised per handle; the launches themselves are asynchronous)
        std::lock_guard<std::mutex> lock(h->slot_mu);
        const SlotUse u = take_slot(h, b, stream);
        const bool followup = plan_followup(h, b, u, L.stage_bytes >= 63u ? L.stage_bytes - 63u : 0u, launched);
        if (launched) { launched->limit = L.stage_bytes; launched->by_length = 0; }
        unsigned long long* stamps = nullptr;
#ifdef GX_DEV
        stamps = h->dev_stamps;
#endif
        h->last_kernel = GX_KERNEL_TILES;
        GX_HIP(launch_extract_tile(h->dev, L, image, at_global, h->num_cus, b, stream, stamps));
        h->steal_parity[u.slot] ^= 1u;
        if (followup) GX_HIP(launch_extract_oversize(h->dev, b, L.stage_bytes, 0, stream));
        done_slot(h, u, stream);
    } else {
        h->last_kernel = GX_KERNEL_PER_LINE;
        GX_HIP(launch_extract_generic(h->dev, b, stream));
    }
}

int finish_create(std::unique_ptr<gx_handle>& h, uint32_t flags, gx_handle** out) {
    h->create_flags = flags;
    h->blob = pack_blob(h->T);
    if (!(flags & GX_CREATE_HOST_ONLY)) upload(h.get());
    else choose_tile_image(h.get());
    *out = h.release();
    return GX_OK;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t bytes) { GX_HIP(hipMalloc(&p, bytes ? bytes : 16)); }
};

// gx_split_lines has no handle to keep its workspace on (an eighth of the text since the text-read-once split: a hipMalloc + hipFree
// of 250 MB per call were 0.1 ms of a 0.7 ms call): one workspace per device, kept between calls, grown as needed; a call holds the
// lock OF ITS DEVICE while it runs (calls on one device take turns: they would on the device anyway; calls on different devices -- one
// process, eight GPUs -- do not wait for each other).  gx_release_scratch(device) gives a device's workspace back.
struct SplitScratch {
    std::mutex mu[64];
    void* p[64] = {};
    size_t cap[64] = {};
    void release(int dev) {
        if (dev < 0 || dev >= 64) return;
        std::lock_guard<std::mutex> lock(mu[dev]);
        if (p[dev]) { (void)hipFree(p[dev]); p[dev] = nullptr; cap[dev] = 0; }
    }
    void* get(int dev, size_t bytes) {   // (the caller holds mu[dev])
        if (dev < 0 || dev >= 64) throw GxError(GX_E_DEVICE, "gx_split_lines: device ordinal beyond 63");
        if (cap[dev] < bytes) {
            if (p[dev]) { (void)hipFree(p[dev]); p[dev] = nullptr; cap[dev] = 0; }
            const size_t want = bytes + bytes / 8 + 256;
            GX_HIP(hipMalloc(&p[dev], want));
            cap[dev] = want;
        }
        return p[dev];
    }
};
SplitScratch g_split_scratch;

// the handle's scratch buffer `which`, at least `bytes` long (the caller holds h->mu)
void* handle_scratch(gx_handle* h, int which, size_t bytes) {
    if (h->scratch_cap[which] < bytes) {
        if (h->scratch[which]) { (void)hipFree(h->scratch[which]); h->scratch[which] = nullptr; h->scratch_cap[which] = 0; }
        const size_t cap = bytes + bytes / 8 + 256;
        GX_HIP(hipMalloc(&h->scratch[which], cap));
        h->scratch_cap[which] = cap;
    }
    return h->scratch[which];
}

}  // namespace

extern "C" {

const char* gx_last_error(void) { return g_last_error.c_str(); }

int gx_release_scratch(int device) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();   // (the runtime keeps the error for the next hipGetLastError(): a later launch's check would see it)
        return fail(GX_E_DEVICE, "gx_release_scratch: no such device");
    }
    g_split_scratch.release(device);
    (void)hipSetDevice(prev);
    return GX_OK;
}

int gx_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int gx_create_from_patterns(const char* const* automaton_rx, const char* const* jdk_rx, int32_t n, uint32_t flags,
                            gx_handle** out) {
    if (!automaton_rx || !out || n <= 0) return fail(GX_E_ARG, "gx_create_from_patterns: bad argument");
    try {
        std::vector<ustr> a, j;
        for (int32_t i = 0; i < n; ++i) {
            if (!automaton_rx[i] || (jdk_rx && !jdk_rx[i])) return fail(GX_E_ARG, "gx_create_from_patterns: null pattern");
            a.push_back(utf8_to_u16(automaton_rx[i]));
            if (jdk_rx) j.push_back(utf8_to_u16(jdk_rx[i]));
        }
        std::unique_ptr<gx_handle> h(new gx_handle());
        h->T = compile_tables(a, jdk_rx ? &j : nullptr);
        return finish_create(h, flags, out);
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

int gx_create_from_blob(const void* blob, size_t size, uint32_t flags, gx_handle** out) {
    if (!blob || !out) return fail(GX_E_ARG, "gx_create_from_blob: bad argument");
    try {
        std::unique_ptr<gx_handle> h(new gx_handle());
        h->T = unpack_blob(blob, size);
        return finish_create(h, flags, out);
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

size_t gx_blob_size(const gx_handle* h) { return h ? h->blob.size() : 0; }

int gx_blob_copy(const gx_handle* h, void* dst, size_t cap) {
    if (!h || !dst || cap < h->blob.size()) return fail(GX_E_ARG, "gx_blob_copy: bad argument");
    memcpy(dst, h->blob.data(), h->blob.size());
    return GX_OK;
}

void gx_destroy(gx_handle* h) {
    if (!h) return;
    if (h->dimage) (void)hipFree(h->dimage);
    if (h->d_lds_image) (void)hipFree(h->d_lds_image);
    if (h->d_lds_image_mo) (void)hipFree(h->d_lds_image_mo);
    if (h->d_l2_image) (void)hipFree(h->d_l2_image);
    if (h->d_lds_image_hop) (void)hipFree(h->d_lds_image_hop);
    if (h->d_lds_image_hop_small) (void)hipFree(h->d_lds_image_hop_small);
    for (void* q : {h->d_lds_image_hop_mo, h->d_lds_image_hop_mo_small, h->d_hop_mo_global}) if (q) (void)hipFree(q);
    if (h->d_hop_global) (void)hipFree(h->d_hop_global);
    if (h->hint_probe) { (void)hipHostFree(h->hint_probe); (void)hipEventDestroy(h->hint_event); }
    for (auto& sl : h->host_slot) {
        for (void* p : {sl.d_bytes, sl.d_off, sl.d_res, sl.d_caps, sl.d_states, static_cast<void*>(sl.d_over)}) if (p) (void)hipFree(p);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (h->d_slots) {
        (void)hipFree(h->d_slots);
        if (h->shared_event) (void)hipEventDestroy(h->shared_event);
        if (h->h_broken) (void)hipHostFree(h->h_broken);
        for (uint32_t* q : h->d_steal) if (q) (void)hipFree(q);
    }
    for (auto& e : h->jsonl) if (e.second.d) (void)hipFree(e.second.d);
    for (void* q : h->scratch) if (q) (void)hipFree(q);
    if (h->pool) (void)hipMemPoolDestroy(h->pool);
    if (h->svc.enabled) {
        if (h->svc.started) {   // tell the wave to leave, wait for it
            __atomic_store_n(&h->svc.host[1], (h->svc.host[1] & 0xFFFFu) | 0x10000u, __ATOMIC_RELEASE);
            (void)hipStreamSynchronize(h->svc.stream);
        }
        if (h->svc.stream) (void)hipStreamDestroy(h->svc.stream);
        if (h->svc.host) (void)hipHostFree(h->svc.host);
    }
    if (h->multi_stream) (void)hipStreamDestroy(h->multi_stream);
    if (h->gather_stream) (void)hipStreamDestroy(h->gather_stream);
    if (h->gather_event) (void)hipEventDestroy(h->gather_event);
    if (h->d_pike_scratch) (void)hipFree(h->d_pike_scratch);
    if (h->pike_event) (void)hipEventDestroy(h->pike_event);
    if (h->one_dev) (void)hipFree(h->one_dev);
    if (h->one_host) (void)hipHostFree(h->one_host);
    delete h;
}

int32_t gx_num_extractions(const gx_handle* h) { return h ? h->T.n_rules : 0; }
int32_t gx_num_groups(const gx_handle* h, int32_t k) {
    if (!h || k < 0 || k >= static_cast<int32_t>(h->T.rules.size())) return 0;
    return h->T.rules[k].n_groups;
}
int32_t gx_max_groups(const gx_handle* h) { return h ? h->T.max_groups : 0; }

int64_t gx_stat(const gx_handle* h, int32_t which) {
    if (!h) return -1;
    switch (which) {
    case 0: return h->T.m_states;
    case 1: return h->T.ncls;
    case 2: { int64_t s = 0; for (auto& r : h->T.rules) s += r.n_states; return s; }
    case 3: { int64_t m = 0; for (auto& r : h->T.rules) m = std::max<int64_t>(m, r.n_regs); return m; }
    case 4: return static_cast<int64_t>(h->blob.size());
    case 5: { GxLds L; return plan_tile_launch(h, 0, &L) ? static_cast<int64_t>(L.total_bytes) : 0; }
    case 6: { GxLds L; return plan_tile_launch(h, 0, &L) ? static_cast<int64_t>(L.nwaves) : 0; }
    case 7: return !h->tile_ok ? 0 : h->lds.tier == 3 ? 4 : h->tile_global ? 2 : h->lds.tier == 2 ? 3 : 1;
    case 8: return h->T.has_capture ? 1 : 0;
    case 10: { GxLds L; return plan_lanes_launch(h, &L, false, true) ? static_cast<int64_t>(L.nwaves) : 0; }   // lane kernel: waves per CU, compact rows
    case 11: { GxLds L; return plan_lanes_launch(h, &L, true, false) ? static_cast<int64_t>(L.nwaves) : 0; }   // ... match-only
    case 12: return h->tile_ok ? static_cast<int64_t>(h->lds.table_bytes) : 0;
    case 13: return h->tile_ok ? static_cast<int64_t>(h->lds.regs_wave_bytes) : 0;
    case 14: return h->hop_ok ? static_cast<int64_t>(h->hop.n_states) : 0;         // hop tier: states (0: no hop image)
    case 15: return h->hop_ok ? static_cast<int64_t>(h->hop.full.n_hot) : 0;       // ... whose records live in LDS (tile kernel)
    case 21: return h->hop_ok ? static_cast<int64_t>(h->hop.small.n_hot) : 0;      // ... (hop slice kernel)
    case 16: return h->hop_ok ? static_cast<int64_t>(h->hop.n_reachable_hot) : 0;  // ... that well-formed lines reach
    case 17: return h->hop_ok ? static_cast<int64_t>(h->hop.n_chains) : 0;         // ... that have a chain
    case 18: { GxLds L; return plan_hop_launch(h, 0, &L) ? static_cast<int64_t>(L.nwaves) : 0; }  // hop tier: waves per CU
    case 20: return h->hop_ok ? static_cast<int64_t>(h->hop.full.n_lds_rows) : 0;       // ... whose dense row is in LDS too (branching states)
    case 22: return h->hop_mo_ok ? static_cast<int64_t>(h->hop_mo.n_states) : 0;   // hop tier of the match automaton alone (match-only batches): states
    case 24: return static_cast<int64_t>(h->promises_broken.load());
    case 30: return static_cast<int64_t>(h->peer_image_bytes);   // table bytes copied from another handle's device (gx_create_on_devices)
    case 25: return h->last_kernel.load();
    case 26: return h->hop_reason;
    case 28: return static_cast<int64_t>(h->svc.enabled ? h->svc.launches : -1);
    case 27: { int64_t c = 0; for (auto& r : h->T.rules) c += r.pike ? 1 : 0; return c; }
    case 23: return h->hop_mo_ok ? static_cast<int64_t>(h->hop_mo.full.n_hot) : 0; // ... whose records are in LDS
    case 19: { GxLds L; return plan_hop_slice_launch(h, &L) ? static_cast<int64_t>(L.nwaves) : 0; }  // ... of the hop slice kernel
    case 9: return !h->tile_ok ? 0 : !h->has_mo ? gx_stat(h, 7) : h->lds_mo.tier == 3 ? 4 : h->lds_mo.tier == 2 ? 3 : h->lds_mo.tier == 1 ? 2 : 1;
    default: return -1;
    }
}

// Accepts the current gx_batch_opts and every earlier, shorter layout of it (struct_size says which).
static bool read_opts(const gx_batch_opts* opts, gx_batch_opts* o) {
    *o = gx_batch_opts{};
    if (!opts) return true;
    const size_t v1 = offsetof(gx_batch_opts, strip_eol);  // the first layout; later ones only appended fields
    if (opts->struct_size < v1 || opts->struct_size > sizeof(gx_batch_opts) || (opts->struct_size & 3u)) return false;
    memcpy(o, opts, opts->struct_size);
    return true;
}

int gx_split_lines(const uint8_t* bytes, uint64_t size, void* offsets, uint64_t cap_lines, uint64_t* n_lines, uint8_t* line_flags,
                   const gx_batch_opts* opts) {
    return gx_split_lines_max(bytes, size, offsets, cap_lines, n_lines, line_flags, nullptr, opts);
}

int gx_split_lines_max(const uint8_t* bytes, uint64_t size, void* offsets, uint64_t cap_lines, uint64_t* n_lines, uint8_t* line_flags,
                       uint64_t* max_line_bytes, const gx_batch_opts* opts) {
    if (!offsets || !n_lines || (size && !bytes)) return fail(GX_E_ARG, "gx_split_lines: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    if (!o.offsets64 && size > 0xFFFFFFFFull) return fail(GX_E_ARG, "gx_split_lines: buffers of 4 GiB and more need offsets64");
    try {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            throw GxError(GX_E_DEVICE, "no HIP device available (libgorp_hip needs a gfx950 GPU; there is no CPU fallback)");
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        const size_t off_w = o.offsets64 ? 8 : 4;
        DevBuf d_bytes, d_off, d_flags;
        int dev = 0;
        GX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return fail(GX_E_DEVICE, "gx_split_lines: device ordinal beyond 63");
        std::lock_guard<std::mutex> ws_lock(g_split_scratch.mu[dev]);
        struct { void* p; } ws{g_split_scratch.get(dev, split_workspace_bytes(size, line_flags != nullptr))};
        const uint8_t* src = bytes;
        void* dst_off = offsets;
        uint8_t* dst_flags = line_flags;
        if (o.device_pointers) {
            if (reinterpret_cast<uintptr_t>(bytes) & 15u) return fail(GX_E_ARG, "gx_split_lines: device buffer must be 16-byte aligned");
        } else {
            d_bytes.alloc(size);
            d_off.alloc((cap_lines + 1) * off_w);
            if (line_flags) d_flags.alloc(cap_lines);
            if (size) GX_HIP(hipMemcpyAsync(d_bytes.p, bytes, size, hipMemcpyHostToDevice, stream));
            src = static_cast<const uint8_t*>(d_bytes.p);
            dst_off = d_off.p;
            dst_flags = line_flags ? static_cast<uint8_t*>(d_flags.p) : nullptr;
        }
        uint64_t* d_n = nullptr;
        uint64_t* d_max = nullptr;
        GX_HIP(launch_split_lines(src, size, dst_off, o.offsets64 ? 1 : 0, cap_lines, dst_flags, ws.p, &d_n, stream, max_line_bytes ? &d_max : nullptr));
        uint64_t n_and_max[2] = {0, 0};   // (n_lines and max_line are neighbours in the workspace)
        GX_HIP(hipMemcpyAsync(n_and_max, d_n, max_line_bytes ? 16 : 8, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        const uint64_t n = n_and_max[0];
        *n_lines = n;
        if (max_line_bytes) *max_line_bytes = n_and_max[1];
        if (n > cap_lines) return fail(GX_E_LIMIT, "gx_split_lines: the buffer holds more lines than cap_lines");
        if (!o.device_pointers) {
            GX_HIP(hipMemcpy(offsets, d_off.p, (n + 1) * off_w, hipMemcpyDeviceToHost));
            if (line_flags && n) GX_HIP(hipMemcpy(line_flags, d_flags.p, n, hipMemcpyDeviceToHost));
        }
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
}

// One JSON template per extraction for ExtractionResult.asMap(idAs) (core/ExtractionResult.java:65-88), uploaded once
// per distinct id_as.
static const GxJsonl& jsonl_templates(gx_handle* h, const char* id_as) {
    const std::string key = id_as ? std::string("1") + id_as : std::string("0");
    auto it = h->jsonl.find(key);
    if (it != h->jsonl.end()) return it->second.dev;
    const Tables& T = h->T;
    if (static_cast<int>(h->meta.size()) != T.n_rules)
        throw GxError(GX_E_ARG, "extraction names are unknown: create the handle with gx_create_from_definition or call "
                                "gx_set_extraction_meta for every extraction");
    std::vector<uint32_t> seg_off(1, 0), lit_off, lit_len, fixed_len;
    std::vector<int32_t> group;
    std::vector<uint8_t> lits;
    for (int k = 0; k < T.n_rules; ++k) {
        const dsl::Extraction& x = h->meta[k];
        if (static_cast<int>(x.extractor_names.size()) != T.rules[k].n_groups)
            throw GxError(GX_E_ARG, "extractor names of extraction '" + x.name + "' do not match its capture groups");
        // LinkedHashMap: a key put again keeps its position and takes the new value
        struct Entry { std::string key; int g; std::string raw; };
        std::vector<Entry> entries;
        auto put = [&](const std::string& key_utf8, int g, const std::string& raw) {
            for (auto& e : entries) if (e.key == key_utf8) { e.g = g; e.raw = raw; return; }
            entries.push_back({key_utf8, g, raw});
        };
        if (id_as) put(id_as, -1, dsl::json_quote(x.name));
        for (size_t g = 0; g < x.extractor_names.size(); ++g) put(x.extractor_names[g], static_cast<int>(g), "");
        if (!x.append_json.empty())
            for (auto& kv : dsl::json_object_entries(x.append_json)) put(kv.first, -1, kv.second);
        std::string lit = "{";
        uint32_t fixed = 0;
        auto close_segment = [&](int g) {
            while (lits.size() % 4) lits.push_back(0);  // the write kernel reads literals as aligned 32-bit words
            lit_off.push_back(static_cast<uint32_t>(lits.size()));
            lit_len.push_back(static_cast<uint32_t>(lit.size()));
            group.push_back(g);
            lits.insert(lits.end(), lit.begin(), lit.end());
            fixed += static_cast<uint32_t>(lit.size());
            lit.clear();
        };
        for (size_t e = 0; e < entries.size(); ++e) {
            lit += (e ? "," : "") + dsl::json_quote(entries[e].key) + ":";
            if (entries[e].g >= 0) close_segment(entries[e].g);
            else lit += entries[e].raw;
        }
        lit += "}\n";
        close_segment(-1);
        seg_off.push_back(static_cast<uint32_t>(group.size()));
        fixed_len.push_back(fixed);
    }
    while (lits.empty() || lits.size() % 4) lits.push_back(0);
    Image img;
    const size_t o_seg = img.put(seg_off), o_lo = img.put(lit_off), o_ll = img.put(lit_len), o_g = img.put(group), o_f = img.put(fixed_len),
                 o_l = img.put(lits);
    gx_handle::JsonlImage ji;
    GX_HIP(hipMalloc(&ji.d, img.bytes.size()));
    GX_HIP(hipMemcpy(ji.d, img.bytes.data(), img.bytes.size(), hipMemcpyHostToDevice));
    const uint8_t* base = static_cast<const uint8_t*>(ji.d);
    ji.dev.seg_off = reinterpret_cast<const uint32_t*>(base + o_seg);
    ji.dev.lit_off = reinterpret_cast<const uint32_t*>(base + o_lo);
    ji.dev.lit_len = reinterpret_cast<const uint32_t*>(base + o_ll);
    ji.dev.group = reinterpret_cast<const int32_t*>(base + o_g);
    ji.dev.fixed_len = reinterpret_cast<const uint32_t*>(base + o_f);
    ji.dev.lits = base + o_l;
    ji.dev.lits_bytes = static_cast<uint32_t>(lits.size());
    ji.dev.n_rules = static_cast<uint32_t>(T.n_rules);
    ji.dev.n_segs = static_cast<uint32_t>(group.size());
    return h->jsonl.emplace(key, ji).first->second.dev;
}

int gx_results_to_jsonl(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, const int32_t* match_id, const int32_t* caps,
                        const char* id_as, uint8_t* out, uint64_t out_cap, uint64_t* out_size, uint64_t* line_out_offsets,
                        const gx_batch_opts* opts) {
    if (!h || !offsets || !out_size || (n && !match_id)) return fail(GX_E_ARG, "gx_results_to_jsonl: bad argument");
    if (!h->on_device) return fail(GX_E_DEVICE, "handle was created host-only; no device tables (there is no CPU fallback)");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    const size_t slots = 2 * static_cast<size_t>(h->T.max_groups);
    if (n && slots && !caps) return fail(GX_E_ARG, "gx_results_to_jsonl: caps is NULL");
    if (slots > 128) return fail(GX_E_LIMIT, "gx_results_to_jsonl: more than 64 capture groups per extraction");
    try {
        GX_HIP(hipSetDevice(h->device));
        std::lock_guard<std::mutex> lock(h->mu);
        const GxJsonl& tm = jsonl_templates(h, id_as);
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        const size_t off_w = o.offsets64 ? 8 : 4;
        DevBuf d_bytes, d_off, d_mid, d_caps, d_out;
        void* ws = handle_scratch(h, 0, jsonl_workspace_bytes(n));
        GxBatch b{};
        b.n = n;
        b.offsets64 = o.offsets64 ? 1 : 0;
        uint64_t* loff = line_out_offsets;
        if (o.device_pointers) {
            b.data = bytes; b.offsets = offsets; b.match_id = const_cast<int32_t*>(match_id); b.caps = const_cast<int32_t*>(caps);
            if (!loff) loff = static_cast<uint64_t*>(handle_scratch(h, 1, (n + 1) * 8));
        } else {
            uint64_t total_in = 0;
            if (n) total_in = o.offsets64 ? static_cast<const uint64_t*>(offsets)[n] : static_cast<const uint32_t*>(offsets)[n];
            d_bytes.alloc(total_in); d_off.alloc((n + 1) * off_w); d_mid.alloc(n * 4); d_caps.alloc(n * slots * 4);
            if (total_in) GX_HIP(hipMemcpyAsync(d_bytes.p, bytes, total_in, hipMemcpyHostToDevice, stream));
            GX_HIP(hipMemcpyAsync(d_off.p, offsets, (n + 1) * off_w, hipMemcpyHostToDevice, stream));
            if (n) GX_HIP(hipMemcpyAsync(d_mid.p, match_id, n * 4, hipMemcpyHostToDevice, stream));
            if (n && slots) GX_HIP(hipMemcpyAsync(d_caps.p, caps, n * slots * 4, hipMemcpyHostToDevice, stream));
            b.data = d_bytes.p; b.offsets = d_off.p; b.match_id = static_cast<int32_t*>(d_mid.p); b.caps = static_cast<int32_t*>(d_caps.p);
            loff = static_cast<uint64_t*>(handle_scratch(h, 1, (n + 1) * 8));
        }
        // mean line length, for the LDS staging of the kernels (device pointers: from the two ends of the offsets array)
        uint64_t first_off = 0, last_off = 0;
        if (n) {
            if (o.device_pointers) {
                GX_HIP(hipMemcpyAsync(&first_off, offsets, off_w, hipMemcpyDeviceToHost, stream));
                GX_HIP(hipMemcpyAsync(&last_off, static_cast<const uint8_t*>(offsets) + n * off_w, off_w, hipMemcpyDeviceToHost, stream));
                GX_HIP(hipStreamSynchronize(stream));
            } else {
                first_off = o.offsets64 ? static_cast<const uint64_t*>(offsets)[0] : static_cast<const uint32_t*>(offsets)[0];
                last_off = o.offsets64 ? static_cast<const uint64_t*>(offsets)[n] : static_cast<const uint32_t*>(offsets)[n];
            }
        }
        const uint32_t mean_in = n ? static_cast<uint32_t>(std::min<uint64_t>((last_off - first_off + n - 1) / n, 1u << 20)) : 1u;
        GX_HIP(launch_jsonl_sizes(tm, b, static_cast<int>(slots), o.utf8_passthrough ? 1 : 0, mean_in, loff, ws, stream));
        uint64_t total = 0;
        GX_HIP(hipMemcpyAsync(&total, loff + n, 8, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        *out_size = total;
        if (!o.device_pointers && line_out_offsets) GX_HIP(hipMemcpy(line_out_offsets, loff, (n + 1) * 8, hipMemcpyDeviceToHost));
        if (!out) return GX_OK;  // size query
        if (total > out_cap) return fail(GX_E_LIMIT, "gx_results_to_jsonl: out_cap is smaller than the text (see *out_size)");
        uint8_t* dst = out;
        if (!o.device_pointers) { d_out.alloc(total); dst = static_cast<uint8_t*>(d_out.p); }
        const uint32_t mean_out = n ? static_cast<uint32_t>(std::min<uint64_t>((total + n - 1) / n, 1u << 20)) : 1u;
        GX_HIP(launch_jsonl_write(tm, b, static_cast<int>(slots), o.utf8_passthrough ? 1 : 0, mean_in, mean_out, loff, dst, ws, stream));
        if (!o.device_pointers && total) GX_HIP(hipMemcpyAsync(out, dst, total, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

int gx_text_to_jsonl(gx_handle* h, const uint8_t* text, uint64_t size, const char* id_as, uint8_t* out, uint64_t out_cap, uint64_t* out_size,
                     uint64_t* n_lines, uint64_t* n_matched, uint64_t* n_exceptions, const gx_batch_opts* opts) {
    if (!h || !out_size || (size && !text)) return fail(GX_E_ARG, "gx_text_to_jsonl: bad argument");
    if (!h->on_device) return fail(GX_E_DEVICE, "handle was created host-only; no device tables (there is no CPU fallback)");
    if (size > 0xFFFFFFFFull) return fail(GX_E_LIMIT, "gx_text_to_jsonl: split texts of 4 GiB and more at a line boundary");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    const size_t slots = 2 * static_cast<size_t>(h->T.max_groups);
    if (slots > 128) return fail(GX_E_LIMIT, "gx_text_to_jsonl: more than 64 capture groups per extraction");
    try {
        GX_HIP(hipSetDevice(h->device));
        std::lock_guard<std::mutex> lock(h->mu);
        const GxJsonl& tm = jsonl_templates(h, id_as);
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        DevBuf d_text, d_out;   // (host buffers only; everything between lives in the handle's scratch: 2 split workspace, 3 offsets, 4 ids, 5 captures, 6 counts)
        const uint8_t* src = text;
        if (!o.device_pointers) {
            d_text.alloc(size);
            if (size) GX_HIP(hipMemcpyAsync(d_text.p, text, size, hipMemcpyHostToDevice, stream));
            src = static_cast<const uint8_t*>(d_text.p);
        } else if (reinterpret_cast<uintptr_t>(text) & 15u) {
            return fail(GX_E_ARG, "gx_text_to_jsonl: device text must be 16-byte aligned");
        }
        // 1. lines: offsets for the guess "64 bytes or more per line"; a text with shorter lines is split a second time
        void* ws_split = handle_scratch(h, 2, split_workspace_bytes(size));
        uint64_t cap = size / 64 + 4096;
        void* d_off2 = handle_scratch(h, 3, (cap + 1) * 4);
        uint64_t* d_n = nullptr;
        uint64_t* d_max = nullptr;
        // (the split pass also leaves a bit per byte that takes one more byte inside a JSON string, and says whether some byte takes five
        // more -- a control character --: without one, the sizes pass below does not read the text again)
        uint16_t* esc_bits = static_cast<uint16_t*>(handle_scratch(h, 7, ((size + 32767) / 32768) * 4096 + 64));   // (written in whole blocks of 32 KiB of text)
        GX_HIP(launch_split_lines(src, size, d_off2, 0, cap, nullptr, ws_split, &d_n, stream, &d_max, esc_bits, o.utf8_passthrough ? 1 : 0));
        uint64_t n_and_max[3] = {0, 0, 0};   // (the line count, the longest line and the control-character word are neighbours in the workspace)
        GX_HIP(hipMemcpyAsync(n_and_max, d_n, 24, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        const uint64_t n = n_and_max[0];
        uint64_t longest = n_and_max[1];
        const bool sizes_from_bits = n_and_max[2] == 0;
        if (n > cap) {
            d_off2 = handle_scratch(h, 3, (n + 1) * 4);
            GX_HIP(launch_split_lines(src, size, d_off2, 0, n, nullptr, ws_split, &d_n, stream));
            longest = 0;   // (measured over the first `cap` lines only: no promise)
        }
        // 2. the path
        void* d_mid = handle_scratch(h, 4, n * 4 + 16);
        void* d_caps = handle_scratch(h, 5, n * slots * 4 + 16);
        GxBatch b{};
        b.data = src; b.offsets = d_off2; b.n = n; b.match_id = static_cast<int32_t*>(d_mid);
        b.caps = h->T.has_capture ? static_cast<int32_t*>(d_caps) : nullptr;
        b.match_only = h->T.has_capture ? 0 : 1;
        b.strip_eol = 1;
        b.max_line_bytes = static_cast<uint32_t>(std::min<uint64_t>(longest, 0xFFFFFFFFull));   // (what the split pass saw: no follow-up launch)
        const uint32_t mean_in = n ? static_cast<uint32_t>(std::min<uint64_t>((size + n - 1) / n, 1u << 20)) : 1u;
        launch_batch(h, b, mean_in, GX_KERNEL_AUTO, stream);
        if (!h->T.has_capture && n && slots) GX_HIP(hipMemsetAsync(d_caps, 0xFF, n * slots * 4, stream));
        b.caps = static_cast<int32_t*>(d_caps);
        void* d_counts = handle_scratch(h, 6, 16);
        GX_HIP(launch_count_outcomes(b.match_id, n, static_cast<unsigned long long*>(d_counts), stream));
        // 3. the text
        void* ws_json = handle_scratch(h, 0, jsonl_workspace_bytes(n));
        uint64_t* loff = static_cast<uint64_t*>(handle_scratch(h, 1, (n + 1) * 8));
        GX_HIP(launch_jsonl_sizes(tm, b, static_cast<int>(slots), o.utf8_passthrough ? 1 : 0, mean_in, loff, ws_json, stream,
                                  sizes_from_bits ? reinterpret_cast<const uint32_t*>(esc_bits) : nullptr));
        uint64_t total = 0;
        unsigned long long counts[2] = {0, 0};
        GX_HIP(hipMemcpyAsync(&total, loff + n, 8, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipMemcpyAsync(counts, d_counts, 16, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        // (the extraction ran on the split pass's own longest line; a kernel that met a longer one after all left rows unwritten)
        if (promise_broken_since(h, stream)) throw GxError(GX_E_ARG, "internal: gx_text_to_jsonl: a line longer than the split pass reported");
        *out_size = total;
        if (n_lines) *n_lines = n;
        if (n_matched) *n_matched = counts[0];
        if (n_exceptions) *n_exceptions = counts[1];
        if (!out) return GX_OK;
        if (total > out_cap) return fail(GX_E_LIMIT, "gx_text_to_jsonl: out_cap is smaller than the text (see *out_size)");
        uint8_t* dst = out;
        if (!o.device_pointers) { d_out.alloc(total); dst = static_cast<uint8_t*>(d_out.p); }
        const uint32_t mean_out = n ? static_cast<uint32_t>(std::min<uint64_t>((total + n - 1) / n, 1u << 20)) : 1u;
        GX_HIP(launch_jsonl_write(tm, b, static_cast<int>(slots), o.utf8_passthrough ? 1 : 0, mean_in, mean_out, loff, dst, ws_json, stream));
        if (!o.device_pointers && total) GX_HIP(hipMemcpyAsync(out, dst, total, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

int gx_pack_results(const int32_t* match_id, const int32_t* caps, uint64_t n, int32_t slots, uint16_t* packed, uint64_t* n_overflow,
                    const gx_batch_opts* opts) {
    if (slots < 0 || !n_overflow || (n && (!match_id || !packed || (slots && !caps)))) return fail(GX_E_ARG, "gx_pack_results: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    try {
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        DevBuf cnt;
        cnt.alloc(8);
        GX_HIP(launch_pack_results(match_id, caps, n, slots, packed, static_cast<unsigned long long*>(cnt.p), stream));
        unsigned long long over = 0;
        GX_HIP(hipMemcpyAsync(&over, cnt.p, 8, hipMemcpyDeviceToHost, stream));
        GX_HIP(hipStreamSynchronize(stream));
        *n_overflow = over;
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
}

int gx_unpack_results(const uint16_t* packed, uint64_t n, int32_t slots, int32_t* match_id, int32_t* caps, const gx_batch_opts* opts) {
    if (slots < 0 || (n && (!match_id || !packed || (slots && !caps)))) return fail(GX_E_ARG, "gx_unpack_results: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    try {
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        GX_HIP(launch_unpack_results(packed, n, slots, match_id, caps, stream));
        if (!o.no_sync) GX_HIP(hipStreamSynchronize(stream));
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
}

int gx_unpack_results8(const uint8_t* rows, uint64_t n, int32_t slots, int32_t* match_id, int32_t* caps, const gx_batch_opts* opts) {
    if (slots < 0 || (n && (!match_id || !rows || (slots && !caps)))) return fail(GX_E_ARG, "gx_unpack_results8: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    try {
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        GX_HIP(launch_unpack_results8(rows, n, slots, match_id, caps, stream));
        if (!o.no_sync) GX_HIP(hipStreamSynchronize(stream));
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
}

int gx_set_extraction_meta(gx_handle* h, int32_t k, const char* name, const char* const* extractor_names, int32_t n_names,
                           const char* append_json) {
    if (!h || !name || k < 0 || k >= h->T.n_rules || n_names < 0 || (n_names && !extractor_names))
        return fail(GX_E_ARG, "gx_set_extraction_meta: bad argument");
    if (n_names != h->T.rules[k].n_groups) return fail(GX_E_ARG, "gx_set_extraction_meta: n_names must equal gx_num_groups(h, k)");
    try {
        std::lock_guard<std::mutex> lock(h->mu);
        if (static_cast<int>(h->meta.size()) != h->T.n_rules) h->meta.assign(h->T.n_rules, dsl::Extraction());
        dsl::Extraction& x = h->meta[k];
        x.name = name;
        x.extractor_names.assign(extractor_names, extractor_names + n_names);
        x.append_json = append_json ? dsl::canonical_json_object(append_json) : std::string();
        for (auto& e : h->jsonl) if (e.second.d) (void)hipFree(e.second.d);
        h->jsonl.clear();
        h->append_entries.clear();
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

// One host-pointer batch through the workers of gx_handle::host_slot.  `proto` carries the batch's modes (wide,
// offsets64, match_only, strip_eol); lines [0, n) are cut into chunks of whole lines of about chunk_bytes, chunk c goes
// to worker c % HOST_WORKERS.  A chunk's lines keep their offsets: the kernels get a data pointer moved back by the
// chunk's first offset instead of rebased offsets.
static void host_pipeline(gx_handle* h, const GxBatch& proto, const uint8_t* bytes, const void* offsets, uint64_t n, uint64_t total,
                          int32_t* match_id, int32_t* caps, int32_t* states, bool compact, bool match_only, uint32_t hint, uint32_t kernel,
                          bool uneven, uint64_t* over_total) {
    if (n == 0) return;
    const size_t unit = proto.wide ? 2 : 1, off_w = proto.offsets64 ? 8 : 4;
    const size_t slots = 2 * static_cast<size_t>(h->T.max_groups);
    auto off_at = [&](uint64_t i) -> uint64_t {
        return proto.offsets64 ? static_cast<const uint64_t*>(offsets)[i] : static_cast<const uint32_t*>(offsets)[i];
    };
    // chunks: large enough to amortise a launch, small enough that the pipeline has several in flight
    const uint64_t total_bytes = total * unit;
    uint64_t chunk_bytes = std::max<uint64_t>(total_bytes / (4 * gx_handle::HOST_WORKERS), 8ull << 20);
    chunk_bytes = std::min<uint64_t>(chunk_bytes, 128ull << 20);
    std::vector<uint64_t> cuts(1, 0);
    while (cuts.back() < n) {
        const uint64_t a = cuts.back(), want = off_at(a) * unit + chunk_bytes;
        uint64_t lo = a + 1, hi = n;  // first line index whose start lies at or beyond `want` (at least one line per chunk)
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (off_at(mid) * unit >= want) hi = mid; else lo = mid + 1; }
        uint64_t b_ = lo;
        if (b_ - a > 0x7FFFFFF0ull) b_ = a + 0x7FFFFFF0ull;
        cuts.push_back(std::min<uint64_t>(b_, n));
    }
    const size_t n_chunks = cuts.size() - 1;
    const int workers = static_cast<int>(std::min<size_t>(gx_handle::HOST_WORKERS, n_chunks));
    std::atomic<uint64_t> over_sum{0};
    std::mutex err_mu;
    int err_code = GX_OK;
    std::string err_msg;
    auto work = [&](int w) {
        try {
            GX_HIP(hipSetDevice(h->device));
            gx_handle::HostSlot& sl = h->host_slot[w];
            if (!sl.stream) GX_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            auto grow = [&](void*& p, size_t& cap, size_t need) {
                if (need <= cap) return;
                if (p) GX_HIP(hipFree(p));
                p = nullptr; cap = 0;
                GX_HIP(hipMalloc(&p, need + need / 8 + 256));
                cap = need + need / 8 + 256;
            };
            if (compact && !sl.d_over) GX_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_over), 8));
            for (size_t c = static_cast<size_t>(w); c < n_chunks; c += static_cast<size_t>(workers)) {
                const uint64_t a = cuts[c], e = cuts[c + 1], m = e - a;
                const uint64_t b0 = off_at(a) * unit, nbytes = off_at(e) * unit - b0;
                grow(sl.d_bytes, sl.cap_bytes, nbytes + 64);
                grow(sl.d_off, sl.cap_off, (m + 1) * off_w);
                GxBatch b = proto;
                b.n = m;
                uint8_t* place = static_cast<uint8_t*>(sl.d_bytes) + 16;  // (a 16-byte aligned first line, room for the aligned span before it)
                b.data = place - b0;
                b.offsets = sl.d_off;
                if (nbytes) GX_HIP(hipMemcpyAsync(place, bytes + b0, nbytes, hipMemcpyHostToDevice, sl.stream));
                GX_HIP(hipMemcpyAsync(sl.d_off, static_cast<const uint8_t*>(offsets) + a * off_w, (m + 1) * off_w, hipMemcpyHostToDevice, sl.stream));
                if (states) { grow(sl.d_states, sl.cap_states, m * 4); b.state_out = static_cast<int32_t*>(sl.d_states); }
                const size_t row_bytes = (1 + slots) * (proto.narrow ? 1 : 2);  // compact rows: u8 or u16 entries
                if (compact) {
                    grow(sl.d_res, sl.cap_res, m * row_bytes);
                    GX_HIP(hipMemsetAsync(sl.d_over, 0, 8, sl.stream));
                    b.packed = static_cast<uint16_t*>(sl.d_res);
                    b.overflow = sl.d_over;
                } else {
                    grow(sl.d_res, sl.cap_res, m * 4);
                    b.match_id = static_cast<int32_t*>(sl.d_res);
                    if (!match_only) { grow(sl.d_caps, sl.cap_caps, m * slots * 4 + 16); b.caps = static_cast<int32_t*>(sl.d_caps); }
                }
                launch_batch(h, b, hint, kernel, sl.stream, uneven);
                unsigned long long over = 0;
                if (compact) {
                    GX_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t*>(caps) + a * row_bytes, sl.d_res, m * row_bytes, hipMemcpyDeviceToHost, sl.stream));
                    GX_HIP(hipMemcpyAsync(&over, sl.d_over, 8, hipMemcpyDeviceToHost, sl.stream));
                } else {
                    GX_HIP(hipMemcpyAsync(match_id + a, sl.d_res, m * 4, hipMemcpyDeviceToHost, sl.stream));
                    if (!match_only && slots) GX_HIP(hipMemcpyAsync(caps + a * slots, sl.d_caps, m * slots * 4, hipMemcpyDeviceToHost, sl.stream));
                }
                if (states) GX_HIP(hipMemcpyAsync(states + a, sl.d_states, m * 4, hipMemcpyDeviceToHost, sl.stream));
                GX_HIP(hipStreamSynchronize(sl.stream));  // this worker's buffers are free again; the other workers keep the bus busy
                over_sum += over;
            }
        } catch (GxError& e) {
            std::lock_guard<std::mutex> g(err_mu);
            if (err_code == GX_OK) { err_code = e.code; err_msg = e.what(); }
        } catch (std::bad_alloc&) {
            std::lock_guard<std::mutex> g(err_mu);
            if (err_code == GX_OK) { err_code = GX_E_NOMEM; err_msg = "out of memory"; }
        }
    };
    if (workers == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int w = 0; w < workers; ++w) pool.emplace_back(work, w);
        for (auto& t : pool) t.join();
    }
    if (err_code != GX_OK) throw GxError(err_code, err_msg);
    *over_total = over_sum.load();
}

// Do the lines of a batch differ much in length?  Lines run in lock step in groups of 64, a group takes as long as its longest
// line: over a sample of up to 64 groups spread over the batch, (sum of 64 x longest line) / (sum of lengths) > 1.25.
// offsets: n + 1 offsets in HOST memory.
static bool lines_are_uneven(const void* offsets, uint64_t n, bool off64) {
    if (n < 128) return false;
    auto at = [&](uint64_t i) -> uint64_t { return off64 ? static_cast<const uint64_t*>(offsets)[i] : static_cast<const uint32_t*>(offsets)[i]; };
    const uint64_t groups = n / 64, sample = std::min<uint64_t>(groups, 64), stride = groups / sample;
    uint64_t lock_step = 0, bytes = 0;
    for (uint64_t g = 0; g < sample; ++g) {
        const uint64_t i0 = g * stride * 64;
        uint64_t longest = 0;
        for (uint64_t i = i0; i < i0 + 64; ++i) longest = std::max(longest, at(i + 1) - at(i));
        lock_step += 64 * longest;
        bytes += at(i0 + 64) - at(i0);
    }
    return bytes > 0 && lock_step * 4 > bytes * 5;
}

// gx_extract_batch, and gx_match_batch when `states` is given (final product-DFA state per line, -1 = dead: the
// per-line generic kernel then, which is the one that keeps it)
static int extract_batch_impl(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, int32_t* match_id, int32_t* caps,
                              int32_t* states, const gx_batch_opts* opts) {
    if (!h || !offsets) return fail(GX_E_ARG, "gx_extract_batch: bad argument");
    if (!h->on_device) return fail(GX_E_DEVICE, "handle was created host-only; no device tables (there is no CPU fallback)");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    if (o.kernel > GX_KERNEL_HOP_SLICES) return fail(GX_E_ARG, "gx_batch_opts.kernel: unknown kernel");
    const bool match_only = o.match_only || states || !h->T.has_capture;
    const bool compact = o.compact_results && !match_only;  // rows of u16[1 + slots] (2: u8[1 + slots]) through `caps`
    if (o.compact_results > 2) return fail(GX_E_ARG, "gx_batch_opts.compact_results: 0, 1 (u16 rows) or 2 (u8 rows)");
    if (compact && o.compact_results == 2 && h->T.n_rules > 126)
        return fail(GX_E_ARG, "gx_batch_opts.compact_results = 2: u8 rows hold match ids -128 .. 127 (at most 126 extractions)");
    if (!compact && !match_id) return fail(GX_E_ARG, "gx_extract_batch: match_id is NULL");
    if (!match_only && !caps && n > 0 && (compact || h->T.max_groups > 0)) return fail(GX_E_ARG, "gx_extract_batch: caps is NULL");
    try {
        GX_HIP(hipSetDevice(h->device));
        hipStream_t stream = static_cast<hipStream_t>(o.stream);
        GxBatch b{};
        b.n = n;
        b.wide = o.utf16 ? 1 : 0;
        b.offsets64 = o.offsets64 ? 1 : 0;
        b.match_only = match_only ? 1 : 0;
        b.strip_eol = o.strip_eol ? 1 : 0;
        b.narrow = (compact && o.compact_results == 2) ? 1 : 0;
        const size_t off_w = o.offsets64 ? 8 : 4;
        if (o.device_pointers) {
            b.data = bytes; b.offsets = offsets;
            b.state_out = states;
            if (compact) {
                b.packed = reinterpret_cast<uint16_t*>(caps);
                b.overflow = static_cast<unsigned long long*>(o.overflow);
            } else {
                b.match_id = match_id;
                b.caps = match_only ? nullptr : caps;
            }
            uint32_t hint = o.line_bytes_hint;
            bool uneven = o.uneven_lines == 2;
            if (hint == 0 && n && o.no_sync) {
                // no hint and no synchronisation allowed: what the previous such batch measured (200 until one has), and
                // a probe of this batch for the next call
                std::lock_guard<std::mutex> lock(h->hint_mu);
                if (!h->hint_probe) {
                    GX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->hint_probe), 16, hipHostMallocDefault));
                    GX_HIP(hipEventCreateWithFlags(&h->hint_event, hipEventDisableTiming));
                }
                if (h->hint_pending && hipEventQuery(h->hint_event) == hipSuccess) {
                    const uint64_t first = h->hint_off64 ? h->hint_probe[0] : (h->hint_probe[0] & 0xFFFFFFFFull);
                    const uint64_t last = h->hint_off64 ? h->hint_probe[1] : (h->hint_probe[1] & 0xFFFFFFFFull);
                    if (h->hint_n && last >= first)
                        h->learned_hint = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((last - first + h->hint_n - 1) / h->hint_n, 4096)));
                    h->hint_pending = false;
                }
                hint = h->learned_hint;
                if (!h->hint_pending) {
                    h->hint_probe[0] = h->hint_probe[1] = 0;
                    GX_HIP(hipMemcpyAsync(&h->hint_probe[0], offsets, off_w, hipMemcpyDeviceToHost, stream));
                    GX_HIP(hipMemcpyAsync(&h->hint_probe[1], static_cast<const uint8_t*>(offsets) + n * off_w, off_w, hipMemcpyDeviceToHost, stream));
                    GX_HIP(hipEventRecord(h->hint_event, stream));
                    h->hint_pending = true;
                    h->hint_n = n;
                    h->hint_off64 = o.offsets64 != 0;
                }
            }
            if (hint == 0 && n && !o.no_sync) {
                // no hint: the mean line length, from the two ends of the offsets array (a small synchronous read;
                // asynchronous callers pass line_bytes_hint themselves)
                uint64_t first = 0, last = 0;
                GX_HIP(hipMemcpyAsync(&first, offsets, off_w, hipMemcpyDeviceToHost, stream));
                GX_HIP(hipMemcpyAsync(&last, static_cast<const uint8_t*>(offsets) + n * off_w, off_w, hipMemcpyDeviceToHost, stream));
                GX_HIP(hipStreamSynchronize(stream));
                hint = static_cast<uint32_t>(std::min<uint64_t>((last - first + n - 1) / n, 4096));
                if (hint == 0) hint = 1;
                if (o.uneven_lines == 0 && n >= 128) {
                    // ... and whether the lines differ much in length: the first 4096 of them
                    const uint64_t m = std::min<uint64_t>(n, 4096);
                    std::vector<uint8_t> sample((m + 1) * off_w);
                    GX_HIP(hipMemcpyAsync(sample.data(), offsets, sample.size(), hipMemcpyDeviceToHost, stream));
                    GX_HIP(hipStreamSynchronize(stream));
                    uneven = lines_are_uneven(sample.data(), m, o.offsets64 != 0);
                }
            }
            b.max_line_bytes = o.max_line_bytes;
            b.caller_no_sync = o.no_sync ? 1u : 0u;
            Launched done;
            launch_batch(h, b, hint, o.kernel, stream, uneven, &done);
            if (!o.no_sync) {
                GX_HIP(hipStreamSynchronize(stream));
                if (done.promised && __atomic_load_n(&h->h_broken[done.slot], __ATOMIC_RELAXED) == done.seq) {
                    // the promise did not hold: the lines the batch kernel left, now (and the word is clean for the stream's next launch)
                    {
                        std::lock_guard<std::mutex> lock(h->slot_mu);
                        if (h->promise_seq[done.slot] == done.seq) h->promise_seq[done.slot] = 0;
                        h->broken_seen[done.slot] = done.seq;   // (seen, and put right below)
                    }
                    h->promises_broken.fetch_add(1);
                    b.seq = done.seq;
                    b.oversize_flag = h->d_broken + done.slot;
                    PikeGate pike_gate(h, stream);
                    GX_HIP(launch_extract_oversize(h->dev, b, done.limit, done.by_length, stream));
                    GX_HIP(hipStreamSynchronize(stream));
                }
            }
            return GX_OK;
        }
        // host pointers: the chunked pipeline (gx_handle::host_slot)
        std::lock_guard<std::mutex> lock(h->mu);
        uint64_t total = 0;  // code units in the batch (offsets need not start at 0: a shard of a larger CSR buffer)
        if (n) total = o.offsets64 ? static_cast<const uint64_t*>(offsets)[n] - static_cast<const uint64_t*>(offsets)[0]
                                   : static_cast<const uint32_t*>(offsets)[n] - static_cast<const uint32_t*>(offsets)[0];
        uint32_t hint = o.line_bytes_hint;
        if (hint == 0 && n) hint = static_cast<uint32_t>(std::min<uint64_t>((total + n - 1) / n, 1u << 20));
        uint64_t over_total = 0;
        const bool uneven = o.uneven_lines == 2 || (o.uneven_lines == 0 && lines_are_uneven(offsets, n, o.offsets64 != 0));
        host_pipeline(h, b, bytes, offsets, n, total, match_id, caps, states, compact, match_only, hint, o.kernel, uneven, &over_total);
        if (compact && o.overflow) *static_cast<uint64_t*>(o.overflow) += over_total;
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
}

int gx_extract_batch(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, int32_t* match_id, int32_t* caps,
                     const gx_batch_opts* opts) {
    return extract_batch_impl(h, bytes, offsets, n, match_id, caps, nullptr, opts);
}

int gx_match_batch(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, int32_t* first_match, int32_t* states,
                   const gx_batch_opts* opts) {
    if (!states) return fail(GX_E_ARG, "gx_match_batch: states is NULL (gx_extract_batch with match_only gives the first match alone)");
    return extract_batch_impl(h, bytes, offsets, n, first_match, nullptr, states, opts);
}

int gx_set_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return fail(GX_E_DEVICE, "gx_set_device: no such device");
    if (hipSetDevice(device) != hipSuccess) return fail(GX_E_DEVICE, "gx_set_device: hipSetDevice failed");
    return GX_OK;
}

int gx_handle_device(const gx_handle* h) { return h && h->on_device ? h->device : -1; }

int gx_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail(GX_E_ARG, "gx_host_register: bad argument");
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(GX_E_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e));
    return GX_OK;
}
int gx_host_unregister(void* p) {
    if (!p) return fail(GX_E_ARG, "gx_host_unregister: bad argument");
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) return fail(GX_E_DEVICE, std::string("hipHostUnregister: ") + hipGetErrorString(e));
    return GX_OK;
}

// One CSR batch in host memory over several devices: lines are independent (core/Gorp.java:159-186 keeps no cross-line
// state), so the batch is cut into contiguous shards of about equal BYTES, one per handle (each on its own GPU, built
// from the same definition or blob), and every shard runs through its handle's host pipeline on its own thread.
int gx_extract_batch_multi(gx_handle* const* handles, int32_t n_handles, const uint8_t* bytes, const void* offsets, uint64_t n,
                           int32_t* match_id, int32_t* caps, const gx_batch_opts* opts) {
    if (!handles || n_handles <= 0 || !offsets) return fail(GX_E_ARG, "gx_extract_batch_multi: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    if (o.device_pointers) return fail(GX_E_ARG, "gx_extract_batch_multi: host buffers only (device buffers belong to one device: use gx_extract_batch per handle)");
    for (int32_t k = 0; k < n_handles; ++k) {
        if (!handles[k] || !handles[k]->on_device) return fail(GX_E_ARG, "gx_extract_batch_multi: NULL or host-only handle");
        if (handles[k]->T.max_groups != handles[0]->T.max_groups || handles[k]->T.n_rules != handles[0]->T.n_rules)
            return fail(GX_E_ARG, "gx_extract_batch_multi: the handles were not built from the same definition");
    }
    const size_t off_w = o.offsets64 ? 8 : 4;
    auto off_at = [&](uint64_t i) -> uint64_t {
        return o.offsets64 ? static_cast<const uint64_t*>(offsets)[i] : static_cast<const uint32_t*>(offsets)[i];
    };
    // shard boundaries by bytes
    std::vector<uint64_t> cuts(static_cast<size_t>(n_handles) + 1, n);
    cuts[0] = 0;
    const uint64_t base = n ? off_at(0) : 0, total = n ? off_at(n) - base : 0;
    for (int32_t k = 1; k < n_handles; ++k) {
        const uint64_t want = base + total / static_cast<uint64_t>(n_handles) * static_cast<uint64_t>(k);
        uint64_t lo = cuts[k - 1], hi = n;
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (off_at(mid) >= want) hi = mid; else lo = mid + 1; }
        cuts[k] = lo;
    }
    const bool compact = o.compact_results && !(o.match_only || !handles[0]->T.has_capture);
    const size_t slots = 2 * static_cast<size_t>(handles[0]->T.max_groups);
    std::vector<int> rc(static_cast<size_t>(n_handles), GX_OK);
    std::vector<std::string> msg(static_cast<size_t>(n_handles));
    std::vector<uint64_t> over(static_cast<size_t>(n_handles), 0);
    std::vector<std::thread> pool;
    for (int32_t k = 0; k < n_handles; ++k) {
        pool.emplace_back([&, k]() {
            const uint64_t a = cuts[k], m = cuts[k + 1] - cuts[k];
            if (m == 0) return;
            gx_batch_opts ok = o;
            ok.struct_size = sizeof(gx_batch_opts);
            ok.stream = nullptr;
            ok.overflow = compact ? &over[k] : nullptr;
            int32_t* mid_k = match_id ? match_id + a : nullptr;
            const size_t row_bytes = (1 + slots) * (o.compact_results == 2 ? 1 : 2);
            int32_t* caps_k = !caps ? nullptr : compact ? reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(caps) + a * row_bytes) : caps + a * slots;
            rc[k] = gx_extract_batch(handles[k], bytes, static_cast<const uint8_t*>(offsets) + a * off_w, m, mid_k, caps_k, &ok);
            if (rc[k] != GX_OK) msg[k] = gx_last_error();
        });
    }
    for (auto& t : pool) t.join();
    for (int32_t k = 0; k < n_handles; ++k) if (rc[k] != GX_OK) return fail(rc[k], msg[k]);
    if (compact && o.overflow) for (uint64_t v : over) *static_cast<uint64_t*>(o.overflow) += v;
    return GX_OK;
}

int gx_extract_batch_multi_device(const gx_device_shard* shards, int32_t n_shards, const gx_batch_opts* opts) {
    if (!shards || n_shards <= 0) return fail(GX_E_ARG, "gx_extract_batch_multi_device: bad argument");
    gx_batch_opts o{};
    if (!read_opts(opts, &o)) return fail(GX_E_ARG, "gx_batch_opts.struct_size mismatch");
    for (int32_t k = 0; k < n_shards; ++k)
        if (!shards[k].handle || !shards[k].handle->on_device) return fail(GX_E_ARG, "gx_extract_batch_multi_device: NULL or host-only handle");
    int prev_device = 0;
    (void)hipGetDevice(&prev_device);
    int first_rc = GX_OK;
    std::string first_msg;
    std::vector<hipStream_t> used(static_cast<size_t>(n_shards), nullptr);
    // enqueue everything first (asynchronous launches from this one thread), wait afterwards
    for (int32_t k = 0; k < n_shards; ++k) {
        const gx_device_shard& sh = shards[k];
        gx_handle* h = sh.handle;
        hipStream_t stream = static_cast<hipStream_t>(sh.stream);
        if (!stream) {
            std::lock_guard<std::mutex> lock(h->slot_mu);
            if (!h->multi_stream) {
                if (hipSetDevice(h->device) != hipSuccess || hipStreamCreateWithFlags(&h->multi_stream, hipStreamNonBlocking) != hipSuccess) {
                    if (first_rc == GX_OK) { first_rc = GX_E_DEVICE; first_msg = "gx_extract_batch_multi_device: no stream on the shard's device"; }
                    continue;
                }
            }
            stream = h->multi_stream;
        }
        used[k] = stream;
        gx_batch_opts ok = o;
        ok.struct_size = sizeof(gx_batch_opts);
        ok.device_pointers = 1;
        ok.no_sync = 1;
        ok.stream = stream;
        ok.overflow = sh.overflow;
        const int rc = sh.n ? gx_extract_batch(h, sh.bytes, sh.offsets, sh.n, sh.match_id, sh.caps, &ok) : GX_OK;
        if (rc != GX_OK && first_rc == GX_OK) { first_rc = rc; first_msg = gx_last_error(); }
    }
    if (!o.no_sync) {
        for (int32_t k = 0; k < n_shards; ++k) {
            if (!used[k]) continue;
            if (hipSetDevice(shards[k].handle->device) != hipSuccess || hipStreamSynchronize(used[k]) != hipSuccess) {
                if (first_rc == GX_OK) { first_rc = GX_E_DEVICE; first_msg = "gx_extract_batch_multi_device: a shard's stream failed"; }
                continue;
            }
            // a shard whose max_line_bytes promise did not hold (its kernel left the longer lines' rows unwritten): the shard again,
            // without the promise -- this call waits for its batches, so its results are right when it returns
            if (o.max_line_bytes != 0 && shards[k].n && promise_broken_since(shards[k].handle, used[k])) {
                gx_batch_opts ok = o;
                ok.struct_size = sizeof(gx_batch_opts);
                ok.device_pointers = 1;
                ok.no_sync = 0;
                ok.max_line_bytes = 0;
                ok.stream = used[k];
                ok.overflow = nullptr;   // (the first run has counted)
                const int rc = gx_extract_batch(shards[k].handle, shards[k].bytes, shards[k].offsets, shards[k].n, shards[k].match_id, shards[k].caps, &ok);
                if (rc != GX_OK && first_rc == GX_OK) { first_rc = rc; first_msg = gx_last_error(); }
            }
        }
    }
    (void)hipSetDevice(prev_device);
    if (first_rc != GX_OK) return fail(first_rc, first_msg);
    return GX_OK;
}

// ---- one process, all GPUs of a node: the tables on every device, the rows back on one (north_star: "broadcast of the DFA tables
// and a final gather over xGMI" -- for the caller that is ONE process, core/Gorp.java:22; ranks of a job use RCCL: gorp_amd/dist.py) ----
int gx_create_on_devices(const void* blob, size_t size, const int32_t* devices, int32_t n_devices, uint32_t flags, gx_handle** handles) {
    if (!blob || !devices || !handles || n_devices <= 0) return fail(GX_E_ARG, "gx_create_on_devices: bad argument");
    if (flags & GX_CREATE_HOST_ONLY) return fail(GX_E_ARG, "gx_create_on_devices: host-only handles live on no device");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(GX_E_DEVICE, "no HIP device available (libgorp_hip needs a gfx950 GPU; there is no CPU fallback)");
    for (int32_t k = 0; k < n_devices; ++k) {
        handles[k] = nullptr;
        if (devices[k] < 0 || devices[k] >= count) return fail(GX_E_ARG, "gx_create_on_devices: no such device");
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    // the first handle from the blob (host work + upload over the bus), the others beside it: their host-side tables in threads of
    // their own, their device images copied from the first handle's device
    int rc = GX_OK;
    std::string msg;
    if (hipSetDevice(devices[0]) != hipSuccess) rc = GX_E_DEVICE, msg = "gx_create_on_devices: hipSetDevice failed";
    if (rc == GX_OK) {
        rc = gx_create_from_blob(blob, size, flags, &handles[0]);
        if (rc != GX_OK) msg = gx_last_error();
    }
    if (rc == GX_OK && n_devices > 1) {
        std::vector<int> rcs(static_cast<size_t>(n_devices), GX_OK);
        std::vector<std::string> msgs(static_cast<size_t>(n_devices));
        std::vector<std::thread> th;
        const gx_handle* first = handles[0];
        for (int32_t k = 1; k < n_devices; ++k)
            th.emplace_back([&, k]() {
                if (hipSetDevice(devices[k]) != hipSuccess) { rcs[k] = GX_E_DEVICE; msgs[k] = "gx_create_on_devices: hipSetDevice failed"; return; }
                int can = 0;   // (peers read each other's memory directly once this is on; failing that the runtime stages the copy)
                if (devices[k] != first->device && hipDeviceCanAccessPeer(&can, devices[k], first->device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(first->device, 0);
                (void)hipGetLastError();
                g_peer_src = first;
                rcs[k] = gx_create_from_blob(blob, size, flags, &handles[k]);
                g_peer_src = nullptr;
                if (rcs[k] != GX_OK) msgs[k] = gx_last_error();
            });
        for (auto& t : th) t.join();
        for (int32_t k = 1; k < n_devices && rc == GX_OK; ++k)
            if (rcs[k] != GX_OK) { rc = rcs[k]; msg = msgs[k]; }
    }
    (void)hipSetDevice(prev);
    if (rc != GX_OK) {
        for (int32_t k = 0; k < n_devices; ++k) { if (handles[k]) gx_destroy(handles[k]); handles[k] = nullptr; }
        return fail(rc, msg);
    }
    return GX_OK;
}

int gx_gather_rows(const gx_rows_shard* shards, int32_t n_shards, uint32_t row_bytes, int32_t dst_device, void* dst_rows, int32_t no_sync) {
    if (!shards || n_shards <= 0 || row_bytes == 0 || !dst_rows) return fail(GX_E_ARG, "gx_gather_rows: bad argument");
    for (int32_t k = 0; k < n_shards; ++k)
        if (!shards[k].handle || !shards[k].handle->on_device || (shards[k].n && !shards[k].rows)) return fail(GX_E_ARG, "gx_gather_rows: NULL or host-only handle, or no rows");
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = GX_OK;
    std::string msg;
    auto bad = [&](const char* what) { if (rc == GX_OK) { rc = GX_E_DEVICE; msg = what; } };
    uint64_t at = 0;
    for (int32_t k = 0; k < n_shards && rc == GX_OK; ++k) {
        gx_handle* h = shards[k].handle;
        const uint64_t bytes = shards[k].n * static_cast<uint64_t>(row_bytes);
        uint8_t* dst = static_cast<uint8_t*>(dst_rows) + at;
        at += bytes;
        if (bytes == 0) continue;
        if (hipSetDevice(h->device) != hipSuccess) { bad("gx_gather_rows: hipSetDevice failed"); break; }
        std::lock_guard<std::mutex> lock(h->slot_mu);
        if (!h->gather_stream) {
            if (hipStreamCreateWithFlags(&h->gather_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&h->gather_event, hipEventDisableTiming) != hipSuccess) { bad("gx_gather_rows: no stream on the shard's device"); break; }
            int can = 0;   // the shard's device writes into the root's memory itself: one link per peer, all of them at once
            if (h->device != dst_device && hipDeviceCanAccessPeer(&can, h->device, dst_device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(dst_device, 0);
            (void)hipGetLastError();
        }
        // behind the shard's kernel (the stream it was enqueued on: the caller's, or the one gx_extract_batch_multi_device used) ...
        hipStream_t ks = static_cast<hipStream_t>(shards[k].stream);
        if (!ks) ks = h->multi_stream;
        if (ks) {
            if (hipEventRecord(h->gather_event, ks) != hipSuccess || hipStreamWaitEvent(h->gather_stream, h->gather_event, 0) != hipSuccess) { bad("gx_gather_rows: event"); break; }
        }
        // ... on the copy stream of the shard's own device: the kernels of the next batch go on beside it
        const hipError_t e = h->device == dst_device ? hipMemcpyAsync(dst, shards[k].rows, bytes, hipMemcpyDeviceToDevice, h->gather_stream)
                                                     : hipMemcpyPeerAsync(dst, dst_device, shards[k].rows, h->device, bytes, h->gather_stream);
        if (e != hipSuccess) bad("gx_gather_rows: the copy between the devices failed");
    }
    if (rc == GX_OK && !no_sync) {
        for (int32_t k = 0; k < n_shards; ++k) {
            gx_handle* h = shards[k].handle;
            if (!h->gather_stream) continue;
            if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->gather_stream) != hipSuccess) bad("gx_gather_rows: a copy stream failed");
        }
    }
    (void)hipSetDevice(prev);
    if (rc != GX_OK) return fail(rc, msg);
    return GX_OK;
}

int gx_gather_wait(gx_handle* const* handles, int32_t n_handles) {
    if (!handles || n_handles <= 0) return fail(GX_E_ARG, "gx_gather_wait: bad argument");
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = GX_OK;
    for (int32_t k = 0; k < n_handles; ++k) {
        gx_handle* h = handles[k];
        if (!h || !h->on_device) { rc = GX_E_ARG; continue; }
        if (!h->gather_stream) continue;
        if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->gather_stream) != hipSuccess) rc = GX_E_DEVICE;
    }
    (void)hipSetDevice(prev);
    if (rc != GX_OK) return fail(rc, "gx_gather_wait: a handle without a device, or a copy stream that failed");
    return GX_OK;
}

int gx_state_accepts(const gx_handle* h, int32_t state, int32_t* indexes, int32_t cap) {
    if (!h || state >= h->T.m_states || (cap > 0 && !indexes)) return -GX_E_ARG;
    if (state < 0) return 0;
    const uint32_t b = h->T.m_accept_off[state], e = h->T.m_accept_off[state + 1];
    for (uint32_t i = b; i < e && static_cast<int32_t>(i - b) < cap; ++i) indexes[i - b] = h->T.m_accept_list[i];
    return static_cast<int>(e - b);
}

// mode: 0 = extract, 1 = match only, -(k + 1) = extraction k's capture regexp alone.
// One String per call is latency, not throughput: the handle keeps a device scratch buffer and a pinned host mirror
// of it ([offsets 8 B][code units][match id, state, captures]), so a call is one copy in, one launch, one copy out.
static int one_line(gx_handle* h, const uint16_t* s, int32_t len, int32_t* match_id, int32_t* caps, int32_t* state, int mode) {
    if (!h || len < 0 || (len && !s)) return fail(GX_E_ARG, "bad argument");
    if (!h->on_device) return fail(GX_E_DEVICE, "handle was created host-only; no device tables (there is no CPU fallback)");
    try {
        GX_HIP(hipSetDevice(h->device));
        std::lock_guard<std::mutex> lock(h->mu);
        const size_t slots = 2 * static_cast<size_t>(h->T.max_groups);
        const size_t out_words = 2 + slots;
        const size_t in_bytes = 8 + ((static_cast<size_t>(len) * 2 + 7) & ~size_t(7));  // offsets + code units: one copy in
        const size_t need = in_bytes + out_words * 4 + 16;
        if (need > h->one_cap) {
            if (h->one_dev) (void)hipFree(h->one_dev);
            if (h->one_host) (void)hipHostFree(h->one_host);
            h->one_dev = nullptr; h->one_host = nullptr; h->one_cap = 0;
            const size_t cap = std::max<size_t>(need * 2, 4096);
            GX_HIP(hipMalloc(&h->one_dev, cap));
            GX_HIP(hipHostMalloc(&h->one_host, cap, hipHostMallocDefault));
            h->one_cap = cap;
        }
        uint8_t* hb = static_cast<uint8_t*>(h->one_host);
        uint8_t* db = static_cast<uint8_t*>(h->one_dev);
        uint32_t* offs = reinterpret_cast<uint32_t*>(hb);
        offs[0] = 0; offs[1] = static_cast<uint32_t>(len);
        if (len) memcpy(hb + 8, s, static_cast<size_t>(len) * 2);
        bool latin1 = mode == 0 && len > 0 && len <= 4096 && h->tile_ok;
        for (int32_t q = 0; latin1 && q < len; ++q) latin1 = s[q] <= 0xFFu;
        if (latin1 && h->svc.enabled && static_cast<uint32_t>(len) <= GX_SERVICE_MAX_BYTES) {
            // the resident wave (gx_service.hip): the line into the mailbox -- every cache line's text before its tag, the first cache
            // line, whose tag is what the wave polls, last -- and a spin on the answer's sequence number
            gx_handle::Service& sv = h->svc;
            uint32_t* mb = sv.host;
            int32_t* ans = reinterpret_cast<int32_t*>(sv.host + 17 * 16);
            uint32_t* state = sv.host + 17 * 16 + 80;
            const uint32_t seq = ++sv.seq ? sv.seq : ++sv.seq;   // (never 0... the wave compares for inequality only, but keep it tidy)
            const uint32_t ulen = static_cast<uint32_t>(len);
            for (uint32_t cl = 1; 56u + 60u * (cl - 1u) < ulen; ++cl) {
                uint8_t* dst = reinterpret_cast<uint8_t*>(mb + 16u * cl) + 4;
                const uint32_t from = 56u + 60u * (cl - 1u), cnt = std::min(60u, ulen - from);
                for (uint32_t q = 0; q < cnt; ++q) dst[q] = static_cast<uint8_t>(s[from + q]);
                __atomic_store_n(mb + 16u * cl, seq, __ATOMIC_RELEASE);
            }
            {
                uint8_t* dst = reinterpret_cast<uint8_t*>(mb) + 8;
                for (uint32_t q = 0; q < std::min(56u, ulen); ++q) dst[q] = static_cast<uint8_t>(s[q]);
                mb[1] = ulen;   // (flags 0)
                __atomic_store_n(mb, seq, __ATOMIC_RELEASE);
            }
            auto start_wave = [&]() {
                __atomic_store_n(state, 1u, __ATOMIC_RELEASE);
                // the wave starts with the PREVIOUS sequence number as the last one it has seen: the request that is waiting is new to it
                GX_HIP(launch_one_service(sv.mode, sv.L, static_cast<const uint8_t*>(h->d_lds_image), sv.dev, reinterpret_cast<int32_t*>(sv.dev + 17 * 16),
                                          sv.dev + 17 * 16 + 80, seq - 1u, h->T.max_groups, 30000ull, 2000000ull, sv.stream));
                sv.started = true;
                ++sv.launches;
            };
            if (!sv.started) start_wave();
            const uint32_t* ans_seq = reinterpret_cast<const uint32_t*>(ans) + 1 + 2 * h->T.max_groups;
            uint64_t spins = 0;
            while (__atomic_load_n(ans_seq, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 63u) == 0 && __atomic_load_n(state, __ATOMIC_ACQUIRE) == 2u && hipStreamQuery(sv.stream) == hipSuccess) {
                    // the wave has left (idle, or its time was up) -- without this request's answer: a fresh one
                    if (__atomic_load_n(ans_seq, __ATOMIC_ACQUIRE) == seq) break;
                    start_wave();
                }
                if (spins > (1ull << 34)) throw GxError(GX_E_DEVICE, "the resident one-line service does not answer");
            }
            if (match_id) *match_id = ans[0];
            if (caps) for (size_t t = 0; t < slots; ++t) caps[t] = h->T.has_capture ? ans[1 + t] : -1;
            return GX_OK;
        }
        if (latin1) {
            // Gorp.extract(String) on a Latin-1 line -- nearly every call: the line's BYTES go through the batch kernels as a batch of
            // one (tables in LDS, the line staged there too: a step costs an LDS round trip, not two trips to L2 as in the per-line
            // kernel), straight out of the pinned buffer and back into it; the host knows the line fits: no follow-up launch
            uint8_t* bytes = hb + 8;
            for (int32_t q = 0; q < len; ++q) bytes[q] = static_cast<uint8_t>(s[q]);   // (in place of the units copied above)
            GxBatch b{};
            b.data = bytes; b.offsets = hb; b.n = 1; b.wide = 0; b.offsets64 = 0;
            b.match_only = h->T.has_capture ? 0 : 1;
            int32_t* out = reinterpret_cast<int32_t*>(hb + in_bytes);
            b.match_id = out; b.caps = h->T.has_capture ? out + 2 : nullptr;
            b.no_followup = 1;
            launch_batch(h, b, static_cast<uint32_t>(len), GX_KERNEL_AUTO, nullptr);
            GX_HIP(hipStreamSynchronize(nullptr));
            if (match_id) *match_id = out[0];
            if (caps) for (size_t t = 0; t < slots; ++t) caps[t] = h->T.has_capture ? out[2 + t] : -1;
            return GX_OK;
        }
        if (len <= 16384) {
            // the short way: the kernel reads the units out of the pinned buffer and writes the result words into it (no copy commands)
            GxBatch b{};
            b.n = 1; b.wide = 1;
            b.match_only = mode < 0 ? mode : ((mode == 1 || !h->T.has_capture) ? 1 : 0);
            int32_t* out = reinterpret_cast<int32_t*>(hb + in_bytes);
            b.match_id = out; b.state_out = out + 1; b.caps = b.match_only == 1 ? nullptr : out + 2;
            PikeGate pike_gate(h, nullptr);
            GX_HIP(launch_extract_one(h->dev, reinterpret_cast<const uint16_t*>(hb + 8), static_cast<uint32_t>(len), b, nullptr));
            GX_HIP(hipStreamSynchronize(nullptr));
            const int32_t* host = out;
            if (match_id) *match_id = host[0];
            if (state) *state = host[1];
            if (caps) for (size_t t = 0; t < slots; ++t) caps[t] = b.match_only == 1 ? -1 : host[2 + t];
            return GX_OK;
        }
        // (the results area is not initialised: the kernel writes every word that is read back)
        GX_HIP(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, nullptr));
        GxBatch b{};
        b.data = db + 8; b.offsets = db; b.n = 1; b.wide = 1; b.offsets64 = 0;
        b.match_only = mode < 0 ? mode : ((mode == 1 || !h->T.has_capture) ? 1 : 0);
        int32_t* out = reinterpret_cast<int32_t*>(db + in_bytes);
        b.match_id = out; b.state_out = out + 1; b.caps = b.match_only == 1 ? nullptr : out + 2;
        PikeGate pike_gate(h, nullptr);
        GX_HIP(launch_extract_generic(h->dev, b, nullptr));
        const size_t back = (b.match_only == 1 ? 2 : out_words) * 4;
        GX_HIP(hipMemcpyAsync(hb + in_bytes, db + in_bytes, back, hipMemcpyDeviceToHost, nullptr));
        GX_HIP(hipStreamSynchronize(nullptr));
        const int32_t* host = reinterpret_cast<const int32_t*>(hb + in_bytes);
        if (match_id) *match_id = host[0];
        if (state) *state = host[1];
        if (caps) for (size_t t = 0; t < slots; ++t) caps[t] = b.match_only == 1 ? -1 : host[2 + t];
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
}

int gx_capture_one_utf16(gx_handle* h, int32_t k, const uint16_t* s, int32_t len, int32_t* matched, int32_t* caps) {
    if (!h || k < 0 || k >= h->T.n_rules || !matched) return fail(GX_E_ARG, "gx_capture_one_utf16: bad argument");
    if (!h->T.has_capture) {  // no extraction of this definition has groups: a plain regexp match (the generic kernel needs capture tables)
        return fail(GX_E_ARG, "gx_capture_one_utf16: the definition has no capture groups");
    }
    int32_t mid = -1;
    const int rc = one_line(h, s, len, &mid, caps, nullptr, -(k + 1));
    if (rc != GX_OK) return rc;
    *matched = mid == k ? 1 : 0;
    return GX_OK;
}

int gx_extract_one_utf16(gx_handle* h, const uint16_t* s, int32_t len, int32_t* match_id, int32_t* caps) {
    if (!match_id) return fail(GX_E_ARG, "gx_extract_one_utf16: match_id is NULL");
    return one_line(h, s, len, match_id, caps, nullptr, 0);
}

int gx_match_one_utf16(gx_handle* h, const uint16_t* s, int32_t len, int32_t* indexes, int32_t cap) {
    int32_t mid = -1, state = -1;
    int rc = one_line(h, s, len, &mid, nullptr, &state, 1);
    if (rc != GX_OK) return -rc;
    if (state < 0) return 0;
    const uint32_t b = h->T.m_accept_off[state], e = h->T.m_accept_off[state + 1];
    for (uint32_t i = b; i < e && static_cast<int32_t>(i - b) < cap; ++i) indexes[i - b] = h->T.m_accept_list[i];
    return static_cast<int>(e - b);
}

static int string_result(const ustr& r, char* out, size_t cap, size_t* out_len) {
    std::string u = u16_to_utf8(r);
    if (out_len) *out_len = u.size();
    if (!out || cap < u.size() + 1) return fail(GX_E_ARG, "output buffer too small");
    memcpy(out, u.c_str(), u.size() + 1);
    return GX_OK;
}

int gx_quote_literal_as_regexp(const char* text, char* out, size_t cap, size_t* out_len) {
    if (!text) return fail(GX_E_ARG, "null text");
    try { return string_result(quote_literal_as_regexp(utf8_to_u16(text)), out, cap, out_len); }
    catch (GxError& e) { return fail(e.code, e.what()); }
}
int gx_massage_regexp_for_automaton(const char* pattern, char* out, size_t cap, size_t* out_len) {
    if (!pattern) return fail(GX_E_ARG, "null pattern");
    try { return string_result(massage_regexp_for_automaton(utf8_to_u16(pattern)), out, cap, out_len); }
    catch (GxError& e) { return fail(e.code, e.what()); }
}
int gx_massage_regexp_for_jdk(const char* pattern, char* out, size_t cap, size_t* out_len) {
    if (!pattern) return fail(GX_E_ARG, "null pattern");
    try { return string_result(massage_regexp_for_jdk(utf8_to_u16(pattern)), out, cap, out_len); }
    catch (GxError& e) { return fail(e.code, e.what()); }
}

int gx_create_from_definition(const char* definition_text, const char* source_ref, uint32_t flags, gx_handle** out) {
    if (!definition_text || !out) return fail(GX_E_ARG, "gx_create_from_definition: bad argument");
    try {
        std::vector<dsl::Extraction> xs = dsl::read_definition(definition_text, source_ref ? source_ref : "<input string>");
        std::vector<ustr> a, j;
        for (auto& x : xs) {
            std::string as, js;
            dsl::build_regex_strings(x, as, js);
            a.push_back(utf8_to_u16(as.c_str()));
            j.push_back(utf8_to_u16(js.c_str()));
        }
        std::unique_ptr<gx_handle> h(new gx_handle());
        try {
            h->T = compile_tables(a, &j);
        } catch (GxError& e) {
            // core/Gorp.java:84-90
            if (e.code == GX_E_DEVICE || e.code == GX_E_NOMEM) throw;
            throw GxError(e.code, std::string("(N/A): Internal error: problem with PolyMatcher construction: ") + e.what());
        }
        h->meta = xs;
        return finish_create(h, flags, out);
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

const char* gx_extraction_name(const gx_handle* h, int32_t k) {
    if (!h || k < 0 || k >= static_cast<int32_t>(h->meta.size())) return nullptr;
    return h->meta[k].name.c_str();
}
const char* gx_extractor_name(const gx_handle* h, int32_t k, int32_t g) {
    if (!h || k < 0 || k >= static_cast<int32_t>(h->meta.size())) return nullptr;
    if (g < 0 || g >= static_cast<int32_t>(h->meta[k].extractor_names.size())) return nullptr;
    return h->meta[k].extractor_names[g].c_str();
}
const char* gx_extraction_append_json(const gx_handle* h, int32_t k) {
    if (!h || k < 0 || k >= static_cast<int32_t>(h->meta.size()) || h->meta[k].append_json.empty()) return nullptr;
    return h->meta[k].append_json.c_str();
}

static const std::vector<std::pair<std::string, std::string>>* append_entries_of(const gx_handle* hc, int32_t k) {
    gx_handle* h = const_cast<gx_handle*>(hc);
    if (!h || k < 0 || k >= static_cast<int32_t>(h->meta.size())) return nullptr;
    std::lock_guard<std::mutex> lock(h->mu);
    if (h->append_entries.size() != h->meta.size()) {
        h->append_entries.assign(h->meta.size(), {});
        for (size_t x = 0; x < h->meta.size(); ++x)
            if (!h->meta[x].append_json.empty()) h->append_entries[x] = dsl::json_object_entries(h->meta[x].append_json);
    }
    return &h->append_entries[k];
}
int32_t gx_extraction_append_count(const gx_handle* h, int32_t k) {
    auto* e = append_entries_of(h, k);
    return e ? static_cast<int32_t>(e->size()) : 0;
}
const char* gx_extraction_append_key(const gx_handle* h, int32_t k, int32_t j) {
    auto* e = append_entries_of(h, k);
    return (e && j >= 0 && j < static_cast<int32_t>(e->size())) ? (*e)[j].first.c_str() : nullptr;
}
const char* gx_extraction_append_value_json(const gx_handle* h, int32_t k, int32_t j) {
    auto* e = append_entries_of(h, k);
    return (e && j >= 0 && j < static_cast<int32_t>(e->size())) ? (*e)[j].second.c_str() : nullptr;
}

int gx_definition_to_json(const char* definition_text, const char* source_ref, const char* stage, char* out, size_t cap,
                          size_t* out_len) {
    if (!definition_text || !stage) return fail(GX_E_ARG, "gx_definition_to_json: bad argument");
    try {
        std::string js = dsl::dump_json(definition_text, source_ref ? source_ref : "<input string>", stage);
        if (out_len) *out_len = js.size();
        if (!out || cap < js.size() + 1) return fail(GX_E_ARG, "output buffer too small");
        memcpy(out, js.c_str(), js.size() + 1);
        return GX_OK;
    } catch (GxError& e) { return fail(e.code, e.what()); }
    catch (std::bad_alloc&) { return fail(GX_E_NOMEM, "out of memory"); }
    catch (std::exception& e) { return fail(GX_E_ARG, e.what()); }
}

}  // extern "C"

#ifdef GX_DEV
// Developer build only (libgorp_hip_dev.so, `python -m gorp_amd.build --dev`): the tile kernel adds up the cycles
// each wave spends in its four phases (stage, prefetch issue, walk, results) into this device buffer,
// 8 x uint64 per wave of the grid (256 workgroups x 12 waves at most; gx_tile_body.hpp: TileIO::stamps).  Not part of the product ABI.
namespace gx { hipError_t jsonl_dev_phases(unsigned long long* out16, int reset); }
// cycles per phase of the JSONL tile kernels summed over their waves since the last reset: [0..5] sizes pass, [8..13] write pass
extern "C" int gx_dev_jsonl_phases(unsigned long long* out16, int reset) { return gx::jsonl_dev_phases(out16, reset) == hipSuccess ? GX_OK : GX_E_DEVICE; }
extern "C" int gx_dev_set_stamps(gx_handle* h, void* device_buffer) {
    if (!h) return GX_E_ARG;
    h->dev_stamps = static_cast<unsigned long long*>(device_buffer);
    return GX_OK;
}
#endif
