// gx_compile.cpp -- regex strings -> (match DFA, per-extraction tagged DFAs).
//
// Pipeline (all over one shared partition of the UTF-16 alphabet into classes):
//   AST -> prioritised Thompson program -> per-extraction DFA -> Hopcroft
//   -> product over extractions (what Automata.construct does,
//      core/autom/Automata.java:57-124) -> Hopcroft again on the product
//   AST(JDK dialect) -> prioritised program with tags -> tagged DFA per extraction
//      (replaces the backtracking Matcher of
//       core/jdkre/JDKRegexpCookedExtraction.java:36-59 with a single forward pass)
#include "gx_compile.hpp"

#include <map>
#include <unordered_map>

namespace gx {
namespace {

// ---------------------------------------------------------------------------
// Prioritised Thompson program
// ---------------------------------------------------------------------------
struct Inst {
    enum Op : uint8_t { CHAR, SPLIT, JMP, TAG, MATCH, FAIL } op;
    int x = 0, y = 0;  // CHAR: x = set id; SPLIT: x preferred over y; JMP: x; TAG: x = tag number
};

struct Prog {
    std::vector<Inst> code;
    int ngroups = 0;
};

const size_t PROG_LIMIT = 400000;

struct SetPool {
    std::map<std::vector<std::pair<int, int>>, int> index;
    std::vector<CharSet> sets;
    int intern(const CharSet& s) {
        auto it = index.find(s.iv);
        if (it != index.end()) return it->second;
        int id = static_cast<int>(sets.size());
        index.emplace(s.iv, id);
        sets.push_back(s);
        return id;
    }
};

class ProgBuilder {
public:
    ProgBuilder(Prog& p, SetPool& pool, bool tags) : p_(p), pool_(pool), tags_(tags) {}

    void build(const Ast* root) {
        gen(root);
        emit(Inst::MATCH);
    }

private:
    Prog& p_;
    SetPool& pool_;
    bool tags_;

    int emit(Inst::Op op, int x = 0, int y = 0) {
        if (p_.code.size() >= PROG_LIMIT) throw GxError(GX_E_LIMIT, "regex expands to too many instructions");
        Inst in;
        in.op = op; in.x = x; in.y = y;
        p_.code.push_back(in);
        return static_cast<int>(p_.code.size()) - 1;
    }
    int here() const { return static_cast<int>(p_.code.size()); }

    void split_to(int at, int body, int exit, bool greedy) {
        p_.code[at].x = greedy ? body : exit;
        p_.code[at].y = greedy ? exit : body;
    }

    void gen(const Ast* n) {
        switch (n->kind) {
        case Ast::EMPTY: return;
        case Ast::FAIL: emit(Inst::FAIL); return;
        case Ast::SET: emit(Inst::CHAR, pool_.intern(n->set)); return;
        case Ast::CAT: for (auto& k : n->kids) gen(k.get()); return;
        case Ast::GROUP:
            if (tags_ && n->cap > 0) emit(Inst::TAG, 2 * (n->cap - 1));
            gen(n->kids[0].get());
            if (tags_ && n->cap > 0) emit(Inst::TAG, 2 * (n->cap - 1) + 1);
            return;
        case Ast::ALT: {
            std::vector<int> exits;
            for (size_t i = 0; i < n->kids.size(); ++i) {
                if (i + 1 == n->kids.size()) { gen(n->kids[i].get()); break; }
                int sp = emit(Inst::SPLIT);
                p_.code[sp].x = here();
                gen(n->kids[i].get());
                exits.push_back(emit(Inst::JMP));
                p_.code[sp].y = here();
            }
            for (int j : exits) p_.code[j].x = here();
            return;
        }
        case Ast::REP: {
            const Ast* body = n->kids[0].get();
            for (int i = 0; i < n->min; ++i) gen(body);
            if (n->max < 0) {
                int sp = emit(Inst::SPLIT);
                gen(body);
                emit(Inst::JMP, sp);
                split_to(sp, sp + 1, here(), n->greedy);
            } else {
                std::vector<int> sps;
                for (int i = n->min; i < n->max; ++i) { sps.push_back(emit(Inst::SPLIT)); gen(body); }
                for (int sp : sps) split_to(sp, sp + 1, here(), n->greedy);
            }
            return;
        }
        }
    }
};

// ---------------------------------------------------------------------------
// Character classes: coarsest partition of [0,0xFFFF] that every set respects
// ---------------------------------------------------------------------------
struct Classes {
    int ncls = 0;
    std::vector<int> bounds;        // ascending interval starts
    std::vector<int> bound_cls;     // class of [bounds[i], bounds[i+1])
    std::vector<std::vector<uint64_t>> set_mask;  // per set id: bitmask over classes
    int words = 0;

    bool set_has(int set_id, int cls) const { return (set_mask[set_id][cls >> 6] >> (cls & 63)) & 1; }
};

Classes build_classes(const SetPool& pool) {
    Classes C;
    std::vector<int> b{0};
    for (auto& s : pool.sets)
        for (auto& r : s.iv) { b.push_back(r.first); if (r.second < 0xFFFF) b.push_back(r.second + 1); }
    std::sort(b.begin(), b.end());
    b.erase(std::unique(b.begin(), b.end()), b.end());
    C.bounds = b;
    const size_t nset = pool.sets.size();
    // membership signature of each atomic interval
    std::map<std::vector<uint64_t>, int> sig_index;
    const size_t sw = (nset + 63) / 64;
    std::vector<std::vector<uint64_t>> sigs(b.size(), std::vector<uint64_t>(sw, 0));
    for (size_t si = 0; si < nset; ++si) {
        for (auto& r : pool.sets[si].iv) {
            size_t i = std::lower_bound(b.begin(), b.end(), r.first) - b.begin();
            for (; i < b.size() && b[i] <= r.second; ++i) sigs[i][si >> 6] |= 1ull << (si & 63);
        }
    }
    C.bound_cls.resize(b.size());
    for (size_t i = 0; i < b.size(); ++i) {
        auto it = sig_index.find(sigs[i]);
        if (it == sig_index.end()) it = sig_index.emplace(sigs[i], static_cast<int>(sig_index.size())).first;
        C.bound_cls[i] = it->second;
    }
    C.ncls = static_cast<int>(sig_index.size());
    C.words = (C.ncls + 63) / 64;
    C.set_mask.assign(nset, std::vector<uint64_t>(C.words, 0));
    for (size_t i = 0; i < b.size(); ++i) {
        int c = C.bound_cls[i];
        for (size_t si = 0; si < nset; ++si)
            if ((sigs[i][si >> 6] >> (si & 63)) & 1) C.set_mask[si][c >> 6] |= 1ull << (c & 63);
    }
    return C;
}

// ---------------------------------------------------------------------------
// Plain DFA over classes (-1 = no transition), with an integer label per state
// ---------------------------------------------------------------------------
struct Dfa {
    int n = 0, ncls = 0;
    std::vector<int32_t> next;   // [n * ncls]
    std::vector<int32_t> label;  // -1 = not accepting
};

const size_t DFA_LIMIT = 2000000;

// epsilon-closure "core": the CHAR / MATCH pcs reachable without consuming input
void closure_core(const Prog& p, int pc, std::vector<char>& seen, std::vector<int>& out, std::vector<int>& stack) {
    stack.push_back(pc);
    while (!stack.empty()) {
        int q = stack.back(); stack.pop_back();
        if (seen[q]) continue;
        seen[q] = 1;
        const Inst& in = p.code[q];
        switch (in.op) {
        case Inst::CHAR: case Inst::MATCH: out.push_back(q); break;
        case Inst::SPLIT: stack.push_back(in.y); stack.push_back(in.x); break;
        case Inst::JMP: stack.push_back(in.x); break;
        case Inst::TAG: stack.push_back(q + 1); break;
        case Inst::FAIL: break;
        }
    }
}

Dfa subset_construct(const Prog& p, const Classes& C) {
    Dfa d;
    d.ncls = C.ncls;
    std::map<std::vector<int>, int> index;
    std::vector<std::vector<int>> states;
    std::vector<char> seen(p.code.size(), 0);
    std::vector<int> stack, core;
    auto close_from = [&](const std::vector<int>& seeds) {
        std::vector<int> out;
        std::fill(seen.begin(), seen.end(), 0);
        for (int s : seeds) closure_core(p, s, seen, out, stack);
        std::sort(out.begin(), out.end());
        return out;
    };
    states.push_back(close_from({0}));
    index[states[0]] = 0;
    for (size_t i = 0; i < states.size(); ++i) {
        if (states.size() > DFA_LIMIT) throw GxError(GX_E_LIMIT, "extraction automaton too large");
        const std::vector<int> cur = states[i];
        bool acc = false;
        for (int q : cur) if (p.code[q].op == Inst::MATCH) acc = true;
        d.label.push_back(acc ? 0 : -1);
        for (int c = 0; c < C.ncls; ++c) {
            std::vector<int> seeds;
            for (int q : cur) if (p.code[q].op == Inst::CHAR && C.set_has(p.code[q].x, c)) seeds.push_back(q + 1);
            if (seeds.empty()) { d.next.push_back(-1); continue; }
            std::vector<int> tgt = close_from(seeds);
            if (tgt.empty()) { d.next.push_back(-1); continue; }
            auto it = index.find(tgt);
            if (it == index.end()) { it = index.emplace(tgt, static_cast<int>(states.size())).first; states.push_back(tgt); }
            d.next.push_back(it->second);
        }
    }
    d.n = static_cast<int>(states.size());
    return d;
}

// Hopcroft partition refinement.  Input DFA may be partial; output is the
// minimal partial DFA with unreachable and dead states removed (state 0 stays
// the start state even when its language is empty).
Dfa minimize(const Dfa& in) {
    const int K = in.ncls;
    const int N = in.n + 1;  // + explicit sink
    const int SINK = in.n;
    auto nx = [&](int s, int c) -> int {
        if (s == SINK) return SINK;
        int t = in.next[static_cast<size_t>(s) * K + c];
        return t < 0 ? SINK : t;
    };
    // inverse transitions, CSR per class
    std::vector<std::vector<int>> inv_off(K, std::vector<int>(N + 1, 0));
    std::vector<std::vector<int>> inv(K);
    for (int c = 0; c < K; ++c) {
        auto& off = inv_off[c];
        for (int s = 0; s < N; ++s) off[nx(s, c) + 1]++;
        for (int t = 0; t < N; ++t) off[t + 1] += off[t];
        inv[c].assign(N, 0);
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int s = 0; s < N; ++s) inv[c][fill[nx(s, c)]++] = s;
    }
    // initial partition by label
    std::vector<int> perm(N), pos(N), blk(N);
    std::vector<int> bbeg, bend, bmark;
    {
        std::map<int, std::vector<int>> by_label;
        for (int s = 0; s < N; ++s) by_label[s == SINK ? -1 : in.label[s]].push_back(s);
        int at = 0;
        for (auto& kv : by_label) {
            int b = static_cast<int>(bbeg.size());
            bbeg.push_back(at);
            for (int s : kv.second) { perm[at] = s; pos[s] = at; blk[s] = b; ++at; }
            bend.push_back(at);
            bmark.push_back(0);
        }
    }
    std::vector<char> in_work(bbeg.size(), 1);
    std::vector<int> work;
    for (size_t b = 0; b < bbeg.size(); ++b) work.push_back(static_cast<int>(b));
    std::vector<int> touched, snapshot;
    while (!work.empty()) {
        int A = work.back(); work.pop_back();
        in_work[A] = 0;
        snapshot.assign(perm.begin() + bbeg[A], perm.begin() + bend[A]);
        for (int c = 0; c < K; ++c) {
            touched.clear();
            for (int t : snapshot) {
                for (int k = inv_off[c][t]; k < inv_off[c][t + 1]; ++k) {
                    int s = inv[c][k];
                    int b = blk[s];
                    int i = pos[s], j = bbeg[b] + bmark[b];
                    if (i < j) continue;  // already marked
                    if (bmark[b] == 0) touched.push_back(b);
                    int other = perm[j];
                    perm[i] = other; pos[other] = i;
                    perm[j] = s; pos[s] = j;
                    bmark[b]++;
                }
            }
            for (int b : touched) {
                int m = bmark[b], size = bend[b] - bbeg[b];
                bmark[b] = 0;
                if (m == size) continue;
                int nb = static_cast<int>(bbeg.size());
                if (m <= size - m) {  // marked front part becomes the new block
                    bbeg.push_back(bbeg[b]); bend.push_back(bbeg[b] + m);
                    bbeg[b] += m;
                } else {              // unmarked back part becomes the new block
                    bbeg.push_back(bbeg[b] + m); bend.push_back(bend[b]);
                    bend[b] = bbeg[b] + m;
                }
                bmark.push_back(0);
                for (int i = bbeg[nb]; i < bend[nb]; ++i) blk[perm[i]] = nb;
                in_work.push_back(1);
                work.push_back(nb);  // nb is the smaller half; if b is queued both halves are now queued
                (void)in_work;
            }
        }
    }
    // liveness on the quotient: blocks that can reach an accepting block
    const int B = static_cast<int>(bbeg.size());
    std::vector<std::vector<int>> rev(B);
    std::vector<int> rep(B);
    for (int b = 0; b < B; ++b) rep[b] = perm[bbeg[b]];
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < K; ++c) rev[blk[nx(rep[b], c)]].push_back(b);
    std::vector<char> live(B, 0);
    std::vector<int> st;
    for (int b = 0; b < B; ++b) if (rep[b] != SINK && blk[rep[b]] == b && in.label[rep[b]] >= 0) { live[b] = 1; st.push_back(b); }
    while (!st.empty()) {
        int b = st.back(); st.pop_back();
        for (int pb : rev[b]) if (!live[pb]) { live[pb] = 1; st.push_back(pb); }
    }
    // renumber reachable live blocks breadth-first from the start block
    Dfa out;
    out.ncls = K;
    std::vector<int> id(B, -1), order;
    id[blk[0]] = 0; order.push_back(blk[0]);
    for (size_t i = 0; i < order.size(); ++i) {
        int b = order[i];
        for (int c = 0; c < K; ++c) {
            int tb = blk[nx(rep[b], c)];
            if (!live[tb]) { out.next.push_back(-1); continue; }
            if (id[tb] < 0) { id[tb] = static_cast<int>(order.size()); order.push_back(tb); }
            out.next.push_back(id[tb]);
        }
        out.label.push_back(rep[b] == SINK ? -1 : in.label[rep[b]]);
    }
    out.n = static_cast<int>(order.size());
    return out;
}

// ---------------------------------------------------------------------------
// Product of the per-extraction DFAs; label = id of the accept set
// ---------------------------------------------------------------------------
struct PairVecHash {
    size_t operator()(const std::vector<std::pair<int, int>>& v) const {
        size_t h = 0xcbf29ce484222325ull;
        for (auto& e : v) { h = (h ^ (static_cast<size_t>(e.first) * 0x9E3779B97F4A7C15ull + e.second)) * 0x100000001b3ull; }
        return h;
    }
};

Dfa product(const std::vector<Dfa>& parts, int ncls, std::vector<std::vector<int32_t>>& accept_sets) {
    typedef std::vector<std::pair<int, int>> Tuple;
    Dfa d;
    d.ncls = ncls;
    std::unordered_map<Tuple, int, PairVecHash> index;
    std::vector<Tuple> queue;
    std::map<std::vector<int32_t>, int> set_index;
    Tuple init;
    for (size_t k = 0; k < parts.size(); ++k) init.push_back({static_cast<int>(k), 0});
    index.emplace(init, 0);
    queue.push_back(init);
    for (size_t head = 0; head < queue.size(); ++head) {
        if (queue.size() > DFA_LIMIT) throw GxError(GX_E_LIMIT, "product automaton too large");
        const Tuple cur = queue[head];
        std::vector<int32_t> acc;
        for (auto& e : cur) if (parts[e.first].label[e.second] >= 0) acc.push_back(e.first);
        if (acc.empty()) d.label.push_back(-1);
        else {
            auto it = set_index.find(acc);
            if (it == set_index.end()) { it = set_index.emplace(acc, static_cast<int>(accept_sets.size())).first; accept_sets.push_back(acc); }
            d.label.push_back(it->second);
        }
        for (int c = 0; c < ncls; ++c) {
            Tuple nxt;
            for (auto& e : cur) {
                int t = parts[e.first].next[static_cast<size_t>(e.second) * ncls + c];
                if (t >= 0) nxt.push_back({e.first, t});
            }
            if (nxt.empty()) { d.next.push_back(-1); continue; }
            auto it = index.find(nxt);
            if (it == index.end()) { it = index.emplace(nxt, static_cast<int>(queue.size())).first; queue.push_back(nxt); }
            d.next.push_back(it->second);
        }
    }
    d.n = static_cast<int>(queue.size());
    return d;
}

// ---------------------------------------------------------------------------
// Tagged DFA (one lookahead symbol, leftmost-greedy thread priority)
// ---------------------------------------------------------------------------
// A state is an ordered list of threads.  Thread order is backtracking
// priority: the first thread standing on MATCH when the input ends is the
// path java.util.regex's backtracker would have returned.  Each thread knows,
// per tag, which register holds the tag's value (or NIL), plus the set of tags
// it crossed since the last consumed symbol ("pending"); pending tags are
// written with the current position by the NEXT transition (or by the final
// step at end of input).
//
// Registers are allocated by VALUE: every tag written by one transition
// receives the same value (the current position), so all of them -- across
// threads and, in the union automaton, across extractions -- share ONE
// register.  A register stays live while any surviving thread refers to it.
// A transition therefore carries at most one "register := position" plus
// copies on state merges, and the register count tracks the number of distinct
// live positions rather than (extractions x tags).
//
// The builder accepts a program holding several extractions (rule k's code ends
// in MATCH k, entered through a priority chain of SPLITs): the resulting "union"
// automaton finds the first-declared extraction that matches the whole line AND
// its captures in a single pass.
const int MAX_TAGS = 64;
const int TDFA_STATE_LIMIT = 60000;
// One extraction whose capture automaton passes this many states is ambiguous in many independent places (n fields that may be empty
// between blanks: 2^n register patterns -- 12 such fields are 4 097 states and 22 s of building, 14 fields 16 385 states and minutes):
// it keeps its program and is run as one (RuleTables::pike).  Extractions of real definitions have tens to hundreds of states.
const int PROGRAM_INSTEAD_STATES = 4096;
const int16_t R_NIL = -1, R_NEW = -2;   // (while a step is built: R_NEW - k = "the current position", k = the tag's run, see tag_run_)

struct Thread {
    int pc;
    uint64_t pending;
    std::vector<int16_t> reg;  // per tag of the thread's rule: register id, R_NIL, or (while building) R_NEW - k
    bool operator==(const Thread& o) const { return pc == o.pc && pending == o.pending && reg == o.reg; }
    bool operator<(const Thread& o) const {
        if (pc != o.pc) return pc < o.pc;
        if (pending != o.pending) return pending < o.pending;
        return reg < o.reg;
    }
};
// A tagged-DFA state: the ordered thread list, plus (fused automaton only) the state of the match
// automaton that runs in product with it.
struct TState {
    int d = 0;
    std::vector<Thread> th;
    bool operator==(const TState& o) const { return d == o.d && th == o.th; }
    bool operator<(const TState& o) const { return d != o.d ? d < o.d : th < o.th; }
};

struct OpListPool {
    std::map<std::vector<uint16_t>, int> index;
    std::vector<std::vector<uint16_t>> lists;
    OpListPool() { lists.emplace_back(); index.emplace(lists[0], 0); }
    int intern(const std::vector<uint16_t>& l) {
        auto it = index.find(l);
        if (it != index.end()) return it->second;
        if (lists.size() >= 65535) throw GxError(GX_E_LIMIT, "too many distinct capture programs");
        int id = static_cast<int>(lists.size());
        index.emplace(l, id);
        lists.push_back(l);
        return id;
    }
};

// One or more extractions in one program.
struct MultiProg {
    std::vector<Inst> code;
    std::vector<int> rule_of_pc;  // -1 for the SPLIT prelude
    std::vector<int> ntags;       // per rule
    std::vector<int> ngroups;     // per rule
};

MultiProg join_programs(const std::vector<const Prog*>& progs) {
    MultiProg mp;
    const int n = static_cast<int>(progs.size());
    for (int k = 0; k + 1 < n; ++k) { Inst sp; sp.op = Inst::SPLIT; mp.code.push_back(sp); mp.rule_of_pc.push_back(-1); }
    std::vector<int> start(n);
    for (int k = 0; k < n; ++k) {
        const int base = static_cast<int>(mp.code.size());
        start[k] = base;
        for (Inst in : progs[k]->code) {
            if (in.op == Inst::SPLIT) { in.x += base; in.y += base; }
            else if (in.op == Inst::JMP) in.x += base;
            else if (in.op == Inst::MATCH) in.x = k;
            mp.code.push_back(in);
            mp.rule_of_pc.push_back(k);
        }
        mp.ntags.push_back(2 * progs[k]->ngroups);
        mp.ngroups.push_back(progs[k]->ngroups);
        if (mp.code.size() > PROG_LIMIT) throw GxError(GX_E_LIMIT, "combined program too large");
    }
    for (int k = 0; k + 1 < n; ++k) {
        mp.code[k].x = start[k];
        mp.code[k].y = (k + 2 < n) ? k + 1 : start[n - 1];
    }
    return mp;
}

class TdfaBuilder {
public:
    // with_rule_id: final-tag records start with the extraction index (union automaton)
    // fused != nullptr: run in product with the match automaton of *fused (see Tables::uni);
    // final-tag records then start with the extraction index.
    // state_limit: what the caller is ready to build ahead of time (one extraction alone: PROGRAM_INSTEAD_STATES -- beyond it the
    // extraction is run as a program; the fused automaton: TDFA_STATE_LIMIT -- beyond it the kernels take two passes)
    TdfaBuilder(const MultiProg& p, const Classes& C, OpListPool& ops, std::vector<uint16_t>& fin_tags, const Tables* fused, int state_limit = TDFA_STATE_LIMIT)
        : p_(p), C_(C), ops_(ops), fin_tags_(fin_tags), with_rule_id_(fused != nullptr), fused_(fused), state_limit_(state_limit) {
        for (int t : p.ntags) if (t > MAX_TAGS) throw GxError(GX_E_LIMIT, "more than 32 capture groups in one extraction");
        // Tags that are set in one step all receive the current position.  Which of them SHARE the register that receives it is a
        // matter of the automaton's shape only, and it is decided by the program, not by the line: tags whose TAG instructions
        // stand in one run ("((", ")(", "))": always set together) share one; any other pair (the two ends of a group that
        // may match the empty string: set together on one path, apart on another) gets a register each.  With one register for
        // all, "this group was empty" stayed visible in the state for ever -- as which registers coincide -- and n such
        // groups made 2^n states (fourteen fields ([^,]*), did not compile).
        tag_run_.resize(p.ntags.size());
        for (size_t k = 0; k < p.ntags.size(); ++k) tag_run_[k].assign(static_cast<size_t>(p.ntags[k]), -1);
        for (size_t pc = 0; pc < p.code.size(); ++pc) {
            if (p.code[pc].op != Inst::TAG) continue;
            const int rule = p.rule_of_pc[pc];
            size_t first = pc;
            while (first > 0 && p.code[first - 1].op == Inst::TAG && p.rule_of_pc[first - 1] == rule) --first;
            int& run = tag_run_[static_cast<size_t>(rule)][static_cast<size_t>(p.code[pc].x)];
            if (run < 0) run = static_cast<int>(first);   // (a tag with several TAG instructions: the first one's run)
        }
    }

    RuleTables build() {
        RuleTables R;
        for (int g : p_.ngroups) R.n_groups = std::max(R.n_groups, g);
        seen_.assign(p_.code.size(), 0);
        TState init;
        {
            Thread seed;
            seed.pc = 0; seed.pending = 0;
            std::vector<Thread> seeds{seed};
            init.th = close(seeds);
            init.d = 0;
        }
        add_state(init);
        for (size_t i = 0; i < states_.size(); ++i) {
            for (int c = 0; c < C_.ncls; ++c) {
                const TState cur = states_[i];  // copy: states_ may grow
                uint32_t w = step(cur, c);
                trans_.push_back(w);
            }
        }
        // append the absorbing dead state
        const int dead = static_cast<int>(states_.size());
        R.n_states = dead + 1;
        R.dead = dead;
        R.trans.resize(static_cast<size_t>(R.n_states) * C_.ncls);
        for (size_t i = 0; i < trans_.size(); ++i) {
            uint32_t w = trans_[i];
            R.trans[i] = (w == DEAD_MARK) ? static_cast<uint32_t>(dead) : w;
        }
        for (int c = 0; c < C_.ncls; ++c) R.trans[static_cast<size_t>(dead) * C_.ncls + c] = static_cast<uint32_t>(dead);
        R.fin.assign(R.n_states, -1);
        for (int s = 0; s < dead; ++s) R.fin[s] = final_of(states_[s]);
        R.n_regs = nregs_;
        return R;
    }

private:
    static const uint32_t DEAD_MARK = 0xFFFFFFFFu;
    const MultiProg& p_;
    const Classes& C_;
    OpListPool& ops_;
    std::vector<uint16_t>& fin_tags_;
    bool with_rule_id_;
    const Tables* fused_;
    int state_limit_;
    int nregs_ = 0;
    int scratch_ = -1;
    std::vector<TState> states_;
    std::map<TState, int> exact_;
    std::map<std::vector<std::pair<int, uint64_t>>, std::vector<int>> by_shape_;
    std::vector<uint32_t> trans_;
    std::vector<char> seen_;
    std::vector<std::vector<int>> tag_run_;   // per rule and tag: the pc that begins the run of TAG instructions the tag's stands in

    int ntags_at(int pc) const { return p_.ntags[p_.rule_of_pc[pc]]; }

    // Ordered epsilon-closure of a list of seed threads (highest priority first).
    std::vector<Thread> close(const std::vector<Thread>& seeds) {
        std::vector<Thread> out;
        std::fill(seen_.begin(), seen_.end(), 0);
        std::vector<std::pair<int, uint64_t>> stack;
        for (const Thread& seed : seeds) {
            stack.clear();
            stack.push_back({seed.pc, seed.pending});
            while (!stack.empty()) {
                auto top = stack.back(); stack.pop_back();
                int q = top.first;
                if (seen_[q]) continue;
                seen_[q] = 1;
                const Inst& in = p_.code[q];
                switch (in.op) {
                case Inst::CHAR: case Inst::MATCH: {
                    Thread t;
                    t.pc = q; t.pending = top.second; t.reg = seed.reg;
                    const int nt = ntags_at(q);
                    if (t.reg.empty()) t.reg.assign(nt, R_NIL);  // the initial seed enters its rule here
                    // a pending tag's old value is dead: the next step overwrites it
                    for (int g = 0; g < nt; ++g) if ((t.pending >> g) & 1) t.reg[g] = R_NIL;
                    out.push_back(std::move(t));
                    break;
                }
                case Inst::SPLIT: stack.push_back({in.y, top.second}); stack.push_back({in.x, top.second}); break;
                case Inst::JMP: stack.push_back({in.x, top.second}); break;
                case Inst::TAG: stack.push_back({q + 1, top.second | (1ull << in.x)}); break;
                case Inst::FAIL: break;
                }
            }
        }
        return out;
    }

    std::vector<std::pair<int, uint64_t>> shape_of(const TState& s) const {
        std::vector<std::pair<int, uint64_t>> k;
        k.push_back({-2, static_cast<uint64_t>(s.d)});
        for (auto& t : s.th) {
            uint64_t nil = 0;
            for (size_t g = 0; g < t.reg.size(); ++g) if (t.reg[g] == R_NIL) nil |= 1ull << g;
            k.push_back({t.pc, t.pending});
            k.push_back({-1, nil});
        }
        return k;
    }

    int add_state(const TState& s) {
        if (static_cast<int>(states_.size()) >= state_limit_) throw GxError(GX_E_LIMIT, "capture automaton too large");
        int id = static_cast<int>(states_.size());
        states_.push_back(s);
        exact_.emplace(s, id);
        by_shape_[shape_of(s)].push_back(id);
        return id;
    }

    // Sequentialise a set of parallel moves dst <- src (src < 0: the current position).
    std::vector<uint16_t> order_moves(std::vector<std::pair<int, int>> moves) {
        std::vector<uint16_t> out;
        moves.erase(std::remove_if(moves.begin(), moves.end(), [](const std::pair<int, int>& m) { return m.first == m.second; }), moves.end());
        while (!moves.empty()) {
            bool progress = false;
            for (size_t i = 0; i < moves.size(); ++i) {
                int dst = moves[i].first;
                bool blocked = false;
                for (size_t j = 0; j < moves.size(); ++j)
                    if (j != i && moves[j].second == dst) { blocked = true; break; }
                if (blocked) continue;
                out.push_back(static_cast<uint16_t>(dst));
                out.push_back(moves[i].second < 0 ? static_cast<uint16_t>(GX_SRC_POS) : static_cast<uint16_t>(moves[i].second));
                moves.erase(moves.begin() + i);
                progress = true;
                break;
            }
            if (progress) continue;
            // only cycles remain: break one with a scratch register
            if (scratch_ < 0) scratch_ = nregs_++;
            int src = moves[0].second;
            out.push_back(static_cast<uint16_t>(scratch_));
            out.push_back(static_cast<uint16_t>(src));
            for (auto& m : moves) if (m.second == src) m.second = scratch_;
        }
        return out;
    }

    uint32_t step(const TState& cur, int cls) {
        int nd = 0;
        if (fused_) {
            nd = static_cast<int>(fused_->m_next[static_cast<size_t>(cur.d) * C_.ncls + cls]);
            if (nd == fused_->m_dead) return DEAD_MARK;  // no extraction can match any more: the line's result is null
        }
        std::vector<Thread> seeds;
        size_t n_new = 0;   // R_NEW - k, k < n_new: the step's k-th new register
        for (const Thread& t : cur.th) {
            const Inst& in = p_.code[t.pc];
            if (in.op != Inst::CHAR || !C_.set_has(in.x, cls)) continue;
            Thread s;
            s.pc = t.pc + 1; s.pending = 0; s.reg = t.reg;
            if (t.pending) {
                // the thread's pending tags by run, in program order: the j-th run takes the step's j-th new register (threads of
                // other extractions that are at the same point of the line share it: one register per distinct live position)
                const std::vector<int>& runs = tag_run_[static_cast<size_t>(p_.rule_of_pc[t.pc])];
                std::vector<int> mine;
                for (size_t g = 0; g < s.reg.size(); ++g)
                    if (((t.pending >> g) & 1) && std::find(mine.begin(), mine.end(), runs[g]) == mine.end()) mine.push_back(runs[g]);
                std::sort(mine.begin(), mine.end());
                for (size_t g = 0; g < s.reg.size(); ++g)
                    if ((t.pending >> g) & 1) {
                        const size_t k = static_cast<size_t>(std::find(mine.begin(), mine.end(), runs[g]) - mine.begin());
                        s.reg[g] = static_cast<int16_t>(R_NEW - static_cast<int>(k));
                    }
                n_new = std::max(n_new, mine.size());
            }
            seeds.push_back(std::move(s));
        }
        std::vector<Thread> nxt;
        if (!seeds.empty()) nxt = close(seeds);
        if (nxt.empty() && !fused_) return DEAD_MARK;
        // registers that receive the current position, one per run of tags that a surviving thread set in this step (the j-th run of
        // every thread the same one): the lowest ones no surviving thread still reads, handed out in the order of the runs in the
        // program, so that what a state looks like does not depend on whether two runs were set in one step or in two
        std::vector<int> new_regs;        // in allocation order
        std::vector<int> reg_of_run(n_new, -1);
        {
            std::vector<char> used(n_new, 0);
            std::vector<char> busy(static_cast<size_t>(nregs_) + n_new + 1, 0);
            for (auto& t : nxt) for (int16_t r : t.reg) { if (r <= R_NEW) used[static_cast<size_t>(R_NEW - r)] = 1; else if (r >= 0) busy[static_cast<size_t>(r)] = 1; }
            int next_free = 0;
            for (size_t k = 0; k < n_new; ++k) {
                if (!used[k]) continue;
                while (next_free < nregs_ && (busy[static_cast<size_t>(next_free)] || next_free == scratch_)) ++next_free;
                if (next_free >= 0x7FF0) throw GxError(GX_E_LIMIT, "capture automaton needs too many registers");
                if (next_free >= nregs_) nregs_ = next_free + 1;
                reg_of_run[k] = next_free;
                new_regs.push_back(next_free);
                ++next_free;
            }
        }
        auto is_new = [&](int r) { return std::find(new_regs.begin(), new_regs.end(), r) != new_regs.end(); };
        TState resolved;
        resolved.d = nd;
        resolved.th = nxt;
        for (auto& t : resolved.th) for (auto& r : t.reg) if (r <= R_NEW) r = static_cast<int16_t>(reg_of_run[static_cast<size_t>(R_NEW - r)]);
        std::vector<std::pair<int, int>> moves;
        auto it = exact_.find(resolved);
        int target = -1;
        if (it != exact_.end()) {
            target = it->second;
            for (int r : new_regs) moves.push_back({r, -1});
        } else {
            // an existing state of the same shape whose registers can be produced from ours by moves?
            auto sh = by_shape_.find(shape_of(resolved));
            if (sh != by_shape_.end()) {
                for (int cand : sh->second) {
                    const TState& z = states_[cand];
                    std::map<int, int> src_of;  // z register -> our register
                    bool ok = true;
                    for (size_t j = 0; j < z.th.size() && ok; ++j)
                        for (size_t g = 0; g < z.th[j].reg.size() && ok; ++g) {
                            int rz = z.th[j].reg[g], ry = resolved.th[j].reg[g];
                            if (rz == R_NIL) continue;
                            auto f = src_of.find(rz);
                            if (f == src_of.end()) src_of[rz] = ry;
                            else if (f->second != ry) ok = false;
                        }
                    if (!ok) continue;
                    target = cand;
                    for (auto& kv : src_of) moves.push_back({kv.first, is_new(kv.second) ? -1 : kv.second});
                    break;
                }
            }
            if (target < 0) {
                target = add_state(resolved);
                for (int r : new_regs) moves.push_back({r, -1});
            }
        }
        if (target > 0xFFFE) throw GxError(GX_E_LIMIT, "capture automaton too large");
        int op_id = ops_.intern(order_moves(moves));
        return static_cast<uint32_t>(target) | (static_cast<uint32_t>(op_id) << 16);
    }

    // Info word of a state.  Per-extraction automaton: offset of its final tag record, or -1 (reject).
    // Fused automaton: offset of the record (which starts with the extraction index), -1 when no
    // extraction matches (Gorp.extract returns null), -2-k when the match automaton chose extraction k
    // but k's capture regex has no accepting thread (ExtractionException, core/Gorp.java:173-177).
    int32_t final_of(const TState& s) {
        int want = -1;
        if (fused_) {
            want = fused_->m_accept_first[s.d];
            if (want < 0) return -1;
        }
        for (const Thread& t : s.th) {
            if (p_.code[t.pc].op != Inst::MATCH) continue;
            if (fused_ && p_.code[t.pc].x != want) continue;
            int32_t off = static_cast<int32_t>(fin_tags_.size());
            if (with_rule_id_) fin_tags_.push_back(static_cast<uint16_t>(p_.code[t.pc].x));
            for (size_t g = 0; g < t.reg.size(); ++g) {
                if ((t.pending >> g) & 1) fin_tags_.push_back(GX_SRC_POS);
                else if (t.reg[g] == R_NIL) fin_tags_.push_back(GX_SRC_NIL);
                else fin_tags_.push_back(static_cast<uint16_t>(t.reg[g]));
            }
            if (!with_rule_id_ && t.reg.empty()) fin_tags_.push_back(GX_SRC_NIL);  // keep offsets distinct from -1
            return off;
        }
        return fused_ ? -2 - want : -1;
    }
};

}  // namespace

// ===========================================================================
Tables compile_tables(const std::vector<ustr>& automaton_rx, const std::vector<ustr>* jdk_rx) {
    const size_t n = automaton_rx.size();
    if (n == 0) throw GxError(GX_E_ARG, "no extractions");
    if (jdk_rx && jdk_rx->size() != n) throw GxError(GX_E_ARG, "pattern list sizes differ");
    SetPool pool;
    std::vector<Prog> aprog(n), jprog(jdk_rx ? n : 0);
    for (size_t k = 0; k < n; ++k) {
        try {
            Parsed pa = parse_automaton_dialect(automaton_rx[k]);
            ProgBuilder(aprog[k], pool, false).build(pa.root.get());
        } catch (GxError& e) {
            // message shape of core/autom/PolyMatcher.java:79-81
            throw GxError(e.code, std::string("Invalid regexp, ") + e.what() + ", source: " + u16_to_utf8(automaton_rx[k]));
        }
        if (jdk_rx) {
            try {
                Parsed pj = parse_jdk_dialect((*jdk_rx)[k]);
                jprog[k].ngroups = pj.ngroups;
                ProgBuilder(jprog[k], pool, true).build(pj.root.get());
            } catch (GxError& e) {
                throw GxError(e.code, std::string("Invalid capture regexp for extraction #") + std::to_string(k) + ": " + e.what() +
                                          ", source: " + u16_to_utf8((*jdk_rx)[k]));
            }
        }
    }
    Classes C = build_classes(pool);

    Tables T;
    T.n_rules = static_cast<int>(n);
    T.ncls = C.ncls;
    T.has_capture = jdk_rx != nullptr;
    // class maps
    {
        size_t bi = 0;
        for (int c = 0; c < 256; ++c) {
            while (bi + 1 < C.bounds.size() && C.bounds[bi + 1] <= c) ++bi;
            T.cls256[c] = static_cast<uint8_t>(C.bound_cls[bi]);
        }
        // classes are numbered by first occurrence in ascending code-unit order, so byte classes fit 8 bits
        T.hi_lo.push_back(256);
        {
            size_t i = std::upper_bound(C.bounds.begin(), C.bounds.end(), 256) - C.bounds.begin() - 1;
            T.hi_cls.push_back(static_cast<uint16_t>(C.bound_cls[i]));
            for (++i; i < C.bounds.size(); ++i) {
                if (static_cast<uint16_t>(C.bound_cls[i]) == T.hi_cls.back()) continue;
                T.hi_lo.push_back(static_cast<uint16_t>(C.bounds[i]));
                T.hi_cls.push_back(static_cast<uint16_t>(C.bound_cls[i]));
            }
        }
    }
    // match automaton
    std::vector<Dfa> parts;
    {
        parts.reserve(n);
        for (size_t k = 0; k < n; ++k) parts.push_back(minimize(subset_construct(aprog[k], C)));
        std::vector<std::vector<int32_t>> accept_sets;
        Dfa prod = minimize(product(parts, C.ncls, accept_sets));
        const int dead = prod.n;
        T.m_states = prod.n + 1;
        T.m_dead = dead;
        T.m_next.resize(static_cast<size_t>(T.m_states) * C.ncls);
        for (size_t i = 0; i < prod.next.size(); ++i) T.m_next[i] = prod.next[i] < 0 ? dead : prod.next[i];
        for (int c = 0; c < C.ncls; ++c) T.m_next[static_cast<size_t>(dead) * C.ncls + c] = dead;
        T.m_accept_off.push_back(0);
        for (int s = 0; s < T.m_states; ++s) {
            int lab = s == dead ? -1 : prod.label[s];
            if (lab < 0) T.m_accept_first.push_back(-1);
            else {
                T.m_accept_first.push_back(accept_sets[lab][0]);
                for (int32_t k : accept_sets[lab]) T.m_accept_list.push_back(k);
            }
            T.m_accept_off.push_back(static_cast<uint32_t>(T.m_accept_list.size()));
        }
    }
    // capture automata
    T.union_ok = false;
    if (jdk_rx) {
        OpListPool ops;
        std::vector<char> is_pike(n, 0);
        bool any_pike = false;
        for (size_t k = 0; k < n; ++k) {
            MultiProg mp = join_programs({&jprog[k]});
            const OpListPool ops_before = ops;
            const std::vector<uint16_t> fin_before = T.fin_tags;
            try {
                TdfaBuilder b(mp, C, ops, T.fin_tags, nullptr, PROGRAM_INSTEAD_STATES);
                T.rules.push_back(b.build());
            } catch (GxError& e) {
                if (e.code != GX_E_LIMIT || C.ncls > 256 || 2 * jprog[k].ngroups > MAX_TAGS) throw;
                // Too large ahead of time.  java.util.regex backtracks at run time and never builds anything: the extraction keeps its
                // program and the kernels run it as it is (gx_kernels.hip: pike_capture).
                ops = ops_before;
                T.fin_tags = fin_before;
                RuleTables R;
                R.n_groups = jprog[k].ngroups;
                R.n_states = 1;
                R.dead = 0;
                R.trans.assign(static_cast<size_t>(C.ncls), 0u);
                R.fin.assign(1, -1);
                R.pike = true;
                T.rules.push_back(R);
                is_pike[k] = 1;
                any_pike = true;
            }
            T.max_groups = std::max(T.max_groups, T.rules.back().n_groups);
        }
        if (any_pike) {
            std::map<int, uint32_t> set_index;
            T.pike_off.push_back(0);
            for (size_t k = 0; k < n; ++k) {
                if (is_pike[k]) {
                    for (const Inst& in : jprog[k].code) {
                        uint32_t x = static_cast<uint32_t>(in.x);
                        if (in.op == Inst::CHAR) {
                            auto it = set_index.find(in.x);
                            if (it == set_index.end()) {
                                it = set_index.emplace(in.x, static_cast<uint32_t>(T.pike_sets.size() / 8)).first;
                                for (int w = 0; w < 4; ++w) {
                                    const uint64_t m = w < C.words ? C.set_mask[static_cast<size_t>(in.x)][static_cast<size_t>(w)] : 0ull;
                                    T.pike_sets.push_back(static_cast<uint32_t>(m));
                                    T.pike_sets.push_back(static_cast<uint32_t>(m >> 32));
                                }
                            }
                            x = it->second;
                        }
                        if (x >= (1u << 24)) throw GxError(GX_E_LIMIT, "capture program too large");
                        T.pike_code.push_back(static_cast<uint32_t>(in.op) | x << 8);
                        T.pike_code.push_back(static_cast<uint32_t>(in.y));
                    }
                }
                T.pike_off.push_back(static_cast<uint32_t>(T.pike_code.size() / 2));
            }
        }
        // Fused automaton (single pass): every extraction's capture automaton, joined in priority order,
        // run in product with the match automaton.  The match component decides WHICH extraction wins
        // (automaton dialect, exactly as PolyMatcher does), the tagged component supplies that
        // extraction's captures (JDK dialect) or proves that its regex rejects the line.
        if (n <= 4096 && !any_pike) {
            try {
                std::vector<const Prog*> ps;
                for (auto& p : jprog) ps.push_back(&p);
                MultiProg mp = join_programs(ps);
                std::vector<uint16_t> fin = T.fin_tags;  // commit only on success
                OpListPool ops2 = ops;
                TdfaBuilder b(mp, C, ops2, fin, &T);
                T.uni = b.build();
                T.fin_tags.swap(fin);
                ops = ops2;
                T.union_ok = true;
            } catch (GxError& e) {
                if (e.code != GX_E_LIMIT) throw;
            }
        }
        T.ops_off.push_back(0);
        for (auto& l : ops.lists) {
            T.ops.insert(T.ops.end(), l.begin(), l.end());
            T.ops_off.push_back(static_cast<uint32_t>(T.ops.size() / 2));
        }
    } else {
        T.ops_off = {0, 0};
    }
    return T;
}

// ===========================================================================
// Blob: one relocatable little-endian image (the RCCL broadcast payload)
// ===========================================================================
namespace {
const uint32_t BLOB_MAGIC = 0x31425847u;  // "GXB1"
const uint32_t BLOB_VERSION = 3;   // 3: + the programs of extractions without an automaton (version 2 blobs are still read)

struct Writer {
    std::vector<uint8_t> buf;
    template <typename V> void pod(const V& v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); buf.insert(buf.end(), p, p + sizeof(V)); }
    template <typename V> void vec(const std::vector<V>& v) {
        pod<uint64_t>(v.size());
        const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data());
        buf.insert(buf.end(), p, p + v.size() * sizeof(V));
        while (buf.size() % 8) buf.push_back(0);
    }
};
struct Reader {
    const uint8_t* p; size_t left;
    template <typename V> V pod() {
        if (left < sizeof(V)) throw GxError(GX_E_ARG, "truncated table blob");
        V v; memcpy(&v, p, sizeof(V)); p += sizeof(V); left -= sizeof(V); return v;
    }
    template <typename V> std::vector<V> vec() {
        uint64_t n = pod<uint64_t>();
        if (n > left / sizeof(V)) throw GxError(GX_E_ARG, "truncated table blob");
        std::vector<V> v(n);
        if (n) memcpy(v.data(), p, n * sizeof(V));
        size_t adv = n * sizeof(V);
        adv = (adv + 7) & ~size_t(7);
        if (adv > left) adv = left;
        p += adv; left -= adv;
        return v;
    }
};
}  // namespace

std::vector<uint8_t> pack_blob(const Tables& t) {
    Writer w;
    w.pod(BLOB_MAGIC); w.pod(BLOB_VERSION);
    w.pod<int32_t>(t.n_rules); w.pod<int32_t>(t.ncls); w.pod<int32_t>(t.max_groups); w.pod<int32_t>(t.has_capture ? 1 : 0);
    w.pod<int32_t>(t.m_states); w.pod<int32_t>(t.m_dead);
    std::vector<uint8_t> c256(t.cls256, t.cls256 + 256);
    w.vec(c256); w.vec(t.hi_lo); w.vec(t.hi_cls);
    w.vec(t.m_next); w.vec(t.m_accept_first); w.vec(t.m_accept_off); w.vec(t.m_accept_list);
    w.pod<uint64_t>(t.rules.size());
    for (auto& r : t.rules) {
        w.pod<int32_t>(r.n_groups); w.pod<int32_t>(r.n_states); w.pod<int32_t>(r.n_regs); w.pod<int32_t>(r.dead);
        w.vec(r.trans); w.vec(r.fin);
    }
    w.vec(t.ops_off); w.vec(t.ops); w.vec(t.fin_tags);
    w.pod<int32_t>(t.union_ok ? 1 : 0);
    w.pod<int32_t>(0);  // keeps the following vectors 8-byte aligned
    if (t.union_ok) {
        w.pod<int32_t>(t.uni.n_groups); w.pod<int32_t>(t.uni.n_states); w.pod<int32_t>(t.uni.n_regs); w.pod<int32_t>(t.uni.dead);
        w.vec(t.uni.trans); w.vec(t.uni.fin);
    }
    w.vec(t.pike_off); w.vec(t.pike_code); w.vec(t.pike_sets);
    return w.buf;
}

Tables unpack_blob(const void* data, size_t size) {
    Reader r{static_cast<const uint8_t*>(data), size};
    if (r.pod<uint32_t>() != BLOB_MAGIC) throw GxError(GX_E_ARG, "not a gorp_amd table blob");
    const uint32_t version = r.pod<uint32_t>();
    if (version != BLOB_VERSION && version != 2u) throw GxError(GX_E_ARG, "table blob version mismatch");
    Tables t;
    t.n_rules = r.pod<int32_t>(); t.ncls = r.pod<int32_t>(); t.max_groups = r.pod<int32_t>(); t.has_capture = r.pod<int32_t>() != 0;
    t.m_states = r.pod<int32_t>(); t.m_dead = r.pod<int32_t>();
    std::vector<uint8_t> c256 = r.vec<uint8_t>();
    if (c256.size() != 256) throw GxError(GX_E_ARG, "corrupt table blob");
    memcpy(t.cls256, c256.data(), 256);
    t.hi_lo = r.vec<uint16_t>(); t.hi_cls = r.vec<uint16_t>();
    t.m_next = r.vec<uint32_t>(); t.m_accept_first = r.vec<int32_t>(); t.m_accept_off = r.vec<uint32_t>(); t.m_accept_list = r.vec<int32_t>();
    uint64_t nr = r.pod<uint64_t>();
    if (nr > 1000000) throw GxError(GX_E_ARG, "corrupt table blob");
    for (uint64_t i = 0; i < nr; ++i) {
        RuleTables rt;
        rt.n_groups = r.pod<int32_t>(); rt.n_states = r.pod<int32_t>(); rt.n_regs = r.pod<int32_t>(); rt.dead = r.pod<int32_t>();
        rt.trans = r.vec<uint32_t>(); rt.fin = r.vec<int32_t>();
        if (rt.trans.size() != static_cast<size_t>(rt.n_states) * t.ncls || rt.fin.size() != static_cast<size_t>(rt.n_states))
            throw GxError(GX_E_ARG, "corrupt table blob");
        t.rules.push_back(std::move(rt));
    }
    t.ops_off = r.vec<uint32_t>(); t.ops = r.vec<uint16_t>(); t.fin_tags = r.vec<uint16_t>();
    t.union_ok = r.pod<int32_t>() != 0;
    (void)r.pod<int32_t>();
    if (t.union_ok) {
        t.uni.n_groups = r.pod<int32_t>(); t.uni.n_states = r.pod<int32_t>(); t.uni.n_regs = r.pod<int32_t>(); t.uni.dead = r.pod<int32_t>();
        t.uni.trans = r.vec<uint32_t>(); t.uni.fin = r.vec<int32_t>();
        if (t.uni.trans.size() != static_cast<size_t>(t.uni.n_states) * t.ncls || t.uni.fin.size() != static_cast<size_t>(t.uni.n_states))
            throw GxError(GX_E_ARG, "corrupt table blob");
    }
    if (version >= 3u) { t.pike_off = r.vec<uint32_t>(); t.pike_code = r.vec<uint32_t>(); t.pike_sets = r.vec<uint32_t>(); }
    if (t.m_next.size() != static_cast<size_t>(t.m_states) * t.ncls || t.hi_lo.size() != t.hi_cls.size() || t.hi_lo.empty())
        throw GxError(GX_E_ARG, "corrupt table blob");
    // Everything the host walkers (gx_state_accepts, gx_match_one_utf16) and the kernels index with values read from
    // the blob is range-checked once here: a truncated or version-skewed blob is refused, never walked.
    auto bad = [](const char* what) { return GxError(GX_E_ARG, std::string("corrupt table blob: ") + what); };
    if (t.n_rules < 0 || t.ncls <= 0 || t.ncls > 65535 || t.max_groups < 0 || t.m_states <= 0) throw bad("header counts");
    if (t.m_dead < 0 || t.m_dead >= t.m_states) throw bad("dead state of the match automaton");
    for (int b = 0; b < 256; ++b) if (t.cls256[b] >= t.ncls) throw bad("class map");
    for (uint16_t c : t.hi_cls) if (c >= t.ncls) throw bad("class map above 255");
    for (uint32_t v : t.m_next) if (v >= static_cast<uint32_t>(t.m_states)) throw bad("successor of the match automaton");
    if (t.m_accept_first.size() != static_cast<size_t>(t.m_states) || t.m_accept_off.size() != static_cast<size_t>(t.m_states) + 1)
        throw bad("accept arrays");
    for (size_t s = 0; s < static_cast<size_t>(t.m_states); ++s) {
        if (t.m_accept_off[s] > t.m_accept_off[s + 1] || t.m_accept_off[s + 1] > t.m_accept_list.size()) throw bad("accept offsets");
        if (t.m_accept_first[s] < -1 || t.m_accept_first[s] >= t.n_rules) throw bad("first accepting extraction");
    }
    for (int32_t k : t.m_accept_list) if (k < 0 || k >= t.n_rules) throw bad("accept list");
    if (t.has_capture && t.rules.size() != static_cast<size_t>(t.n_rules)) throw bad("capture automata missing");
    if (!t.has_capture && (!t.rules.empty() || t.union_ok)) throw bad("capture automata without capture flag");
    if (t.ops_off.empty() || t.ops.size() % 2) throw bad("capture programs");
    for (size_t i = 0; i + 1 < t.ops_off.size(); ++i)
        if (t.ops_off[i] > t.ops_off[i + 1] || t.ops_off[i + 1] > t.ops.size() / 2) throw bad("capture program offsets");
    const size_t n_lists = t.ops_off.size() - 1;
    auto check_rule = [&](const RuleTables& rt, bool fused) {
        if (rt.n_states <= 0 || rt.n_groups < 0 || rt.n_regs < 0 || rt.dead < 0 || rt.dead >= rt.n_states) throw bad("capture automaton header");
        if (rt.n_groups > t.max_groups && !fused) throw bad("group count");
        for (uint32_t w : rt.trans) {
            if ((w & 0xFFFFu) >= static_cast<uint32_t>(rt.n_states)) throw bad("successor of a capture automaton");
            if ((w >> 16) >= n_lists) throw bad("capture program index");
        }
        for (int32_t f : rt.fin) {
            if (f < -1 && (!fused || -2 - f >= t.n_rules)) throw bad("final marker");
            if (f < 0) continue;
            size_t need = static_cast<size_t>(f);
            if (fused) {  // the list starts with the extraction
                if (need >= t.fin_tags.size() || t.fin_tags[need] >= t.rules.size()) throw bad("final record");
                need += 1 + 2 * static_cast<size_t>(t.rules[t.fin_tags[need]].n_groups);
            } else need += 2 * static_cast<size_t>(rt.n_groups);
            if (need > t.fin_tags.size()) throw bad("final record");
        }
    };
    for (auto& rt : t.rules) check_rule(rt, false);
    if (t.union_ok) check_rule(t.uni, true);
    int max_regs = 0;
    for (auto& rt : t.rules) max_regs = std::max(max_regs, rt.n_regs);
    if (t.union_ok) max_regs = std::max(max_regs, t.uni.n_regs);
    for (size_t i = 0; i < t.ops.size(); i += 2) {
        if (t.ops[i] >= max_regs) throw bad("capture program destination");
        if (t.ops[i + 1] != GX_SRC_POS && t.ops[i + 1] >= max_regs) throw bad("capture program source");
    }
    for (uint16_t v : t.fin_tags) if (v != GX_SRC_POS && v != GX_SRC_NIL && v >= std::max(max_regs, t.n_rules)) throw bad("final tag");
    // the programs of the extractions without an automaton: every target inside the extraction's own program, every set and tag in range
    if (!t.pike_code.empty() || !t.pike_off.empty() || !t.pike_sets.empty()) {
        if (!t.has_capture || t.union_ok || t.ncls > 256 || t.pike_off.size() != static_cast<size_t>(t.n_rules) + 1 || t.pike_code.size() % 2 || t.pike_sets.size() % 8 ||
            t.pike_off[0] != 0 || t.pike_off.back() != t.pike_code.size() / 2)
            throw bad("capture programs of extractions without an automaton");
        for (size_t k = 0; k < static_cast<size_t>(t.n_rules); ++k) {
            if (t.pike_off[k] > t.pike_off[k + 1]) throw bad("capture program offsets of extractions without an automaton");
            const uint32_t len = t.pike_off[k + 1] - t.pike_off[k];
            t.rules[k].pike = len != 0;
            if (len > 65535u) throw bad("capture program length");
            for (uint32_t q = 0; q < len; ++q) {
                const uint32_t w0 = t.pike_code[2 * (t.pike_off[k] + q)], y = t.pike_code[2 * (t.pike_off[k] + q) + 1];
                const uint32_t op = w0 & 0xFFu, x = w0 >> 8;
                switch (op) {
                case 0: if (x >= t.pike_sets.size() / 8 || q + 1 >= len) throw bad("capture program: set"); break;
                case 1: if (x >= len || y >= len) throw bad("capture program: split"); break;
                case 2: if (x >= len) throw bad("capture program: jump"); break;
                case 3: if (x >= 2u * static_cast<uint32_t>(t.rules[k].n_groups) || q + 1 >= len) throw bad("capture program: tag"); break;
                case 4: case 5: break;
                default: throw bad("capture program: instruction");
                }
            }
        }
    }
    return t;
}

}  // namespace gx
