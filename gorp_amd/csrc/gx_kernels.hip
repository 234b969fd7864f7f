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
// Generic kernel: one lane per line, tables read through L1/L2.  Handles every
// table size (16- or 32-bit match states), any line length, bytes or UTF-16.
// It is the correctness backstop; the LDS-tier kernels below are the fast path.
// ---------------------------------------------------------------------------
template <typename CH, typename OFF, typename MS>
__global__ void __launch_bounds__(256)
k_extract_generic(GxDev T, const CH* __restrict__ data, const OFF* __restrict__ off, uint64_t n,
                  int32_t* __restrict__ match_id, int32_t* __restrict__ caps, int32_t* __restrict__ state_out,
                  int match_only, const MS* __restrict__ m_next) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const int ncls = T.ncls;
    const int slots = 2 * T.max_groups;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = off[i], e = off[i + 1];
        const CH* s = data + b;
        const int64_t len = static_cast<int64_t>(e - b);

        // ---- hot loop #1: walk the match automaton ----
        uint32_t st = 0;
        const uint32_t dead = static_cast<uint32_t>(T.m_dead);
        for (int64_t p = 0; p < len; ++p) {
            st = m_next[static_cast<size_t>(st) * ncls + class_of(T, s[p])];
            if (st == dead) break;  // the reference's early return on -1
        }
        const int32_t k = T.m_accept_first[st];
        if (state_out) state_out[i] = (st == dead) ? -1 : static_cast<int32_t>(st);
        if (match_only || !T.has_capture) { match_id[i] = k; continue; }

        int32_t* cp = caps + i * static_cast<uint64_t>(slots);
        if (k < 0) {
            match_id[i] = -1;
            for (int t = 0; t < slots; ++t) cp[t] = -1;
            continue;
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
        if (f < 0) {  // DFA said yes, capture regex says no -> ExtractionException
            match_id[i] = -2 - k;
            for (int t = 0; t < slots; ++t) cp[t] = -1;
            continue;
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
                           b.state_out, b.match_only, dev.m_next16);
    else
        hipLaunchKernelGGL((k_extract_generic<CH, OFF, uint32_t>), grid, dim3(block), 0, stream, dev,
                           static_cast<const CH*>(b.data), static_cast<const OFF*>(b.offsets), b.n, b.match_id, b.caps,
                           b.state_out, b.match_only, dev.m_next32);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_extract_generic(const GxDev& dev, const GxBatch& b, hipStream_t stream) {
    if (b.wide) {
        return b.offsets64 ? launch_generic_t<uint16_t, uint64_t>(dev, b, stream) : launch_generic_t<uint16_t, uint32_t>(dev, b, stream);
    }
    return b.offsets64 ? launch_generic_t<uint8_t, uint64_t>(dev, b, stream) : launch_generic_t<uint8_t, uint32_t>(dev, b, stream);
}

}  // namespace gx
