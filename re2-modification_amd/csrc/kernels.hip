// HIP kernels of the match path, written for gfx950 (MI355X): 64-wide wavefronts,
// one input string per lane, per-string automaton state in LDS (struct-of-arrays, one
// bank per lane), the automaton itself read through the scalar cache.
//
//   mfa_walk_kernel  -- MFA::match (reference mfa.cpp:215-236) for a batch of strings
//   dfa_walk_kernel  -- Automata::match (reference automata.cpp:177-210) on the tabulated
//                       step function (image_host.cpp: tabulate_nfa)
//
// ---- how the MFA kernel restates mfa.cpp ---------------------------------------------------
// The reference keeps a std::set of (pos, node, memory) states, but each step only the first
// state per node (in set order) is evaluated and all others are dropped (mfa.cpp:206-211).
// So a string's live state is one SLOT per automaton node, holding the minimum state for that
// node under the set order:
//     (pos, name of the first memory cell, allocation order of that cell's Variable).
// Allocation order is creation order of the state (copy_memory allocates the cells when the
// state is created, mfa.cpp:107-114), which for states created within one step is
//     (pos of the source state, node of the source state, DFS index of the creating edge)
// because sources are evaluated in (pos, node) order; a state re-inserted unchanged (the
// "wait" branch, mfa.cpp:195-197) keeps its old Variables and is therefore older than every
// state created in the current step.  Slots carry that key as three words P, Q, R.
// Memory cell values are always contiguous spans of the scan-order input (every write appends
// exactly the text just consumed, mfa.cpp:89-104), so a cell is (start, len, flags).
// `finish` is reached only through epsilon edges and only kept when pos == len
// (mfa.cpp:138-140,143-147): it is an accept flag, and once set it stays set.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "mfa_internal.h"
#ifndef MFA_DFA_LINE
#define MFA_DFA_LINE 128
#endif
#include "device_common.h"

namespace mfa {

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) { set_last_hip_error((int)e_); return MFA_ERR_HIP; } \
    } while (0)

static constexpr uint32_t kEmpty = MFA_EMPTY;

template <int K>
struct Mem {                 // the memory of one state, in registers
    uint32_t start[K];
    uint32_t len[K];
    uint32_t fl[K];          // F_* | ch << 8
};

struct DevImg {
    const uint32_t* edge_begin;
    const uint2*    edges;
    uint32_t n_nodes, start, finish, reversed;
};

// ---- LDS slot arrays --------------------------------------------------------------------------
// word index of (buf, node, w) for this lane:  (((buf * N + node) * W + w) * 64 + lane
template <int K>
struct SlotLayout {
    static constexpr int W = 4 + 3 * K;      // P, Q, R (low word), then per cell start, len, flags, then R's high word
    static constexpr int RHI = 3 + 3 * K;    // R = 5 bits per level of the unset-cell recursion, K + 1 levels: 50 bits for nine cells
};

template <int K, bool LDS_SLOTS>
struct Slots {
    uint32_t* base;          // LDS or global scratch, already offset to this wave and lane
    uint32_t  N;
    __device__ __forceinline__ uint32_t* at(uint32_t buf, uint32_t node, uint32_t w) const {
        return base + (((buf * N + node) * SlotLayout<K>::W + w) << 6);
    }
};

template <int K>
__device__ __forceinline__ uint32_t first_name(const Mem<K>& m) {
#pragma unroll
    for (int c = 0; c < K; c++)
        if (m.fl[c] & F_PRESENT) return (uint32_t)(c + 1);
    return 0u;
}

// ---- the walk -----------------------------------------------------------------------------------
template <int K, bool REV, bool LDS_SLOTS>
struct Walker {
    DevImg   g;
    Slots<K, LDS_SLOTS> S;
    Input    in;
    uint32_t curbuf;         // wave-uniform
    uint32_t turn = 0;       // wave-uniform: iterations since the byte windows were last turned over (device_common.h)
    // per-lane
    bool     active;         // has a string in flight
    bool     accept;
    bool     any_next;       // something was inserted into the next state set this step
    uint32_t i;              // step index (scan position)
    uint32_t ch;             // scan[i]
    bool     final_pass;     // i == len

    __device__ __forceinline__ void load_mem(uint32_t buf, uint32_t node, Mem<K>& m) const {
#pragma unroll
        for (int c = 0; c < K; c++) {
            m.start[c] = *S.at(buf, node, 3 + 3 * c);
            m.len[c]   = *S.at(buf, node, 4 + 3 * c);
            m.fl[c]    = *S.at(buf, node, 5 + 3 * c);
        }
    }

    // insert (P,Q,R,m) into the next-set slot of `node` if it beats what is there
    __device__ __forceinline__ void insert(bool pred, uint32_t node, uint32_t P, uint32_t Q, uint64_t R, const Mem<K>& m) {
        uint32_t nb = curbuf ^ 1u;
        uint32_t eP = *S.at(nb, node, 0), eQ = *S.at(nb, node, 1);
        const uint64_t eR = ((uint64_t)*S.at(nb, node, SlotLayout<K>::RHI) << 32) | *S.at(nb, node, 2);
        bool win = pred && (P < eP || (P == eP && (Q < eQ || (Q == eQ && R < eR))));
        if (win) {
            *S.at(nb, node, 0) = P; *S.at(nb, node, 1) = Q; *S.at(nb, node, 2) = (uint32_t)R; *S.at(nb, node, SlotLayout<K>::RHI) = (uint32_t)(R >> 32);
#pragma unroll
            for (int c = 0; c < K; c++) {
                *S.at(nb, node, 3 + 3 * c) = m.start[c];
                *S.at(nb, node, 4 + 3 * c) = m.len[c];
                *S.at(nb, node, 5 + 3 * c) = m.fl[c];
            }
            any_next = true;
        }
    }

    // MFA::doMemoryWriteActions (mfa.cpp:80-105) on a copy; the consumed text is
    // scan[tstart, tstart+tlen), all bytes equal `tch` iff tuni
    __device__ __forceinline__ void apply_actions(Mem<K>& m, uint32_t actions, uint32_t tstart, uint32_t tlen,
                                                  bool tuni, uint32_t tch) const {
#pragma unroll
        for (int c = 0; c < K; c++) {
            uint32_t act = (actions >> (2 * (c + 1))) & 3u;       // wave-uniform
            uint32_t f = m.fl[c];
            if (act == MFA_ACT_OPEN) {                            // create if absent, open(), write(t)
                m.start[c] = tstart; m.len[c] = tlen;
                m.fl[c] = F_PRESENT | F_OPEN | (tuni ? F_UNI : 0u) | (tch << 8);
            } else if (act == MFA_ACT_CLOSE) {                    // close()
                m.fl[c] = f & ~F_OPEN;
            } else {                                              // write(t) when open
                bool w = (f & (F_PRESENT | F_OPEN)) == (F_PRESENT | F_OPEN) && tlen != 0u;
                bool was_empty = m.len[c] == 0u;
                uint32_t fch = (f >> 8) & 0xffu;
                bool uni = was_empty ? tuni : ((f & F_UNI) && tuni && fch == tch);
                uint32_t nf = (f & (F_PRESENT | F_OPEN | F_READ)) | (uni ? F_UNI : 0u) | ((was_empty ? tch : fch) << 8);
                if (w) {
                    if (was_empty) m.start[c] = tstart;
                    m.len[c] += tlen;
                    m.fl[c] = nf;
                }
            }
        }
    }

    __device__ __forceinline__ bool suffix_ok(bool pred, const Mem<K>& m) const {        // mfa.cpp:116-133
        if (!REV) return pred;
        uint32_t needed = 0;
#pragma unroll
        for (int c = 0; c < K; c++)
            if ((m.fl[c] & F_PRESENT) && ((m.fl[c] & F_OPEN) || !(m.fl[c] & F_READ))) needed += m.len[c];
        return pred && needed <= in.len - i;
    }

    // MFA::evaluateState (mfa.cpp:136-200) for the lanes in `live`, all of which sit on `node`
    // with state (pos, m).  LEVEL = recursion depth through "unset cell" edges (mfa.cpp:148-160).
    template <int LEVEL>
    __device__ __forceinline__ void eval_node(uint32_t node, bool live, uint32_t pos, Mem<K>& m, uint32_t Q, uint64_t Rprefix) {
        live = suffix_ok(live, m);
        if (!__any(live)) return;
        const bool here    = live && !final_pass && pos == i;     // may consume (mfa.cpp:161)
        const bool waiting = live && !final_pass && pos > i;      // mfa.cpp:195
        bool reinsert = false;
        const uint32_t e0 = g.edge_begin[node], e1 = g.edge_begin[node + 1];
        for (uint32_t e = e0; e < e1; e++) {
            const uint2 ed = g.edges[e];                           // uniform -> scalar load
            const uint32_t label = ed.x & 0xffu, eflags = (ed.x >> 8) & 0xffu, target = ed.x >> 16, actions = ed.y;
            const uint64_t R = Rprefix | ((uint64_t)(e - e0 + 1u) << (5 * (K - LEVEL)));
            if (eflags & MFA_EDGE_EPS) {                           // mfa.cpp:143-147 -> finish keeps it iff pos == len
                if (live && pos == in.len) accept = true;
                continue;
            }
            const bool is_digit = label >= '1' && label <= '9';
            const int  d = is_digit ? (int)label - '1' : 0;       // 0-based cell, uniform
            bool absent = false;
            if (is_digit) {
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c == d) absent = !(m.fl[c] & F_PRESENT);
            }
            const bool take_absent = live && absent;
            if constexpr (LEVEL < K) {
                if (__any(take_absent)) {                          // mfa.cpp:148-160: create the cell, recurse, consume nothing
                    Mem<K> t = m;
                    uint32_t act = (actions >> (2 * (d + 1))) & 3u;
#pragma unroll
                    for (int c = 0; c < K; c++)
                        if (c == d) {
                            t.start[c] = pos; t.len[c] = 0u;
                            t.fl[c] = F_PRESENT | (act == MFA_ACT_OPEN ? F_OPEN : 0u) | F_UNI;
                        }
                    eval_node<LEVEL + 1>(target, take_absent, pos, t, Q, R);
                }
            }
            const bool other = live && !absent;
            const bool consume = other && here;
            const bool lit = consume && (label == '.' || label == ch);             // mfa.cpp:171
            if (__any(lit)) {
                Mem<K> t = m;
                apply_actions(t, actions, i, 1u, true, ch);
                insert(lit, target, ((pos + 1u) << 4) | first_name(t), Q, R, t);
            }
            const bool rd = consume && !lit && is_digit;                           // mfa.cpp:176
            if (__any(rd)) {
                Mem<K> t = m;                                     // copy BEFORE read() marks the source (mfa.cpp:167 vs 177)
                uint32_t vs = 0, vl = 0, vf = 0;
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c == d) { vs = m.start[c]; vl = m.len[c]; vf = m.fl[c]; if (rd) m.fl[c] |= F_READ; }
                bool ok = false;
                if (rd) ok = read_matches<REV>(in, i, ch, vs, vl, vf);
                if (__any(ok)) {
                    apply_actions(t, actions, i, vl, (vf & F_UNI) != 0u, (vf >> 8) & 0xffu);
                    insert(ok, target, ((pos + vl) << 4) | first_name(t), Q, R, t);
                }
            }
            reinsert = reinsert || (other && waiting);                             // mfa.cpp:195-197
        }
        if (__any(reinsert)) {
            // the state itself goes back into the set; at LEVEL 0 it is the old object (older than
            // anything created this step: Q = R = 0), deeper it is the state the unset-cell edge made
            insert(reinsert, node, (pos << 4) | first_name(m), LEVEL == 0 ? 0u : Q, LEVEL == 0 ? (uint64_t)0 : Rprefix, m);
        }
    }

    __device__ __forceinline__ void clear_lane_slots() {
        for (uint32_t n = 0; n < g.n_nodes; n++) { *S.at(0, n, 0) = kEmpty; *S.at(1, n, 0) = kEmpty; }
    }

    __device__ __forceinline__ void start_string(uint64_t base, uint32_t len) {
        input_reset(in, base, len);
        i = 0; accept = false; active = true;
        clear_lane_slots();
        *S.at(curbuf, g.start, 0) = 0u;                            // (pos 0, start, {})  mfa.cpp:217-219
#pragma unroll
        for (int c = 0; c < K; c++) {
            *S.at(curbuf, g.start, 3 + 3 * c) = 0u; *S.at(curbuf, g.start, 4 + 3 * c) = 0u; *S.at(curbuf, g.start, 5 + 3 * c) = 0u;
        }
    }

    // one evaluateStates call (mfa.cpp:203-213) for every active lane; returns per lane
    // whether the string is finished after it
    __device__ __forceinline__ bool step() {
        final_pass = (i == in.len);
        ch = 0x100u;
        if (turn == 0u) window_turn<REV>(in, i, active && !final_pass);
        if (active && !final_pass) ch = stream_byte<REV>(in, i, 16u - turn);
        turn = (turn + 1u) & 15u;
        any_next = false;
        for (uint32_t n = 0; n < g.n_nodes; n++) {
            uint32_t P = active ? *S.at(curbuf, n, 0) : kEmpty;
            bool live = P != kEmpty;
            if (!__any(live)) continue;
            if (live) *S.at(curbuf, n, 0) = kEmpty;               // consumed: this buffer is the next step's empty set
            uint32_t pos = P >> 4;
            live = live && pos >= i;                               // pos < i: nothing can be inserted from a stale state
            if (!__any(live)) continue;
            Mem<K> m;
            load_mem(curbuf, n, m);
            eval_node<0>(n, live, pos, m, (pos << 8) | n, 0u);
        }
        bool done = false;
        if (active) {
            if (accept || final_pass || !any_next) done = true;    // mfa.cpp:224-225, 227-235
            i++;
        }
        return done;
    }
};

template <int K, bool REV, bool LDS_SLOTS>
__global__ void __launch_bounds__(64)
mfa_walk_kernel(DevImg g, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                uint8_t* __restrict__ results, unsigned long long* counter, uint32_t* scratch) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    Walker<K, REV, LDS_SLOTS> w;
    w.g = g;
    w.S.N = g.n_nodes;
    if (LDS_SLOTS) w.S.base = lds + lane;
    else w.S.base = scratch + (size_t)blockIdx.x * (2u * g.n_nodes * SlotLayout<K>::W * 64u) + lane;
    w.in.bytes = bytes;
    w.in.total16 = (offsets[n] + 15u) & ~(uint64_t)15;
    input_reset(w.in, 0, 0);
    w.curbuf = 0;
    w.active = false; w.accept = false; w.i = 0;
    uint64_t sid = 0;
    bool exhausted = false;
    for (;;) {
        // hand a new string to every idle lane
        if (!w.active && !exhausted) {
            for (;;) {
                sid = atomicAdd(counter, 1ull);
                if (sid >= n) { exhausted = true; break; }
                uint64_t b = offsets[sid], e = offsets[sid + 1];
                uint64_t len = e - b;
                if (len > MFA_MAX_STRING_BYTES) { results[sid] = 2; continue; }    // flagged; the launcher reports it
                w.start_string(b, (uint32_t)len);
                break;
            }
        }
        if (!__any(w.active)) break;
        bool done = w.step();
        if (done) {
            results[sid] = w.accept ? 1 : 0;
            w.active = false;
        }
        w.curbuf ^= 1u;
    }
}

// ---- table walk for memory-less automata -------------------------------------------------------
// One string per lane.  LDS holds one fused table: next[state][byte] (16-bit entries, the state
// pre-multiplied by the row stride), so a step is: extract byte, OR it into the state word, one
// ds_read_u16.  Rows are padded by one dword so that equal bytes in different states fall into
// different banks.  State 0 is the empty set: absorbing and rejecting, the reference's early
// `break` (automata.cpp:186-188,196-198).  Input is read 16 bytes per lane per load.
static constexpr uint32_t kDfaRow = 258;     // 16-bit entries per state row (256 + 2 pad)

template <bool REV>
__global__ void __launch_bounds__(256)
dfa_walk_kernel(const uint16_t* __restrict__ trans, const uint8_t* __restrict__ accept_tab,
                const uint8_t* __restrict__ byte_class, uint32_t n_states, uint32_t n_classes,
                const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                uint8_t* __restrict__ results) {
    extern __shared__ uint32_t lds[];
    uint16_t* s_next = reinterpret_cast<uint16_t*>(lds);             // [n_states][kDfaRow], entry = next_state * kDfaRow
    for (uint32_t k = threadIdx.x; k < n_states * 256u; k += blockDim.x) {
        const uint32_t st = k >> 8, b = k & 255u;
        s_next[st * kDfaRow + b] = (uint16_t)(trans[st * n_classes + byte_class[b]] * kDfaRow);
    }
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t sid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sid < n; sid += stride) {
        const uint64_t b = offsets[sid], e = offsets[sid + 1];
        uint32_t st = kDfaRow;                                        // state 1 = {start}
        if (!REV) {
            uint64_t p = b;
            while (p < e && st != 0u) {
                const uint64_t blk = p & ~(uint64_t)15;
                const uint4 d = load16(bytes, blk);
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
                const uint32_t lo = (uint32_t)(p - blk), hi = (e - blk) < 16u ? (uint32_t)(e - blk) : 16u;
                if (lo == 0u && hi == 16u) {
#pragma unroll
                    for (int k = 0; k < 16; k++) st = s_next[st + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const uint32_t nx = s_next[st + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                        st = ((uint32_t)k >= lo && (uint32_t)k < hi) ? nx : st;
                    }
                }
                p = blk + 16u;
            }
        } else {
            uint64_t p = e;                                           // exclusive upper end, walk downwards
            while (p > b && st != 0u) {
                const uint64_t blk = (p - 1u) & ~(uint64_t)15;
                const uint4 d = load16(bytes, blk);
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
                const uint32_t hi = (uint32_t)(p - blk), lo = b > blk ? (uint32_t)(b - blk) : 0u;
                if (lo == 0u && hi == 16u) {
#pragma unroll
                    for (int k = 15; k >= 0; k--) st = s_next[st + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                } else {
#pragma unroll
                    for (int k = 15; k >= 0; k--) {
                        const uint32_t nx = s_next[st + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                        st = ((uint32_t)k >= lo && (uint32_t)k < hi) ? nx : st;
                    }
                }
                p = blk;
            }
        }
        results[sid] = accept_tab[st / kDfaRow];
    }
}

// ---- tiled table walk -----------------------------------------------------------------------------
// Same walk, input staged through LDS so that HBM is read in whole 128-byte lines (MFA_DFA_LINE; 64-byte
// rows double the resident waves but measured 3.7 against 4.55 TB/s; a second table stepping two bytes per
// dependent read -- n_states * classes^2 entries -- measured 4.4: the walk is not bound by that chain): a wave owns 64
// strings; each round, groups of 8 lanes fetch one 128-byte line of one string (8 x 16 B, a single
// contiguous segment per group), the 64 lines are written to a padded LDS tile, and every lane then
// reads its own string's line back with ds_read_b128.  The pad (144-byte row stride) keeps the 16
// lanes of a ds_read_b128 group on distinct banks.
//   PACKED: automata with <= 8 state sets and <= 7 literal byte classes keep the whole table in
//   SGPRs -- one 32-bit word per class, 4 bits per state -- so a step is a byte compare/select
//   (independent of the state) plus a 2-instruction dependent chain (shift, bit-field extract)
//   instead of an LDS round trip.
struct DfaPacked {
    uint32_t n_lit;          // literal classes
    uint32_t lit[7];         // their bytes
    uint32_t tab[8];         // tab[c]: nibble s = next state of s on class c; tab[n_lit] = every other byte
    uint32_t accept_mask;    // bit s = state s accepts
};

static constexpr uint32_t kLine = MFA_DFA_LINE;          // bytes of a string staged per round (one row of the tile)
static constexpr uint32_t kTileRow = kLine + 16;         // row stride in the LDS tile: the pad keeps ds_read_b128 groups on distinct banks
static constexpr uint32_t kLineLanes = kLine / 16;       // lanes that fetch one row, 16 bytes each
static constexpr uint32_t kFetches = kLineLanes;         // 64 / kLineLanes strings per load instruction -> kLineLanes instructions per round

template <bool REV, bool PACKED, int NLIT>
__global__ void __launch_bounds__(256)
dfa_tiled_kernel(DfaPacked pk, const uint16_t* __restrict__ trans, const uint8_t* __restrict__ accept_tab,
                 const uint8_t* __restrict__ byte_class, uint32_t n_states, uint32_t n_classes,
                 const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                 uint8_t* __restrict__ results) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t* tile = reinterpret_cast<uint8_t*>(lds) + wave * (64u * kTileRow);
    uint16_t* s_next = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(lds) + 4u * 64u * kTileRow);
    if (!PACKED) {
        for (uint32_t k = threadIdx.x; k < n_states * 256u; k += blockDim.x) {
            const uint32_t st = k >> 8, b = k & 255u;
            s_next[st * kDfaRow + b] = (uint16_t)(trans[st * n_classes + byte_class[b]] * kDfaRow);
        }
        __syncthreads();
    }
    const uint64_t total16 = (offsets[n] + 15u) & ~(uint64_t)15;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
    uint32_t lit[NLIT > 0 ? NLIT : 1], tabs[NLIT + 1];
#pragma unroll
    for (int c = 0; c < NLIT; c++) lit[c] = pk.lit[c];
#pragma unroll
    for (int c = 0; c <= NLIT; c++) tabs[c] = pk.tab[c];
    for (uint64_t w0 = ((uint64_t)blockIdx.x * 4u + wave) * 64u; w0 < n; w0 += n_waves * 64u) {
        const uint64_t sid = w0 + lane;
        const bool have = sid < n;
        const uint64_t b = have ? offsets[sid] : 0, e = have ? offsets[sid + 1] : 0;
        uint64_t p = REV ? e : b;                 // forward: next byte to consume; reverse: one past it
        uint32_t st = PACKED ? 1u : kDfaRow;      // state 1 = {start}
        bool active = have && (REV ? p > b : p < e);
        if (__any(active)) {                      // (a wave of empty strings has nothing to read)
        uint64_t line = (REV ? p - 1u : p) & ~(uint64_t)(kLine - 1u);
        // fetch: lane group g = lane>>3 serves strings g, g+8, ..., one 128-byte line each
        uint4 v[kFetches];
        auto fetch = [&](uint64_t ln, bool act) {
#pragma unroll
            for (int k = 0; k < (int)kFetches; k++) {
                const int src = k * (int)(64u / kLineLanes) + (int)(lane / kLineLanes);
                const uint32_t lo = __shfl((uint32_t)ln, src), hi = __shfl((uint32_t)(ln >> 32), src);
                const int a = __shfl((int)act, src);
                const uint64_t addr = (((uint64_t)hi << 32) | lo) + (lane % kLineLanes) * 16u;
                v[k] = (a && addr < total16) ? load16(bytes, addr) : make_uint4(0, 0, 0, 0);
            }
        };
        fetch(line, active);
        for (;;) {
#pragma unroll
            for (int k = 0; k < (int)kFetches; k++) {
                const uint32_t src = (uint32_t)k * (64u / kLineLanes) + lane / kLineLanes;
                *reinterpret_cast<uint4*>(tile + src * kTileRow + (lane % kLineLanes) * 16u) = v[k];
            }
            __builtin_amdgcn_wave_barrier();
            // the next line of every string (if it has one) is requested now and arrives while this one is walked
            const uint64_t p_next = REV ? line : line + kLine;
            fetch((REV ? p_next - 1u : p_next) & ~(uint64_t)(kLine - 1u), active && (REV ? p_next > b : p_next < e));
            // walk this lane's line
            const uint32_t lo_b = active ? (uint32_t)((REV ? (b > line ? b - line : 0) : p - line)) : 0u;
            const uint32_t hi_b = active ? (uint32_t)((REV ? p - line : (e - line < kLine ? e - line : kLine))) : 0u;
            const bool full = __all(!active || (lo_b == 0u && hi_b == kLine));
            if (!PACKED && full && __all(active)) {
                // every lane walks a whole line: nothing to mask (the dead state 0 maps to itself, so a string that
                // dies inside the line stays dead) -- three instructions per byte: extract, add, table read
#pragma unroll 1
                for (int q = 0; q < (int)kLineLanes; q++) {
                    const int qq = REV ? (int)kLineLanes - 1 - q : q;
                    const uint4 d = *reinterpret_cast<const uint4*>(tile + lane * kTileRow + (uint32_t)qq * 16u);
                    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) {
                        const int k = REV ? 15 - kk : kk;
                        st = s_next[st + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                    }
                }
            } else
#pragma unroll 1
            for (int q = 0; q < (int)kLineLanes; q++) {
                const int qq = REV ? (int)kLineLanes - 1 - q : q;
                const uint4 d = *reinterpret_cast<const uint4*>(tile + lane * kTileRow + (uint32_t)qq * 16u);
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    const int k = REV ? 15 - kk : kk;
                    const uint32_t byte = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    uint32_t nx;
                    if (PACKED) {
                        uint32_t sel = tabs[NLIT];
#pragma unroll
                        for (int c = 0; c < NLIT; c++) sel = (byte == lit[c]) ? tabs[c] : sel;
                        nx = __builtin_amdgcn_ubfe(sel, st << 2, 4u);
                    } else {
                        nx = s_next[st + byte];
                    }
                    const uint32_t idx = (uint32_t)qq * 16u + (uint32_t)k;
                    st = (full || (idx >= lo_b && idx < hi_b)) ? (active ? nx : st) : st;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (active) p = p_next;
            if (!REV && p > e) p = e;
            active = have && st != 0u && (REV ? p > b : p < e);
            if (!__any(active)) break;
            line = (REV ? p - 1u : p) & ~(uint64_t)(kLine - 1u);
        }
        }
        if (have) results[sid] = PACKED ? (uint8_t)((pk.accept_mask >> st) & 1u) : accept_tab[st / kDfaRow];
    }
}

// Tabulated automata whose table does not fit LDS (more than 127 state sets): the table stays in global memory -- a few
// hundred KiB at most, resident in L2 -- and only the byte classes go to LDS.  One string per lane, 32-bit state.
template <bool REV>
__global__ void __launch_bounds__(256)
dfa_big_kernel(const uint16_t* __restrict__ trans, const uint8_t* __restrict__ accept_tab, const uint8_t* __restrict__ byte_class,
               uint32_t n_classes, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
               uint8_t* __restrict__ results) {
    __shared__ uint8_t s_class[256];
    s_class[threadIdx.x] = byte_class[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t sid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sid < n; sid += stride) {
        const uint64_t b = offsets[sid], e = offsets[sid + 1];
        uint32_t st = 1u;                                             // state 1 = {start}, state 0 = the empty set
        uint64_t p = REV ? e : b;
        while ((REV ? p > b : p < e) && st != 0u) {
            const uint64_t blk = (REV ? p - 1u : p) & ~(uint64_t)15;
            const uint4 d = load16(bytes, blk);
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
            const uint32_t lo = b > blk ? (uint32_t)(b - blk) : 0u, hi = (e - blk) < 16u ? (uint32_t)(e - blk) : 16u;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int k = REV ? 15 - j : j;
                if ((uint32_t)k >= lo && (uint32_t)k < hi) st = trans[st * n_classes + s_class[(w[k >> 2] >> (8 * (k & 3))) & 0xffu]];
            }
            p = REV ? blk : blk + 16u;
        }
        results[sid] = accept_tab[st];
    }
}

// ---- launchers ------------------------------------------------------------------------------------
template <int K, bool REV>
static int launch_mfa_k(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                        uint64_t n, uint8_t* d_results, hipStream_t stream) {
    DevImg g{ds.d_edge_begin, ds.d_edges, img.h.n_nodes, img.h.start, img.h.finish, img.h.is_reversed};
    const size_t per_wave = (size_t)2 * img.h.n_nodes * SlotLayout<K>::W * 64 * sizeof(uint32_t);
    const bool lds_slots = per_wave <= 40 * 1024;             // >= 4 waves per CU out of 160 KiB
    HIP_TRY(hipMemsetAsync(cx.d_counter, 0, sizeof(unsigned long long), stream));
    int waves_per_cu = 8;
    if (lds_slots) {
        int by_lds = (int)((160 * 1024) / per_wave);
        if (by_lds < waves_per_cu) waves_per_cu = by_lds;
    }
    uint64_t want = (n + 63) / 64;
    uint64_t grid = (uint64_t)ds.n_cus * waves_per_cu;
    if (grid > want) grid = want;
    if (grid == 0) grid = 1;
    if (!lds_slots) {
        int rc = ctx_reserve((void**)&cx.d_scratch, &cx.scratch_bytes, (size_t)grid * per_wave);
        if (rc != MFA_OK) return rc;
    }
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_start, stream));
    if (lds_slots) {
        auto kern = mfa_walk_kernel<K, REV, true>;
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)per_wave));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), per_wave, stream, g, d_bytes, d_offsets, n, d_results,
                           cx.d_counter, (uint32_t*)nullptr);
    } else {
        hipLaunchKernelGGL((mfa_walk_kernel<K, REV, false>), dim3((unsigned)grid), dim3(64), 0, stream, g, d_bytes,
                           d_offsets, n, d_results, cx.d_counter, cx.d_scratch);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_stop, stream));
    return MFA_OK;
}

template <int K>
static int launch_mfa_rev(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const uint8_t* b, const uint64_t* o, uint64_t n,
                          uint8_t* r, hipStream_t s) {
    return img.h.is_reversed ? launch_mfa_k<K, true>(img, ds, cx, b, o, n, r, s) : launch_mfa_k<K, false>(img, ds, cx, b, o, n, r, s);
}

int launch_mfa_walk(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                    uint64_t n, uint8_t* d_results, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (img.h.n_cells) {
        case 0:
        case 1: return launch_mfa_rev<1>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 2: return launch_mfa_rev<2>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 3: return launch_mfa_rev<3>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 4: return launch_mfa_rev<4>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 5: return launch_mfa_rev<5>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 6: return launch_mfa_rev<6>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 7: return launch_mfa_rev<7>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 8: return launch_mfa_rev<8>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
        case 9: return launch_mfa_rev<9>(img, ds, cx, d_bytes, d_offsets, n, d_results, s);
    }
    return MFA_ERR_UNSUPPORTED;
}

static bool make_packed(const HostImage& img, DfaPacked& pk) {
    if (img.dfa_states > 8 || img.n_classes > 8 || img.n_classes < 1) return false;
    std::memset(&pk, 0, sizeof pk);
    pk.n_lit = img.n_classes - 1;                      // tabulate_nfa: literal classes first, "every other byte" last
    for (uint32_t c = 0; c < pk.n_lit; c++) {
        int rep = -1;
        for (int b = 0; b < 256; b++)
            if (img.byte_class[b] == c) { if (rep >= 0) return false; rep = b; }
        if (rep < 0) return false;
        pk.lit[c] = (uint32_t)rep;
    }
    for (uint32_t c = 0; c < img.n_classes; c++)
        for (uint32_t s = 0; s < img.dfa_states; s++) pk.tab[c] |= (uint32_t)img.dfa_trans[s * img.n_classes + c] << (4 * s);
    for (uint32_t s = 0; s < img.dfa_states; s++) pk.accept_mask |= (uint32_t)(img.dfa_accept[s] != 0) << s;
    return true;
}

template <bool REV, bool PACKED, int NLIT>
static int launch_dfa_tiled(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const DfaPacked& pk, const uint8_t* d_bytes,
                            const uint64_t* d_offsets, uint64_t n, uint8_t* d_results, hipStream_t s) {
    size_t lds = 4 * 64 * kTileRow + (PACKED ? 0 : (size_t)img.dfa_states * kDfaRow * sizeof(uint16_t));
    uint64_t per_cu = (160u * 1024u) / lds;                 // resident blocks a CU's LDS allows (at most 8: 32 waves)
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    uint64_t blocks = (n + 255) / 256, cap = (uint64_t)ds.n_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    auto kern = dfa_tiled_kernel<REV, PACKED, NLIT>;
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_start, s));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, pk, ds.d_dfa_trans, ds.d_dfa_accept, ds.d_byte_class,
                       img.dfa_states, img.n_classes, d_bytes, d_offsets, n, d_results);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_stop, s));
    return MFA_OK;
}

int launch_dfa_walk(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                    uint64_t n, uint8_t* d_results, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)img.dfa_states * kDfaRow > 0xffffu) {                               // beyond 16-bit pre-multiplied states: table in L2
        uint64_t blocks = (n + 255) / 256, cap = (uint64_t)ds.n_cus * 8;
        if (blocks > cap) blocks = cap;
        if (blocks == 0) blocks = 1;
        HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_start, s));
        if (img.h.is_reversed)
            hipLaunchKernelGGL(dfa_big_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, ds.d_dfa_trans, ds.d_dfa_accept, ds.d_byte_class,
                               img.n_classes, d_bytes, d_offsets, n, d_results);
        else
            hipLaunchKernelGGL(dfa_big_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, ds.d_dfa_trans, ds.d_dfa_accept, ds.d_byte_class,
                               img.n_classes, d_bytes, d_offsets, n, d_results);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_stop, s));
        return MFA_OK;
    }
    {
        const char* mode = getenv("MFA_DFA_KERNEL");                                 // "simple" selects the untiled walk
        if (!(mode && mode[0] == 's')) {
            DfaPacked pk;
            // measured on MI355X, (a|b)*abb, 1M x 1 KiB: table in LDS 4.58 TB/s (3.25 when the packed form was measured: 2.65),
            // untiled 1.0 TB/s -- the LDS table is the default, "packed" selects the SGPR form
            const bool packed = mode && mode[0] == 'p' && make_packed(img, pk);
            if ((size_t)img.dfa_states * kDfaRow * 2 + 4 * 64 * kTileRow <= 64 * 1024) {
#define MFA_DFA_GO(REVV, P, NL) return launch_dfa_tiled<REVV, P, NL>(img, ds, cx, pk, d_bytes, d_offsets, n, d_results, s)
                if (packed && pk.n_lit <= 4) {
                    if (img.h.is_reversed) {
                        switch (pk.n_lit) { case 0: MFA_DFA_GO(true, true, 0); case 1: MFA_DFA_GO(true, true, 1); case 2: MFA_DFA_GO(true, true, 2);
                                            case 3: MFA_DFA_GO(true, true, 3); default: MFA_DFA_GO(true, true, 4); }
                    } else {
                        switch (pk.n_lit) { case 0: MFA_DFA_GO(false, true, 0); case 1: MFA_DFA_GO(false, true, 1); case 2: MFA_DFA_GO(false, true, 2);
                                            case 3: MFA_DFA_GO(false, true, 3); default: MFA_DFA_GO(false, true, 4); }
                    }
                }
                if (img.h.is_reversed) MFA_DFA_GO(true, false, 0);
                MFA_DFA_GO(false, false, 0);
#undef MFA_DFA_GO
            }
        }
    }
    size_t lds = (size_t)img.dfa_states * kDfaRow * sizeof(uint16_t);
    if (lds > 64 * 1024) return MFA_ERR_UNSUPPORTED;
    uint64_t blocks = (n + 255) / 256;
    uint64_t cap = (uint64_t)ds.n_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_start, s));
    if (img.h.is_reversed) {
        HIP_TRY(hipFuncSetAttribute((const void*)dfa_walk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dfa_walk_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, ds.d_dfa_trans, ds.d_dfa_accept,
                           ds.d_byte_class, img.dfa_states, img.n_classes, d_bytes, d_offsets, n, d_results);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)dfa_walk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dfa_walk_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, ds.d_dfa_trans, ds.d_dfa_accept,
                           ds.d_byte_class, img.dfa_states, img.n_classes, d_bytes, d_offsets, n, d_results);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_stop, s));
    return MFA_OK;
}

}  // namespace mfa
