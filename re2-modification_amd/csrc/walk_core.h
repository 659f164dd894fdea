// The live-list walk: MFA::match (reference mfa.cpp:215-236) for one string per lane, interpreted from tables.
//
// Included by walk.hip (the kernel, gfx950, 64 lanes per wave) and by tests/emul/walk_emul.cpp (the same source compiled for the
// host as a wave of ONE lane: a differential test of the step function and of the jump logic against the oracle that needs no
// GPU; test infrastructure only, the product path is the kernel).
//
// ---- state ------------------------------------------------------------------------------------------------------------
// mfa.cpp:206-211 evaluates the first state per node in set order and drops the rest, so what lives between two steps is one
// state per node at most, the minimum under (pos, first cell name, allocation time).  A lane keeps those states as a LIST of
// entries (lists are short: 2-8 entries on the README automata, whatever the node count), two lists per lane (the step reads
// one and builds the other) in LDS, [entry][word][lane] so that every lane owns a bank; entries beyond the LDS capacity C
// spill to a per-wave area in global memory.
//   entry = P | W1 | per cell: S, L                       (2 + 2K words)
//     P  = pos << 4 | name of the first cell present      (the set order's leading key, compared as one integer)
//     W1 = vid | tie << 16                                 vid = the state's vnode (walk_tables.h): node << vb | variant
//     S  = start | first byte << 24     L = len | flags << 24      the cell value is scan[start, start + len) (cells are spans:
//          every write appends the text just consumed, mfa.cpp:89-104); flags F_PRESENT | F_OPEN | F_READ | F_UNI
// Allocation time = creation order (copy_memory allocates a state's cells when the state is created, mfa.cpp:107-114).  Within
// one step the candidates for a node are created in this order: carried waiting states (they keep their old cells), then the
// children of states with pos == i in node order, then the children of waiting states in node order; within one source in
// edge order, depth first.  `tie` encodes the first two levels (0 carried, 1 + node, 1025 + node), the last is the order in
// which a source's effective edges are listed: a candidate replaces an entry only if (P, tie) is strictly smaller.
//
// ---- jumps ------------------------------------------------------------------------------------------------------------
// Inside a stretch of input that repeats with a short period (the region table of regions.hip) the list usually moves affinely
// from period to period.  The step exists in two instantiations, on plain values and on dual values (value, change per
// period; device_common.h): after two periods that moved the list by the same amounts a lane takes one period of dual steps,
// which yield the next list, where the step map sends the direction, and for how many periods every comparison made keeps its
// outcome; if the direction reproduced itself the lane adds (periods - 1) * direction and skips those steps.  Exact: a jump is
// only taken over steps whose every branch outcome is proven.
#ifndef MFA_WALK_CORE_H
#define MFA_WALK_CORE_H

#include "../../include/mfa_image_format.h"
// the wave-wide scans of device_common.h with ONE block per lane in flight: their second block cost this kernel 10 registers at its
// peak (the cell read sits in the innermost loop of the step) and bought no time
#define MFA_SCAN_DEPTH 1
#define MFA_RUN_DEPTH 1
#include "device_common.h"

#ifndef WALK_WV
#define WALK_WV 64u          /* lanes per wave = stride of the per-lane arrays */
#endif
#ifndef WALK_DEV
#define WALK_DEV __device__ __forceinline__
#endif
#ifndef WALK_WITH_DUAL
#define WALK_WITH_DUAL 1
#endif
#ifndef WALK_PROBE_PERIODS
#define WALK_PROBE_PERIODS 5u
#endif
#ifndef WALK_NODE_MAP
#define WALK_NODE_MAP 0      /* 1: the kernel for long lists (walk.hip: WALK_LONG_LISTS) finds a node's entry through a per-lane map in LDS */
#endif
#ifndef WALK_LEAN_MIN_LEN
#define WALK_LEAN_MIN_LEN 256u      /* shorter strings are not worth a second look */
#endif
#ifndef WALK_TAIL
#define WALK_TAIL 8u         /* a probe with period p starts only where 4 p + WALK_TAIL positions of the region are left */
#endif

namespace mfa_walk {

// vinfo bits (walk_tables.h)
#define VI_VALID 1u
#define VI_EPS   2u
#define VI_CACC  4u
#define VI_QUAL  8u
#define VI_HASC  (1u << 17)

#define TIE_CARRY 0u
#define TIE_HERE  1u
#define TIE_WAIT  1025u

// ---- wave primitives (one lane on the host) ---------------------------------------------------------------------------------
#ifdef MFA_HOST_EMUL
static unsigned long long g_ev[8];      // development counts: entries, edge evaluations, inserts, search iterations, -, -, waiting insertions
#define WALK_EV(k) (g_ev[k]++)
WALK_DEV uint32_t wv_lane() { return 0u; }
WALK_DEV uint64_t wv_shfl64(uint64_t v, int) { return v; }
WALK_DEV uint64_t wv_load_fresh(const uint64_t* p) { return *p; }
WALK_DEV void wv_nap() {}
WALK_DEV uint32_t wv_atomic_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
WALK_DEV uint32_t wv_bcast32(uint32_t v, int) { return v; }
// the wave-wide scans of device_common.h assume 64 lanes: scalar restatements for the one-lane wave
template <bool REV>
inline uint32_t coop_run_end_x(const uint8_t* bytes, uint64_t base, uint32_t len, uint32_t i0, uint32_t) {
    if (i0 + 1u >= len) return len;
    auto at = [&](uint32_t j) { return bytes[base + (REV ? (uint64_t)(len - 1u - j) : (uint64_t)j)]; };
    const uint8_t c = at(i0);
    uint32_t j = i0 + 1u;
    while (j < len && at(j) == c) j++;
    return j;
}
inline bool coop_mem_equal_x(const uint8_t* bytes, uint64_t pa, uint64_t pb, uint32_t l, uint32_t) {
    for (uint32_t k = 0; k < l; k++)
        if (bytes[pa + k] != bytes[pb + k]) return false;
    return true;
}
#else
#define WALK_EV(k) ((void)0)
WALK_DEV uint32_t wv_lane() { return threadIdx.x & 63u; }
// a load that does not take this CU's L1 copy for an answer (agent scope: `sc1`), and a pause between two polls
WALK_DEV uint64_t wv_load_fresh(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WALK_DEV void wv_nap() { __builtin_amdgcn_s_sleep(16); }
WALK_DEV uint32_t wv_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
WALK_DEV uint32_t wv_bcast32(uint32_t v, int L) { return (uint32_t)__shfl((int)v, L); }
WALK_DEV uint64_t wv_shfl64(uint64_t v, int L) { return ((uint64_t)__shfl((uint32_t)(v >> 32), L) << 32) | __shfl((uint32_t)v, L); }
// Out of line on purpose: a scan is rare (a run the region table does not hold, a long comparison outside known regions), sits in
// the innermost loop of the step, and inlined it raised the kernel's register count by 25 for every wave, scanning or not.
template <bool REV>
__device__ __attribute__((noinline)) uint32_t coop_run_end_x(const uint8_t* bytes, uint64_t base, uint32_t len, uint32_t i0, uint32_t lane) { return coop_run_end<REV>(bytes, base, len, i0, lane); }
__device__ __attribute__((noinline)) bool coop_mem_equal_x(const uint8_t* bytes, uint64_t pa, uint64_t pb, uint32_t l, uint32_t lane) { return coop_mem_equal(bytes, pa, pb, l, lane); }
#endif

// ---- address spaces ------------------------------------------------------------------------------------------------------------
// Pointers into LDS carry their address space in the type: through plain pointers the compiler emits FLAT instructions (which
// count in both wait counters and return out of order) for every list access.
#ifdef MFA_HOST_EMUL
#define WALK_LDS
#else
#define WALK_LDS __attribute__((address_space(3)))
#endif
typedef const WALK_LDS uint32_t* TablePtr;      // tables in LDS (the usual case)
typedef const uint32_t* TablePtrG;               // tables too large for LDS: read from global memory (L1 / L2)

// ---- a lane's input: its string, the byte window, what it knows about runs and periodic regions ----------------------------------
// (device_common.h's Input, without per-lane pointers: what is the same for every lane of the wave stays in scalar registers)
struct WIn {
    const uint8_t*  bytes;       // wave-uniform: the whole batch
    uint64_t        total16;     //   its size rounded up to 16: loads stay below this offset
    const uint64_t* regions;     //   region table of the launch (regions.hip), nullptr = none
    uint32_t gate;               //   0, or 0x1000 | the stamp every word of this call's table rows carries (bits 52..63): the table is being written
                                 //   while this kernel runs (regions.hip: GATE), a word without the stamp has not arrived yet or is a stale copy
    WALK_LDS uint64_t* rtc;      //   the wave's LDS copy of every lane's first MFA_RT_CACHED table entries, [entry][lane]
    uint64_t base;               // offset of this lane's string
    uint32_t len, sid;
    uint64_t blk;                // offset of the 16 bytes in w0..w3 (any alignment), MFA_NO_WINDOW = none
    uint32_t w0, w1, w2, w3;
    uint32_t run_lo, run_hi, run_ch;       // scan[run_lo, run_hi) == run_ch, maximal to the right
    uint32_t per_lo, per_hi, per_q;        // scan[j] == scan[j + per_q] for per_lo <= j < per_hi - per_q (0: none)
    uint32_t prev_lo, prev_hi, prev_q;     // the region known before that one
    uint32_t dual_p;                       // steps per period of the dual step in flight, 0 in plain steps
    uint32_t rt_cnt;                       // entries of this string's table row
    // what the step could not decide without a wave-wide scan (the main loop scans between steps and runs the step again):
    uint32_t rq;                           // 0 none, 1 the end of the run of byte rq_b at scan index rq_a, 2 scan[rq_a, +rq_l) == scan[rq_b, +rq_l)?
    uint32_t rq_a, rq_b, rq_l;
    uint32_t cq_n;                         // comparisons answered for the step being executed (Store::gq)
};

WALK_DEV void w_drop_window(WIn& in) { in.blk = MFA_NO_WINDOW; }
WALK_DEV void w_reset(WIn& in, uint64_t base, uint32_t len, uint32_t sid) {
    in.base = base; in.len = len; in.sid = sid;
    w_drop_window(in);
    in.run_lo = in.run_hi = 0; in.run_ch = 0x100u;
    in.per_lo = in.per_hi = 0; in.per_q = 0; in.dual_p = 0;
    in.prev_lo = in.prev_hi = 0; in.prev_q = 0;
    in.rt_cnt = 0; in.rq = 0; in.rq_a = in.rq_b = in.rq_l = 0; in.cq_n = 0;
}

// Gated launches (in.gate != 0): the row's header and its first MFA_RT_CACHED entries must carry the call's stamp.  A lane whose row does
// not (yet) polls it -- fresh loads, a pause between two -- and gives up after ~50 ms: it then walks without a table (slower, never wrong).
WALK_DEV bool w_rt_stamped(const WIn& in, const uint4 a, const uint4 b) {
    const uint32_t want = in.gate & 0xfffu, cnt = a.x & 0xffu;
    return (a.y >> 20) == want && (cnt < 1u || (a.w >> 20) == want) && (cnt < 2u || (b.y >> 20) == want);
}
WALK_DEV void w_rt_await(const WIn& in, bool mine, uint64_t sid, uint4& a, uint4& b) {
    if (in.gate == 0u || in.regions == nullptr) return;
    bool ok = !mine || w_rt_stamped(in, a, b);
    for (uint32_t tries = 0; __any(!ok) && tries < 50000u; tries++) {
        if (!ok) {
            const uint64_t* t = in.regions + sid * MFA_RT_WORDS;
            const uint64_t w0 = wv_load_fresh(t), w1 = wv_load_fresh(t + 1), w2 = wv_load_fresh(t + 2);
            a = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
            b = make_uint4((uint32_t)w2, (uint32_t)(w2 >> 32), 0u, 0u);
            ok = w_rt_stamped(in, a, b);
        }
        if (__any(!ok)) wv_nap();
    }
    if (!ok) a = b = make_uint4(0, 0, 0, 0);
}

// The table row of the lane's string: header and first MFA_RT_CACHED entries arrive in (a, b) (two 16-byte loads that go out
// with the loads of the string's offsets); they are kept in LDS, and the input around both ends of those regions and at the end
// of the string is touched: that is where the lane needs bytes next (a jump ends near the end of its region).
WALK_DEV void w_rt_attach(WIn& in, uint32_t& warm, const uint4 a, const uint4 b) {
    if (in.regions == nullptr) return;
    const uint32_t lane = wv_lane();
    in.rt_cnt = a.x & 0xffu;
    const uint64_t e0 = ((uint64_t)a.w << 32) | a.z, e1 = ((uint64_t)b.y << 32) | b.x;
    in.rtc[lane] = e0; in.rtc[WALK_WV + lane] = e1;
    const uint64_t lim = in.total16 - 4u;
    auto touch = [&](uint64_t off) {
        off = off < lim ? off : lim;
        warm ^= *reinterpret_cast<const uint32_t*>(in.bytes + (off & ~(uint64_t)3));
    };
    if (in.len > 64u) touch(in.base + in.len - 4u);
    const uint64_t es[MFA_RT_CACHED] = {e0, e1};
#pragma unroll
    for (uint32_t k = 0; k < MFA_RT_CACHED; k++)
        if (k < in.rt_cnt) {
            const uint32_t mlo = (uint32_t)es[k] & 0x00ffffffu, mhi = (uint32_t)(es[k] >> 24) & 0x00ffffffu;
            touch(in.base + mhi); touch(in.base + (mhi >= 32u ? mhi - 32u : 0u));
            touch(in.base + mlo); touch(in.base + mlo + 32u);
        }
}
// entry e in scan coordinates
template <bool REV>
WALK_DEV void w_rt_entry(const WIn& in, uint32_t e, uint32_t& lo, uint32_t& hi, uint32_t& q) {
    uint64_t w;
    if (e < MFA_RT_CACHED) w = in.rtc[e * WALK_WV + wv_lane()];
    else if (in.gate == 0u) w = in.regions[(uint64_t)in.sid * MFA_RT_WORDS + 1u + e];
    else {                                                       // past this CU's L1 (which may hold the row an earlier call wrote), and only with the call's stamp
        w = wv_load_fresh(in.regions + (uint64_t)in.sid * MFA_RT_WORDS + 1u + e);
        if ((uint32_t)(w >> 52) != (in.gate & 0xfffu)) w = 0ull;      // (not there: the lane walks that stretch step by step)
    }
    const uint32_t mlo = (uint32_t)w & 0x00ffffffu, mhi = (uint32_t)(w >> 24) & 0x00ffffffu;
    q = (uint32_t)(w >> 48) & 15u;
    lo = REV ? in.len - mhi : mlo;
    hi = REV ? in.len - mlo : mhi;
}
// the region with the smallest period that contains scan index i and leaves room for a probe; next = the nearest region start behind i
template <bool REV>
WALK_DEV bool w_rt_find(const WIn& in, uint32_t i, uint32_t& lo, uint32_t& hi, uint32_t& q, uint32_t& next) {
    bool found = false;
    next = ~0u; lo = hi = q = 0u;
    for (uint32_t e = 0; e < in.rt_cnt; e++) {
        uint32_t l, h, qq;
        w_rt_entry<REV>(in, e, l, h, qq);
        if (l <= i && i < h) {
            if (h - i >= 4u * qq + WALK_TAIL && (!found || qq < q)) { found = true; lo = l; hi = h; q = qq; }
        } else if (l > i && l < next) next = l;
    }
    return found;
}
// exclusive end of the run of equal bytes that contains scan index i, if the table has it (q = 1 entries are maximal runs)
template <bool REV>
WALK_DEV bool w_rt_run(const WIn& in, uint32_t i, uint32_t& hi) {
    for (uint32_t e = 0; e < in.rt_cnt; e++) {
        uint32_t l, h, qq;
        w_rt_entry<REV>(in, e, l, h, qq);
        if (qq == 1u && l <= i && i < h) { hi = h; return true; }
    }
    return false;
}
template <bool REV> WALK_DEV uint64_t w_scan_addr(const WIn& in, uint32_t j) { return in.base + (REV ? (uint64_t)(in.len - 1u - j) : (uint64_t)j); }
// address of the 16-byte window whose FIRST byte in scan order is scan index j; kept inside [0, total16)
template <bool REV> WALK_DEV uint64_t w_window_addr(const WIn& in, uint32_t j) {
    if (REV) { const uint64_t e = in.base + in.len; return e >= (uint64_t)j + 16u ? e - j - 16u : 0; }
    const uint64_t a = in.base + j;
    return a + 16u <= in.total16 ? a : in.total16 - 16u;
}
// the byte at scan index i (i < len).  A lane keeps 16 bytes of its string in registers (any alignment); it loads the next 16
// when it runs off them, or lands somewhere else after a jump: one load per 16 steps, waited for on the spot (a step of this
// kernel is several microseconds: a second window requested ahead, as the generated kernels keep one, bought nothing here and
// cost five registers)
template <bool REV> WALK_DEV uint32_t w_stream_byte(WIn& in, uint32_t i) {
    const uint64_t addr = w_scan_addr<REV>(in, i);
    uint64_t o = addr - in.blk;
    if (o >= 16u) {
        const uint64_t a = w_window_addr<REV>(in, i);
        const uint4 d = load16u(in.bytes, a);
        in.w0 = d.x; in.w1 = d.y; in.w2 = d.z; in.w3 = d.w; in.blk = a;
        o = addr - a;
    }
    const uint32_t ob = (uint32_t)o;
    const uint32_t lo = (ob & 4u) ? in.w1 : in.w0;
    const uint32_t hi = (ob & 4u) ? in.w3 : in.w2;
    const uint32_t w = (ob & 8u) ? hi : lo;
    return (w >> ((ob & 3u) * 8u)) & 0xffu;
}
// exclusive end of the run of byte c that starts at scan index i, looking at no more than `limit` positions (false: it goes on)
template <bool REV> WALK_DEV bool w_run_end_bounded(const WIn& in, uint32_t i, uint32_t c, uint32_t limit, uint32_t& end) {
    if (i + 1u >= in.len) { end = in.len; return true; }
    if (!REV) {
        const uint64_t p = in.base + i + 1u, e = in.base + in.len, e2 = e - p > limit ? p + limit : e;
        const uint64_t q = first_not_equal(in.bytes, p, e2, c);
        if (q < e2 || e2 == e) { end = (uint32_t)(q - in.base); return true; }
        return false;
    }
    const int64_t lo = (int64_t)in.base, hi = (int64_t)(in.base + in.len - 2u - i), lo2 = hi - lo >= (int64_t)limit ? hi - (int64_t)limit + 1 : lo;
    const int64_t q = last_not_equal(in.bytes, lo2, hi, c);
    if (q >= lo2) { end = (uint32_t)((int64_t)in.len - 1 - (q - lo)); return true; }
    if (lo2 == lo) { end = in.len; return true; }
    return false;
}
// equality of scan[a, a+l) and scan[b, b+l), lane by lane (short spans only)
template <bool REV> WALK_DEV bool w_spans_equal(const WIn& in, uint32_t a, uint32_t b, uint32_t l) {
    const uint8_t* pa = in.bytes + (REV ? in.base + (in.len - a - l) : in.base + a);
    const uint8_t* pb = in.bytes + (REV ? in.base + (in.len - b - l) : in.base + b);
    uint32_t k = 0;
    for (; k + 8 <= l; k += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, pa + k, 8);
        __builtin_memcpy(&y, pb + k, 8);
        if (x != y) return false;
    }
    for (; k < l; k++)
        if (pa[k] != pb[k]) return false;
    return true;
}
// end of the run of equal bytes that contains i as a value of type U: in a dual step over a region with period > 1 the run structure
// repeats every period, the end moves with i
WALK_DEV uint32_t w_run_hi_u(const WIn& in, uint32_t, tb_t&) { return in.run_hi; }
WALK_DEV Dual w_run_hi_u(const WIn& in, Dual, tb_t& TB) {
    if (in.per_q <= 1u || in.dual_p == 0u) return Dual{in.run_hi, 0};
    Dual r{in.run_hi, (int32_t)in.dual_p};
    (void)lt(r, Dual{in.per_hi, 0}, TB);                 // exact only while the shifted run ends inside the periodic region
    return r;
}
// byte-wise comparison of a cell value with the text at i: is its outcome the same in every period?
WALK_DEV void w_spans_period_bound(const WIn&, uint32_t, uint32_t, uint32_t, tb_t&) {}
WALK_DEV void w_spans_period_bound(const WIn& in, Dual i, Dual start, Dual l, tb_t& TB) {
    const bool ok = in.dual_p != 0u && in.per_q != 0u && l.d == 0 && start.d >= 0 && (uint32_t)start.d % in.per_q == 0u &&
                    start.v >= in.per_lo && i.v >= in.per_lo;
    if (!ok) { tb_min(TB, 1); return; }
    (void)le(add(i, l), Dual{in.per_hi, 0}, TB);         // both sides stay inside the periodic region: the same bytes every period
    if (start.d != 0) (void)le(add(start, l), Dual{in.per_hi, 0}, TB);
}
WALK_DEV bool w_span_in_region(const WIn& in, uint32_t s, uint32_t l, uint32_t& q) {
    if (in.per_q != 0u && s >= in.per_lo && s + l <= in.per_hi) { q = in.per_q; return true; }
    if (in.prev_q != 0u && s >= in.prev_lo && s + l <= in.prev_hi) { q = in.prev_q; return true; }
    return false;
}
// would the cell read have to look for the end of the run of bytes at i?  (then the whole wave finds it first)
WALK_DEV bool w_uni_needs_run(const WIn& in, uint32_t i, uint32_t ch, uint32_t l, uint32_t fl) {
    if (!(fl & F_UNI) || l <= 1u || in.len - i < l) return false;
    const uint32_t c = (fl >> 8) & 0xffu;
    return c == ch && !(in.run_ch == c && in.run_lo <= i && i < in.run_hi);
}
// A cell read without its byte-wise comparison: the outcome, or need_cmp when scan[i, i+l) has to be compared with
// scan[start, start+l) byte by byte (the caller does that with the whole wave).  mfa.cpp:178-187
template <bool REV, class U>
WALK_DEV bool w_read_pre(WIn& in, U i, uint32_t ch, U start, U l, uint32_t fl, tb_t& TB, bool& need_cmp) {
    if (lt(sub(konst<U>(in.len), i), l, TB)) return false;
    if (eq(l, konst<U>(0u), TB)) return true;
    if (fl & F_UNI) {                                    // the value is one byte repeated: compare its length with the run of bytes at i
        const uint32_t c = (fl >> 8) & 0xffu;
        if (c != ch) return false;
        if (eq(l, konst<U>(1u), TB)) return true;
        // (the caller has made sure the run at i is known: w_uni_needs_run)
        return ge(sub(w_run_hi_u(in, i, TB), i), l, TB);
    }
    if (((fl >> 8) & 0xffu) != ch) return false;          // bits 8..15 of the flags hold the FIRST byte of a non-empty value
    w_spans_period_bound(in, i, start, l, TB);
    const uint32_t lv = val(l), head = lv < 16u ? lv : 16u;
    if (!w_spans_equal<REV>(in, REV ? val(start) + (lv - head) : val(start), REV ? val(i) + (lv - head) : val(i), head)) return false;
    if (lv <= 16u) return true;
    // two q-periodic spans (q <= 8) whose first 16 bytes agree are equal.  Plain steps only (a lane in its dual period has to
    // bound the outcome over the periods to come: w_spans_period_bound above).
    if (in.dual_p == 0u) {
        uint32_t qa = 0, qb = 0;
        if (w_span_in_region(in, val(start), lv, qa) && w_span_in_region(in, val(i), lv, qb) && qa == qb) return true;
    }
    need_cmp = true;
    return false;
}

// ---- per-lane view of an automaton's tables ------------------------------------------------------------------------------------
struct Aut {                 // two registers per lane: where the automaton's table block starts, and its dimensions
    uint32_t at, dims;       // dims = classes | variant bits << 8 | vnodes << 12   (at most 255 classes, 13 variant bits, 2^13 vnodes)
    WALK_DEV uint32_t nc() const { return dims & 0xffu; }
    WALK_DEV uint32_t vbits() const { return (dims >> 8) & 0xfu; }
    WALK_DEV uint32_t nv() const { return dims >> 12; }
    // the block's layout is fixed (walk_tables.cpp): header, class map, vinfo[nv], vc[nv], vb[nv][nc], effective edges
    WALK_DEV uint32_t cmap() const { return at + 16u; }
    WALK_DEV uint32_t vinfo() const { return at + 80u; }
    WALK_DEV uint32_t vc() const { return at + 80u + nv(); }
    WALK_DEV uint32_t vb() const { return at + 80u + 2u * nv(); }
    WALK_DEV uint32_t ee() const { return at + 80u + nv() * (2u + nc()); }
};
template <class TP> WALK_DEV void aut_load(Aut& a, TP T, uint32_t at) {
    a.at = at; a.dims = T[at + 2] | (T[at + 1] << 8) | (T[at] << 12);
}
template <class TP> WALK_DEV uint32_t aut_start(TP T, const Aut& a) { return T[a.at + 4]; }

// ---- list storage -----------------------------------------------------------------------------------------------------------------
constexpr uint32_t CMP_CACHE = 4;      // answered comparisons a lane can hold for one step (beyond them it compares by itself)
template <int K> struct Lay {
    static constexpr uint32_t W = 2 + 2 * K;            // value words per entry
    static constexpr uint32_t DW = (1 + 2 * K + 1) / 2;  // direction words per entry: int16 each (pos, then S, L per cell)
    static constexpr uint32_t EEW = K <= 6 ? 2 : 3;      // words per effective edge
};

struct Store {               // wave-uniform bases; every access adds the lane
    WALK_LDS uint32_t* lv;   // LDS  [2][C][W][WV]    list values
    WALK_LDS uint32_t* ld;   // LDS  [2][C][DW][WV]   list directions (dual steps)
    WALK_LDS uint32_t* sb;   // LDS  [C][W][WV]       the list one period ago
    WALK_LDS uint32_t* sa;   // LDS  [C][DW][WV]      the movement over the last period
    uint32_t* gv;            // global [2][CX][W][WV]  entries C, C+1, ... of either list
    uint32_t* gd;            // global [2][CX][DW][WV]
    uint32_t* gsb;           // global [CX][W][WV]
    uint32_t* gsa;           // global [CX][DW][WV]
    uint32_t* gq;            // global [CMP_CACHE][4][WV]  long comparisons answered for the step in progress: a, b, l, equal
    uint32_t C, CX;
    uint32_t CI;             // entries of the two probe images that are in LDS (C, or 0: the images live in global memory)
    // WALK_NODE_MAP: where in the list being built does node x have its entry?  One byte per node and lane in LDS, [node / 4][lane][node % 4] (a
    // lane's column is cleared with one word per four nodes): 0xff = nowhere, else position | list << 7.  A position of the list being READ
    // is stale the moment its entry has been read (the entry clears it then), so "nowhere" and "in the other list" are the same answer.
    WALK_LDS uint8_t* nm;
    uint32_t nm_words;       // words of a lane's column: (nodes of the launch's largest automaton + 3) / 4
};
WALK_DEV uint32_t nm_at(uint32_t node) { return ((node >> 2) * WALK_WV + wv_lane()) * 4u + (node & 3u); }

// single words (the images of a probe use them: rare)
template <int K> WALK_DEV uint32_t rd_v(const Store& st, uint32_t l, uint32_t e, uint32_t w) {
    return e < st.C ? st.lv[((l * st.C + e) * Lay<K>::W + w) * WALK_WV + wv_lane()] : st.gv[((l * st.CX + (e - st.C)) * Lay<K>::W + w) * WALK_WV + wv_lane()];
}
template <int K> WALK_DEV void wr_v(const Store& st, uint32_t l, uint32_t e, uint32_t w, uint32_t v) {
    if (e < st.C) st.lv[((l * st.C + e) * Lay<K>::W + w) * WALK_WV + wv_lane()] = v; else st.gv[((l * st.CX + (e - st.C)) * Lay<K>::W + w) * WALK_WV + wv_lane()] = v;
}
template <int K> WALK_DEV uint32_t rd_d(const Store& st, uint32_t l, uint32_t e, uint32_t w) {
    return e < st.C ? st.ld[((l * st.C + e) * Lay<K>::DW + w) * WALK_WV + wv_lane()] : st.gd[((l * st.CX + (e - st.C)) * Lay<K>::DW + w) * WALK_WV + wv_lane()];
}
template <int K> WALK_DEV void wr_d(const Store& st, uint32_t l, uint32_t e, uint32_t w, uint32_t v) {
    if (e < st.C) st.ld[((l * st.C + e) * Lay<K>::DW + w) * WALK_WV + wv_lane()] = v; else st.gd[((l * st.CX + (e - st.C)) * Lay<K>::DW + w) * WALK_WV + wv_lane()] = v;
}

// whole entries: ONE branch on where the entry lives, then W (or DW) accesses with constant offsets from one address
template <int K> WALK_DEV void rd_words(const Store& st, uint32_t l, uint32_t e, uint32_t (&w)[Lay<K>::W]) {
    if (e < st.C) {
        const WALK_LDS uint32_t* p = st.lv + ((l * st.C + e) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) w[k] = p[k * WALK_WV];
    } else {
        const uint32_t* p = st.gv + (size_t)((l * st.CX + (e - st.C)) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) w[k] = p[k * WALK_WV];
    }
}
template <int K> WALK_DEV void wr_words(const Store& st, uint32_t l, uint32_t e, const uint32_t (&w)[Lay<K>::W]) {
    if (e < st.C) {
        WALK_LDS uint32_t* p = st.lv + ((l * st.C + e) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) p[k * WALK_WV] = w[k];
    } else {
        uint32_t* p = st.gv + (size_t)((l * st.CX + (e - st.C)) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) p[k * WALK_WV] = w[k];
    }
}
template <int K> WALK_DEV void rd_dwords(const Store& st, uint32_t l, uint32_t e, uint32_t (&w)[Lay<K>::DW]) {
    if (e < st.C) {
        const WALK_LDS uint32_t* p = st.ld + ((l * st.C + e) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = p[k * WALK_WV];
    } else {
        const uint32_t* p = st.gd + (size_t)((l * st.CX + (e - st.C)) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = p[k * WALK_WV];
    }
}
template <int K> WALK_DEV void wr_dwords(const Store& st, uint32_t l, uint32_t e, const uint32_t (&w)[Lay<K>::DW]) {
    if (e < st.C) {
        WALK_LDS uint32_t* p = st.ld + ((l * st.C + e) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) p[k * WALK_WV] = w[k];
    } else {
        uint32_t* p = st.gd + (size_t)((l * st.CX + (e - st.C)) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) p[k * WALK_WV] = w[k];
    }
}

// an entry in registers
template <class U, int K> struct Ent { U P; uint32_t vid; U S[K], L[K]; uint32_t F[K]; };

WALK_DEV int32_t d16(uint32_t w, uint32_t k) { return (int32_t)(int16_t)(uint16_t)(w >> (16u * (k & 1u))); }
WALK_DEV void setv(uint32_t& x, uint32_t v) { x = v; }
WALK_DEV void setv(Dual& x, uint32_t v) { x.v = v; x.d = 0; }

template <int K> WALK_DEV void load_dirs(const Store&, uint32_t, uint32_t, bool, Ent<uint32_t, K>&) {}
template <int K> WALK_DEV void load_dirs(const Store& st, uint32_t l, uint32_t e, bool want, Ent<Dual, K>& x) {
    uint32_t w[Lay<K>::DW];
#pragma unroll
    for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = 0u;
    if (want) rd_dwords<K>(st, l, e, w);
    x.P.d = d16(w[0], 0) * 16;
#pragma unroll
    for (int c = 0; c < K; c++) { x.S[c].d = d16(w[(1 + 2 * c) / 2], 1 + 2 * c); x.L[c].d = d16(w[(2 + 2 * c) / 2], 2 + 2 * c); }
}
// direction of the key of an entry only
template <int K> WALK_DEV void load_pdir(const Store&, uint32_t, uint32_t, bool, uint32_t&) {}
template <int K> WALK_DEV void load_pdir(const Store& st, uint32_t l, uint32_t e, bool want, Dual& P) { P.d = want ? d16(rd_d<K>(st, l, e, 0), 0) * 16 : 0; }

// want_d: the lane's directions are meaningful (it is in its dual period); other lanes carry direction 0
template <class U, int K> WALK_DEV void load_entry(const Store& st, uint32_t l, uint32_t e, bool want_d, Ent<U, K>& x) {
    uint32_t w[Lay<K>::W];
    rd_words<K>(st, l, e, w);
    setv(x.P, w[0]);
    x.vid = w[1] & 0xffffu;
#pragma unroll
    for (int c = 0; c < K; c++) {
        setv(x.S[c], w[2 + 2 * c] & 0x00ffffffu);
        setv(x.L[c], w[3 + 2 * c] & 0x00ffffffu);
        x.F[c] = ((w[3 + 2 * c] >> 24) & 0xfu) | ((w[2 + 2 * c] >> 24) << 8);
    }
    load_dirs<K>(st, l, e, want_d, x);
}

template <int K> WALK_DEV void store_dirs(const Store&, uint32_t, uint32_t, const Ent<uint32_t, K>&, bool&) {}
template <int K> WALK_DEV void store_dirs(const Store& st, uint32_t l, uint32_t e, const Ent<Dual, K>& x, bool& fits) {
    int32_t d[2 * Lay<K>::DW];
#pragma unroll
    for (uint32_t k = 0; k < 2 * Lay<K>::DW; k++) d[k] = 0;
    d[0] = x.P.d / 16;
    bool ok = (x.P.d & 15) == 0;
#pragma unroll
    for (int c = 0; c < K; c++) { d[1 + 2 * c] = x.S[c].d; d[2 + 2 * c] = x.L[c].d; }
#pragma unroll
    for (uint32_t k = 0; k < 2 * Lay<K>::DW; k++) ok = ok && d[k] == (int32_t)(int16_t)d[k];
    if (!ok) fits = false;
    uint32_t w[Lay<K>::DW];
#pragma unroll
    for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = ((uint32_t)d[2 * k] & 0xffffu) | ((uint32_t)d[2 * k + 1] << 16);
    wr_dwords<K>(st, l, e, w);
}

template <class U, int K> WALK_DEV void store_entry(const Store& st, uint32_t l, uint32_t e, const Ent<U, K>& x, uint32_t tie, bool& fits) {
    uint32_t w[Lay<K>::W];
    w[0] = val(x.P);
    w[1] = x.vid | (tie << 16);
#pragma unroll
    for (int c = 0; c < K; c++) {
        w[2 + 2 * c] = (val(x.S[c]) & 0x00ffffffu) | ((x.F[c] >> 8) << 24);
        w[3 + 2 * c] = (val(x.L[c]) & 0x00ffffffu) | ((x.F[c] & 0xfu) << 24);
    }
    wr_words<K>(st, l, e, w);
    store_dirs<K>(st, l, e, x, fits);
}

// ---- memory actions (MFA::doMemoryWriteActions, mfa.cpp:80-105) on one cell; the text just consumed is scan[ts, ts + tl) ------------
template <class U> WALK_DEV void cell_open(U& S, U& L, uint32_t& F, U ts, U tl, bool tuni, uint32_t tch) {
    S = ts; L = tl; F = F_PRESENT | F_OPEN | (tuni ? F_UNI : 0u) | (tch << 8);               // create if absent, open(), write(t)
}
template <class U> WALK_DEV void cell_write(U& S, U& L, uint32_t& F, U ts, U tl, bool tuni, uint32_t tch, tb_t& TB) {
    const uint32_t f = F;                                                                      // write(t) when open
    const bool w = (f & (F_PRESENT | F_OPEN)) == (F_PRESENT | F_OPEN) && !eq(tl, konst<U>(0u), TB);
    if (!w) return;
    const bool was_empty = eq(L, konst<U>(0u), TB);
    const uint32_t fch = (f >> 8) & 0xffu;
    const bool uni = was_empty ? tuni : ((f & F_UNI) && tuni && fch == tch);
    S = sel(was_empty, ts, S);
    L = add(L, tl);
    F = (f & (F_PRESENT | F_OPEN | F_READ)) | (uni ? F_UNI : 0u) | ((was_empty ? tch : fch) << 8);
}

// the actions of an edge on the lanes in `pred` (t is a private copy of those lanes' cells)
template <class U, int K>
WALK_DEV void apply_actions(Ent<U, K>& t, uint32_t actions, bool pred, U ts, U tl, bool tuni, uint32_t tch, tb_t& TB) {
#pragma unroll
    for (int c = 0; c < K; c++) {
        const uint32_t a = (actions >> (2 * c)) & 3u;
        if (pred) {
            if (a == MFA_ACT_OPEN) cell_open<U>(t.S[c], t.L[c], t.F[c], ts, tl, tuni, tch);
            else if (a == MFA_ACT_CLOSE) t.F[c] &= ~F_OPEN;                                  // close(); an absent cell stays absent
            else cell_write<U>(t.S[c], t.L[c], t.F[c], ts, tl, tuni, tch, TB);
        }
    }
}

// ---- a cell read (mfa.cpp:177-191) on the lanes in `rd`: does scan[i, i + |v|) equal the value? ----------------------------------------
// Run extents and byte-wise comparisons are done by the whole wave, one lane's at a time (device_common.h).

// The exclusive end of the run of byte ch that contains scan index i, for the lanes in `nr`: from the region table (q = 1 entries are
// maximal runs), else measured -- a short stretch lane by lane, the rest by the whole wave.  Out of line on purpose: it is needed when
// a lane enters a new run (not every step), sits in the innermost loop of the step, and inlined it raised the kernel's register count
// by 25 for every wave.  scanned: the end comes from a measurement (the caller then also knows a periodic region).
struct RunEnd { uint32_t end; bool scanned; };
#ifdef MFA_HOST_EMUL
template <bool REV> inline
#else
template <bool REV> __device__ __attribute__((noinline))
#endif
RunEnd run_end_for(const uint8_t* bytes, uint64_t total16, const uint64_t* regions, uint32_t gate, WALK_LDS uint64_t* rtc, uint64_t base, uint32_t len, uint32_t sid,
                   uint32_t rt_cnt, bool nr, uint32_t i, uint32_t ch) {
    WIn in;
    in.bytes = bytes; in.total16 = total16; in.regions = regions; in.gate = gate; in.rtc = rtc; in.base = base; in.len = len; in.sid = sid; in.rt_cnt = rt_cnt;
    RunEnd out{0u, false};
    const uint32_t lane = wv_lane();
    if (nr && regions != nullptr) {
        uint32_t rh;
        if (w_rt_run<REV>(in, i, rh) || w_run_end_bounded<REV>(in, i, ch, 192u, rh)) { out.end = rh; nr = false; }
    }
    for (unsigned long long sb = __ballot(nr); sb; sb &= sb - 1ull) {
        const int L = __builtin_ctzll(sb);
        const uint32_t r = coop_run_end_x<REV>(bytes, wv_shfl64(base, L), __shfl(len, L), __shfl(i, L), lane);
        if (lane == (uint32_t)L) { out.end = r; out.scanned = true; }
    }
    return out;
}

// The end of the run of byte ch at scan index i, if the 16 bytes the lane holds show it (no load): most short runs do end there.
template <bool REV> WALK_DEV bool w_run_end_in_window(const WIn& in, uint32_t i, uint32_t ch, uint32_t& end) {
    const uint64_t o64 = w_scan_addr<REV>(in, i) - in.blk;
    if (o64 >= 16u) return false;
    const uint32_t o = (uint32_t)o64;
    const uint32_t m = mismatch_mask16(make_uint4(in.w0, in.w1, in.w2, in.w3), ch);      // bit k: byte k of the window is not ch
    if (!REV) {
        const uint32_t after = m >> (o + 1u);                      // scan indices i+1, i+2, ... = window bytes o+1, o+2, ...
        if (after) { const uint32_t e = i + 1u + (uint32_t)__builtin_ctz(after); end = e < in.len ? e : in.len; return true; }
        if (i + (16u - o) >= in.len) { end = in.len; return true; }      // the window reaches the end of the string
        return false;
    }
    const uint32_t below = m & ((1u << o) - 1u);                   // scan indices i+1, i+2, ... = window bytes o-1, o-2, ...
    if (below) { const uint32_t e = i + (o - (31u - (uint32_t)__builtin_clz(below))); end = e < in.len ? e : in.len; return true; }
    if (i + 1u + o >= in.len) { end = in.len; return true; }
    return false;
}
// ... or the 16 bytes behind them (one load: runs of up to 17-32 bytes; longer ones are asked for, see cell_read)
template <bool REV> WALK_DEV bool w_run_end_next_block(const WIn& in, uint32_t i, uint32_t ch, uint32_t& end) {
    const uint64_t o64 = w_scan_addr<REV>(in, i) - in.blk;
    if (o64 >= 16u) return false;
    const uint32_t o = (uint32_t)o64;
    if (!REV) {
        const uint64_t a2 = in.blk + 16u;
        if (a2 + 16u > in.total16) return false;
        const uint32_t m = mismatch_mask16(load16u(in.bytes, a2), ch), first = i + (16u - o);      // scan index of the block's byte 0
        if (m) { const uint32_t e = first + (uint32_t)__builtin_ctz(m); end = e < in.len ? e : in.len; return true; }
        if (first + 16u >= in.len) { end = in.len; return true; }
        return false;
    }
    if (in.blk < 16u) return false;
    const uint32_t m = mismatch_mask16(load16u(in.bytes, in.blk - 16u), ch);                        // byte k = scan index i + o + 16 - k
    if (m) { const uint32_t e = i + o + 16u - (31u - (uint32_t)__builtin_clz(m)); end = e < in.len ? e : in.len; return true; }
    if (i + o + 17u >= in.len) { end = in.len; return true; }
    return false;
}

// The step itself scans nothing: a read that needs the end of a run the lane does not know, or a byte-wise comparison of long
// spans, leaves a REQUEST in the lane's input state and counts as failed; the main loop answers the requests between steps (with
// the whole wave, at a point where few registers are live) and executes the step again -- the step is a function of the list, the
// byte and what the lane knows, so the second execution is the real one.  (Inlined into the step, the scans raised the kernel's
// register count by 25-40 for every wave, scanning or not.)
template <bool REV, class U>
WALK_DEV bool cell_read(const Store& st, WIn& in, bool rd, U i, uint32_t ch, U vs, U vl, uint32_t vf, tb_t& TB) {
    bool nr = rd && w_uni_needs_run(in, val(i), ch, val(vl), vf);
    if (nr) {
        uint32_t e;
        if (w_run_end_in_window<REV>(in, val(i), ch, e) || w_run_end_next_block<REV>(in, val(i), ch, e)) { in.run_lo = val(i); in.run_hi = e; in.run_ch = ch; nr = false; }
        else if (in.rq == 0u) { in.rq = 1u; in.rq_a = val(i); in.rq_b = ch; }
    }
    bool ok = false, cmp = false;
    if (rd && !nr) ok = w_read_pre<REV, U>(in, i, ch, vs, vl, vf, TB, cmp);
    if (__any(cmp)) {
        if (cmp) {
            const uint32_t ca = val(vs), cb = val(i), cl = val(vl);
            bool known = false;
            for (uint32_t k = 0; k < in.cq_n; k++) {
                uint32_t* q = st.gq + (size_t)(k * 4u) * WALK_WV + wv_lane();
                if (q[0] == ca && q[WALK_WV] == cb && q[2 * WALK_WV] == cl) { ok = q[3 * WALK_WV] != 0u; known = true; }
            }
            if (!known) {
                if (in.cq_n >= CMP_CACHE) ok = w_spans_equal<REV>(in, ca, cb, cl);      // (more long comparisons in one step than a lane can hold: by itself)
                else if (in.rq == 0u) { in.rq = 2u; in.rq_a = ca; in.rq_b = cb; in.rq_l = cl; }
            }
        }
    }
    return ok;
}

// the requests of the wave's lanes, answered by the whole wave (main loop, between two executions of a step)
template <bool REV>
WALK_DEV void answer_requests(const Store& st, WIn& in, bool active) {
    const uint32_t lane = wv_lane();
    {
        const bool nr = active && in.rq == 1u;
        if (__any(nr)) {
            const RunEnd re = run_end_for<REV>(in.bytes, in.total16, in.regions, in.gate, in.rtc, in.base, in.len, in.sid, in.rt_cnt, nr, in.rq_a, in.rq_b);
            if (nr) {
                in.run_lo = in.rq_a; in.run_hi = re.end; in.run_ch = in.rq_b;
                // a measured run is a periodic region too (unless one that reaches at least as far is known: a probe may rely on it)
                if (re.scanned && !(in.per_q != 0u && in.per_lo <= in.rq_a && in.rq_a < in.per_hi && in.per_hi >= re.end)) { in.per_lo = in.rq_a; in.per_hi = re.end; in.per_q = 1u; }
            }
        }
    }
    const bool cm = active && in.rq == 2u;
    for (unsigned long long sb = __ballot(cm); sb; sb &= sb - 1ull) {
        const int L = __builtin_ctzll(sb);
        const uint64_t pa = in.base + (REV ? (uint64_t)(in.len - in.rq_a - in.rq_l) : (uint64_t)in.rq_a), pb = in.base + (REV ? (uint64_t)(in.len - in.rq_b - in.rq_l) : (uint64_t)in.rq_b);
        const bool r = coop_mem_equal_x(in.bytes, wv_shfl64(pa, L), wv_shfl64(pb, L), __shfl(in.rq_l, L), lane);
        if (lane == (uint32_t)L) {
            uint32_t* q = st.gq + (size_t)(in.cq_n * 4u) * WALK_WV + lane;
            q[0] = in.rq_a; q[WALK_WV] = in.rq_b; q[2 * WALK_WV] = in.rq_l; q[3 * WALK_WV] = r ? 1u : 0u;
            in.cq_n++;
        }
    }
    in.rq = 0u;
}

// ---- the step --------------------------------------------------------------------------------------------------------------------------
#ifndef WALK_KEYS
#define WALK_KEYS 4
#endif
constexpr uint32_t KEYS = WALK_KEYS;      // entries of the list being built whose keys are also kept in registers (plain steps; eight of them cost
                                  // four more registers at the kernel's peak -- one region wave fewer beside two walk waves --: automata with long lists get a kernel
                                  // of their own, which finds entries through a node map instead: WALK_NODE_MAP, walk.hip)
template <class U> struct KeyCache;
template <> struct KeyCache<uint32_t> {
    static constexpr uint32_t N = KEYS;
    uint32_t id[KEYS], P[KEYS];      // node | tie << 16 and P of the first KEYS entries: a candidate for one of them is decided without reading the list
    unsigned long long seen;         // bit (node & 63): some entry of the list has such a node -- a clear bit spares the search behind the keys
};
template <> struct KeyCache<Dual> { static constexpr uint32_t N = 0; uint32_t id[1], P[1]; unsigned long long seen; };      // dual steps are rare: they search the list
template <class U, int K>
struct StepCtx {
    const Store& st;
    uint32_t nxt;            // the list being built
    uint32_t n_next;         // per lane: entries in it
    bool dual_lane;          // per lane: directions are meaningful
    bool fits;               // per lane: every direction stored so far fits 16 bits
    tb_t TB;
    KeyCache<U> keys;
};

// candidate (P, tie, cells of t) for the node of `vid` on the lanes in `pred`
template <class U, int K>
WALK_DEV void insert(StepCtx<U, K>& cx, bool pred, uint32_t vid, uint32_t vbits, U P, uint32_t tie, Ent<U, K>& t) {
    const uint32_t node = vid >> vbits;
    if (pred) WALK_EV(2);
    uint32_t at = ~0u, old_id = 0u;
    U old = konst<U>(0u);
    constexpr uint32_t NK = WALK_NODE_MAP ? 0u : KeyCache<U>::N;
#if WALK_NODE_MAP
    // the node's entry in the list being built, if it has one: one byte of the lane's map instead of a search through keys and list (lists of
    // 10-18 entries on the 77-node automata: eight keys in registers, a compare-and-select chain per insertion, a loop behind them)
    uint32_t old_tie = 0u;
    if (pred) {
        const uint32_t m = cx.st.nm[nm_at(node)];
        if (m != 0xffu && (m >> 7) == cx.nxt) {
            at = m & 0x7fu;
            old_tie = rd_v<K>(cx.st, cx.nxt, at, 1) >> 16;
            setv(old, rd_v<K>(cx.st, cx.nxt, at, 0));
        }
    }
#else
    // (slots the list has not reached hold node 0xffff, which no automaton has; lanes outside `pred` find whatever they find: every use of
    // `at` below asks for pred)
#pragma unroll
    for (uint32_t k = 0; k < (NK < 4u ? NK : 4u); k++)
        if ((cx.keys.id[k] & 0xffffu) == node) { at = k; old_id = cx.keys.id[k]; setv(old, cx.keys.P[k]); }
    if (NK > 4u && __any(pred && at == ~0u && cx.n_next > 4u)) {      // (lists of the README automata never get here)
#pragma unroll
        for (uint32_t k = 4; k < NK; k++)
            if ((cx.keys.id[k] & 0xffffu) == node) { at = k; old_id = cx.keys.id[k]; setv(old, cx.keys.P[k]); }
    }
    uint32_t old_tie = old_id >> 16;
    if (__any(pred && at == ~0u && cx.n_next > NK)) {                   // longer lists: look through the rest (the filter only knows the entries behind the keys)
        const bool maybe = pred && at == ~0u && (NK == 0u || ((cx.keys.seen >> (node & 63u)) & 1ull) != 0ull);
        for (uint32_t j = NK; __any(maybe && at == ~0u && j < cx.n_next); j++) {
            WALK_EV(3);
            const bool look = maybe && at == ~0u && j < cx.n_next;
            const uint32_t x = look ? rd_v<K>(cx.st, cx.nxt, j, 1) : 0u;
            if (look && ((x & 0xffffu) >> vbits) == node) { at = j; old_tie = x >> 16; setv(old, rd_v<K>(cx.st, cx.nxt, j, 0)); }
        }
    }
#endif
    bool win = pred;
    if (pred && at != ~0u) {
        load_pdir<K>(cx.st, cx.nxt, at, cx.dual_lane, old);
        win = lt(P, old, cx.TB) || (eq(P, old, cx.TB) && tie < old_tie);
    }
    if (NK != 0u && __any(win && at == ~0u && cx.n_next >= NK)) {       // an entry behind the keys: the filter learns its node
        if (win && at == ~0u && cx.n_next >= NK) cx.keys.seen |= 1ull << (node & 63u);
    }
    if (win) {
        if (at == ~0u) {
            at = cx.n_next++;
#if WALK_NODE_MAP
            cx.st.nm[nm_at(node)] = (uint8_t)((cx.nxt << 7) | at);
#endif
        }
        t.P = P; t.vid = vid;
        store_entry<U, K>(cx.st, cx.nxt, at, t, tie, cx.fits);
#pragma unroll
        for (uint32_t k = 0; k < NK; k++)
            if (at == k) { cx.keys.id[k] = node | (tie << 16); cx.keys.P[k] = val(P); }
    }
}

#if WALK_NODE_MAP
// A step that is executed again (the wave has answered what its first execution asked for) builds its list anew: the places the abandoned
// list holds in the lanes' maps are given back first.
template <int K> WALK_DEV void map_forget(const Store& st, uint32_t vbits, uint32_t l, bool pred, uint32_t n) {
    for (uint32_t j = 0; __any(pred && j < n); j++)
        if (pred && j < n) {
            WALK_LDS uint8_t* const mp = st.nm + nm_at((rd_v<K>(st, l, j, 1) & 0xffffu) >> vbits);
            if (*mp == (uint8_t)((l << 7) | j)) *mp = 0xffu;
        }
}
#endif

template <int K, class TP>
WALK_DEV void ee_decode(TP T, uint32_t at, bool p, uint32_t& e0, uint32_t& actions, uint32_t& cm, uint32_t& com, uint32_t& rdm) {
    e0 = p ? T[at] : 0u;
    const uint32_t e1 = p ? T[at + 1] : 0u;
    if (Lay<K>::EEW == 2) {
        actions = e1 & 0xfffu; cm = (e1 >> 12) & 0x3fu; com = (e1 >> 18) & 0x3fu; rdm = (e1 >> 24) & 0x3fu;
    } else {
        const uint32_t e2 = p ? T[at + 2] : 0u;
        actions = e1 & 0x3ffffu; cm = (e1 >> 18) & 0x1ffu; com = e2 & 0x1ffu; rdm = (e2 >> 9) & 0x1ffu;
    }
}

// the state an effective edge starts from: the entry's cells, the cells created on the way (empty, at the entry's pos:
// mfa.cpp:151-158) and the is_read marks earlier read attempts of this evaluation have left
template <class U, int K>
WALK_DEV void frame_state(const Ent<U, K>& E, U pos, uint32_t cm, uint32_t com, uint32_t rdm, Ent<U, K>& s) {
#pragma unroll
    for (int c = 0; c < K; c++) {
        const bool created = (cm >> c) & 1u;
        s.S[c] = sel(created, pos, E.S[c]);
        s.L[c] = sel(created, konst<U>(0u), E.L[c]);
        s.F[c] = created ? (F_PRESENT | F_UNI | (((com >> c) & 1u) ? F_OPEN : 0u)) : E.F[c];
        if ((rdm >> c) & 1u) s.F[c] |= F_READ;
    }
}

// One MFA::evaluateStates call (mfa.cpp:203-213) for every active lane: reads list `cur` (n_cur entries), builds the other one.
#ifndef WALK_STEP_ATTR
#define WALK_STEP_ATTR WALK_DEV
#endif
template <class U, int K, bool REV, class TP>
WALK_STEP_ATTR void walk_step(const Store& st, TP T, const Aut& au, WIn& in, uint32_t cur, uint32_t n_cur, uint32_t& n_next,
                        const U i, const U len, const uint32_t ch, const bool final_pass, const bool active, const bool dual_lane,
                        bool& accept, bool& fits, tb_t& TB, uint32_t& shape) {
    StepCtx<U, K> cx{st, cur ^ 1u, 0u, dual_lane, fits, TB, {}};
    cx.keys.seen = 0ull;
    uint32_t sh = 0u;                                                // the vnodes of the list read, in list order, folded into one word (ShapeHist)
#pragma unroll
    for (uint32_t k = 0; k < KeyCache<U>::N; k++) cx.keys.id[k] = 0xffffffffu;
    const uint32_t cls = (active && !final_pass) ? (T[au.cmap() + ((ch & 0xffu) >> 2)] >> (8u * (ch & 3u))) & 0xffu : 0u;
    for (uint32_t e = 0; __any(active && e < n_cur); e++) {
        const bool have = active && e < n_cur;
        if (have) WALK_EV(0);
        Ent<U, K> E;
        load_entry<U, K>(st, cur, have ? e : 0u, dual_lane && have, E);
        const U pos = have ? posof(E.P, cx.TB) : konst<U>(0u);
        bool live = have && ge(pos, i, cx.TB);                       // a state the scan has passed does nothing (but it held its node)
        if (REV) {                                                   // mfa.cpp:116-133
            if (live) {
                U need = konst<U>(0u);
#pragma unroll
                for (int c = 0; c < K; c++)
                    if ((E.F[c] & F_PRESENT) && ((E.F[c] & F_OPEN) || !(E.F[c] & F_READ))) need = add(need, E.L[c]);
                live = le(need, sub(len, i), cx.TB);
            }
        }
        const bool here = live && !final_pass && eq(pos, i, cx.TB);
        const bool wait = live && !final_pass && !here;
        const uint32_t vi = have ? T[au.vinfo() + E.vid] : 0u;
        const uint32_t node = E.vid >> au.vbits();
#if WALK_NODE_MAP
        if (have) {                                                  // this entry has been read: its place in the map is free (unless the list being built has taken it)
            WALK_LDS uint8_t* const mp = st.nm + nm_at(node);
            if (*mp == (uint8_t)((cur << 7) | e)) *mp = 0xffu;
        }
#endif
        if (have) {                                                  // (a state the scan has passed counts as far ahead: it only holds its node)
            const uint32_t ahead = val(pos) - val(i);
            sh = ((sh << 7) | (sh >> 25)) ^ (E.vid * 4u + (ahead < 3u ? ahead : 3u) + 0x9e37u);
        }
        // an epsilon edge: the state reaches `finish`, which keeps it iff pos == len (mfa.cpp:138-147); accepting is sticky
        if ((vi & VI_EPS) && live && !accept && eq(pos, len, cx.TB)) accept = true;
        {   // a waiting state goes back into the set as it is (mfa.cpp:195-197): older than everything created in this step
            const bool carry = wait && (vi & VI_QUAL) != 0u;
            if (__any(carry)) { Ent<U, K> t = E; insert<U, K>(cx, carry, E.vid, au.vbits(), E.P, TIE_CARRY, t); }
        }
        // ---- the state at pos == i consumes (mfa.cpp:161-194)
        if (__any(here)) {
            const uint32_t bl = here ? T[au.vb() + E.vid * au.nc() + cls] : 0u;
            const uint32_t bbeg = bl >> 12, bcnt = bl & 0xfffu;
            for (uint32_t j = 0; __any(j < bcnt); j++) {
                const bool p = here && j < bcnt;
                if (p) WALK_EV(1);
                uint32_t e0, actions, cm, com, rdm;
                ee_decode<K, TP>(T, au.ee() + (bbeg + j) * Lay<K>::EEW, p, e0, actions, cm, com, rdm);
                const uint32_t tvid = e0 >> 9, tfn = (e0 >> 5) & 15u, cell = (e0 >> 1) & 15u;
                // the state the edge starts from = the candidate before the edge's actions: the entry's cells (most edges), or those
                // plus the cells created on the way and the marks of earlier reads.  The copy is taken before read() marks the source
                // (mfa.cpp:167 / 177): the edge's own mark is not in rdm.
                Ent<U, K> t = E;
                if (__any(p && (cm | rdm) != 0u)) frame_state<U, K>(E, pos, cm, com, rdm, t);
                const bool lit = p && (e0 & 1u) == 0u, rd = p && (e0 & 1u) != 0u;
                if (__any(lit)) {                                    // a letter (or dot) edge takes the byte (mfa.cpp:171-175)
                    apply_actions<U, K>(t, actions, lit, i, konst<U>(1u), true, ch, cx.TB);
                    insert<U, K>(cx, lit, tvid, au.vbits(), mkp(add(i, konst<U>(1u)), tfn), TIE_HERE + node, t);
                }
                if (__any(rd)) {                                     // a cell edge reads the cell's value (mfa.cpp:176-191)
                    U vs = konst<U>(0u), vl = konst<U>(0u);
                    uint32_t vf = 0u;
#pragma unroll
                    for (int c = 0; c < K; c++)
                        if ((uint32_t)c == cell) { vs = t.S[c]; vl = t.L[c]; vf = t.F[c]; }
                    const bool ok = cell_read<REV, U>(st, in, rd, i, ch, vs, vl, vf, cx.TB);
                    if (__any(ok)) {
                        apply_actions<U, K>(t, actions, ok, i, vl, (vf & F_UNI) != 0u, (vf >> 8) & 0xffu, cx.TB);
                        insert<U, K>(cx, ok, tvid, au.vbits(), mkp(add(i, vl), tfn), TIE_HERE + node, t);
                    }
                }
            }
        }
        // ---- a waiting state (and, in the final pass, a state at pos == len) only follows edges of absent cells (mfa.cpp:148-160)
        const bool late = live && !here;
        if ((vi & VI_CACC) && late && !accept && eq(pos, len, cx.TB)) accept = true;
        if (__any(wait && (vi & VI_HASC) != 0u)) {
            const uint32_t cl = (wait && (vi & VI_HASC) != 0u) ? T[au.vc() + E.vid] : 0u;
            const uint32_t cbeg = cl >> 12, ccnt = cl & 0xfffu;
            for (uint32_t j = 0; __any(j < ccnt); j++) {
                const bool p = wait && j < ccnt;
                if (p) WALK_EV(6);
                uint32_t e0, actions, cm, com, rdm;
                ee_decode<K, TP>(T, au.ee() + (cbeg + j) * Lay<K>::EEW, p, e0, actions, cm, com, rdm);
                Ent<U, K> t;
                frame_state<U, K>(E, pos, cm, com, 0u, t);
                insert<U, K>(cx, p, e0 >> 9, au.vbits(), mkp(pos, (e0 >> 5) & 15u), TIE_WAIT + node, t);
            }
        }
    }
    n_next = cx.n_next; fits = cx.fits; TB = cx.TB; shape = sh;
}

// ---- the list one period ago (SB) and its movement (SA) ----------------------------------------------------------------------------------
// Whole entries at a time: ONE branch on where the entry's image lives (LDS below CI, the wave's area in global memory above), then W (or
// DW) accesses with constant offsets from one address.  (Word by word -- a branch and an address per word -- the five functions below were
// 18 % of a wave-iteration on the 77-node automata, whose lists are mostly in global memory.)
template <int K> WALK_DEV void sb_rd_words(const Store& st, uint32_t e, uint32_t (&w)[Lay<K>::W]) {
    if (e < st.CI) {
        const WALK_LDS uint32_t* p = st.sb + (e * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) w[k] = p[k * WALK_WV];
    } else {
        const uint32_t* p = st.gsb + (size_t)((e - st.CI) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) w[k] = p[k * WALK_WV];
    }
}
template <int K> WALK_DEV void sb_wr_words(const Store& st, uint32_t e, const uint32_t (&w)[Lay<K>::W]) {
    if (e < st.CI) {
        WALK_LDS uint32_t* p = st.sb + (e * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) p[k * WALK_WV] = w[k];
    } else {
        uint32_t* p = st.gsb + (size_t)((e - st.CI) * Lay<K>::W) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::W; k++) p[k * WALK_WV] = w[k];
    }
}
template <int K> WALK_DEV void sa_rd_dwords(const Store& st, uint32_t e, uint32_t (&w)[Lay<K>::DW]) {
    if (e < st.CI) {
        const WALK_LDS uint32_t* p = st.sa + (e * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = p[k * WALK_WV];
    } else {
        const uint32_t* p = st.gsa + (size_t)((e - st.CI) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) w[k] = p[k * WALK_WV];
    }
}
template <int K> WALK_DEV void sa_wr_dwords(const Store& st, uint32_t e, const uint32_t (&w)[Lay<K>::DW]) {
    if (e < st.CI) {
        WALK_LDS uint32_t* p = st.sa + (e * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) p[k * WALK_WV] = w[k];
    } else {
        uint32_t* p = st.gsa + (size_t)((e - st.CI) * Lay<K>::DW) * WALK_WV + wv_lane();
#pragma unroll
        for (uint32_t k = 0; k < Lay<K>::DW; k++) p[k * WALK_WV] = w[k];
    }
}

// index of a value word's direction: 0 = P (in units of 16), 1 + 2c = S of cell c, 2 + 2c = L of cell c; word 1 has none
WALK_DEV int dir_index(uint32_t w) { return w == 0u ? 0 : (int)w - 1; }

// the movement of an entry since its image `b` was taken, as packed directions; wide: one does not fit 16 bits (or the first-cell name changed)
template <int K> WALK_DEV void entry_movement(const uint32_t (&v)[Lay<K>::W], const uint32_t (&b)[Lay<K>::W], uint32_t (&nw)[Lay<K>::DW], bool& wide) {
    int32_t d[2 * Lay<K>::DW];
#pragma unroll
    for (uint32_t k = 0; k < 2 * Lay<K>::DW; k++) d[k] = 0;
#pragma unroll
    for (uint32_t w = 0; w < Lay<K>::W; w++) {
        if (w == 1u) continue;
        int32_t dv = (int32_t)(v[w] - b[w]);
        if (w == 0u) { if (dv & 15) wide = true; dv >>= 4; }
        d[dir_index(w)] = dv;
        if (dv != (int32_t)(int16_t)dv) wide = true;
    }
#pragma unroll
    for (uint32_t k = 0; k < Lay<K>::DW; k++) nw[k] = ((uint32_t)d[2 * k] & 0xffffu) | ((uint32_t)d[2 * k + 1] << 16);
}

// save list `cur` of the lanes in `pred`
template <int K> WALK_DEV void image_save(const Store& st, uint32_t cur, bool pred, uint32_t n) {
    for (uint32_t e = 0; __any(pred && e < n); e++)
        if (pred && e < n) {
            uint32_t v[Lay<K>::W];
            rd_words<K>(st, cur, e, v);
            v[1] &= 0xffffu;
            sb_wr_words<K>(st, e, v);
        }
}

// a period boundary on the lanes in `pred`: movement of the list since the image was taken -> SA, list -> SB.
// moved: the movement differs from the one before; wide: a movement does not fit 16 bits; vac: other nodes than a period ago
template <int K> WALK_DEV void image_measure(const Store& st, uint32_t cur, bool pred, uint32_t n, uint32_t sb_n, bool& moved, bool& wide, bool& vac) {
    moved = wide = vac = false;
    if (pred && n != sb_n) { moved = true; vac = true; }
    for (uint32_t e = 0; __any(pred && e < n); e++)
        if (pred && e < n) {
            uint32_t v[Lay<K>::W], b[Lay<K>::W], nw[Lay<K>::DW], sa[Lay<K>::DW];
            rd_words<K>(st, cur, e, v);
            v[1] &= 0xffffu;
            sb_rd_words<K>(st, e, b);
            sa_rd_dwords<K>(st, e, sa);
            sb_wr_words<K>(st, e, v);
            if (v[1] != b[1]) { vac = true; moved = true; }
            entry_movement<K>(v, b, nw, wide);
#pragma unroll
            for (uint32_t k = 0; k < Lay<K>::DW; k++)
                if (e >= sb_n || nw[k] != sa[k]) moved = true;
            sa_wr_dwords<K>(st, e, nw);
        }
}

// the lanes in `pred` start their dual period: the list's directions are the movement just measured
template <int K> WALK_DEV void image_dirs(const Store& st, uint32_t cur, bool pred, uint32_t n) {
    for (uint32_t e = 0; __any(pred && e < n); e++)
        if (pred && e < n) {
            uint32_t sa[Lay<K>::DW];
            sa_rd_dwords<K>(st, e, sa);
            wr_dwords<K>(st, cur, e, sa);
        }
}

// after the dual period: did the list move by exactly SA again, and was SA mapped to itself?
template <int K> WALK_DEV bool image_same(const Store& st, uint32_t cur, bool pred, uint32_t n, uint32_t sb_n) {
    bool same = pred && n == sb_n;
    for (uint32_t e = 0; __any(same && e < n); e++)
        if (same && e < n) {
            uint32_t v[Lay<K>::W], b[Lay<K>::W], nw[Lay<K>::DW], sa[Lay<K>::DW], dd[Lay<K>::DW];
            rd_words<K>(st, cur, e, v);
            sb_rd_words<K>(st, e, b);
            sa_rd_dwords<K>(st, e, sa);
            rd_dwords<K>(st, cur, e, dd);
            bool wide = false;
            entry_movement<K>(v, b, nw, wide);
            uint32_t differs = ((v[1] & 0xffffu) ^ b[1]) | (wide ? 1u : 0u);
#pragma unroll
            for (uint32_t k = 0; k < Lay<K>::DW; k++) {
                differs |= sa[k] ^ dd[k];                                                              // the direction was mapped to itself
                differs |= sa[k] ^ nw[k];                                                              // the list moved by it again
            }
            if (differs) same = false;
        }
    return same;
}

// list += skip * direction
template <int K> WALK_DEV void image_advance(const Store& st, uint32_t cur, bool pred, uint32_t n, uint32_t skip) {
    for (uint32_t e = 0; __any(pred && e < n); e++)
        if (pred && e < n) {
            uint32_t v[Lay<K>::W], dw[Lay<K>::DW];
            rd_words<K>(st, cur, e, v);
            rd_dwords<K>(st, cur, e, dw);
#pragma unroll
            for (uint32_t w = 0; w < Lay<K>::W; w++) {
                if (w == 1u) continue;
                const int k = dir_index(w);
                const int32_t d = d16(dw[k / 2], (uint32_t)k) * (w == 0u ? 16 : 1);
                v[w] += skip * (uint32_t)d;
            }
            wr_words<K>(st, cur, e, v);
        }
}

// ---- one wave ------------------------------------------------------------------------------------------------------------------------------
struct Batch {
    const uint8_t* bytes;
    const uint64_t* offsets;
    uint64_t n;
    uint8_t* results;
    const uint64_t* regions;     // region table (regions.hip) or nullptr
    uint32_t accel;
    uint32_t refill;             // idle lanes of a wave that make it fetch new strings (1: every lane at once when it runs out)
    uint32_t n_seg;              // segments of the batch: strings seg_first[s] .. seg_first[s+1]-1 belong to the automaton whose tables
    const uint32_t* seg_first;   //   start at word seg_table[s] of the table block (n_seg + 1 / n_seg entries, 32-bit string indices)
    const uint32_t* seg_table;
    uint32_t gate;               // WIn::gate
    uint32_t* lean_queue;        // strings WITHOUT a periodic stretch (an empty table row) of at least WALK_LEAN_MIN_LEN bytes are not walked by this kernel:
    uint32_t* lean_count;        //   their numbers go to this queue (nullptr: none), which walk_wave_lean works off afterwards
};

struct WaveStats {
    unsigned long long iters = 0, dual = 0, skipped = 0, probes = 0, hits = 0, steps = 0, spills = 0;
    unsigned long long t_start = 0, t_byte = 0, t_look = 0, t_plain = 0, t_dual = 0, t_post = 0, t_total = 0;      // wave cycles by section (stats builds)
    unsigned long long strings = 0;
#ifdef MFA_HOST_EMUL
    unsigned long long hist[80] = {0};
#endif
};
#ifdef MFA_HOST_EMUL
WALK_DEV unsigned long long wv_clock() { return 0ull; }
#else
WALK_DEV unsigned long long wv_clock() { return __builtin_readcyclecounter(); }
#endif

// Feeder::take(want, sid): hands the next string index to every lane that wants one; returns false when the batch is exhausted
// The probe's small counters in ONE register (they are all alive across the step, where the kernel's register peak is: seven registers
// less there): phase 0..3 | pk, pp <= 16 | pmin (periods below it have failed in this region) | fails < 8 | log2 of the back-off (8 .. 4096)
// | plain periods of the probe (saturating) | chain (1: the lane has just jumped and tries the same movement again after one period)
struct ProbeCtl {
    uint32_t w;
    WALK_DEV uint32_t get(uint32_t sh, uint32_t bits) const { return (w >> sh) & ((1u << bits) - 1u); }
    WALK_DEV void put(uint32_t sh, uint32_t bits, uint32_t v) { w = (w & ~(((1u << bits) - 1u) << sh)) | ((v & ((1u << bits) - 1u)) << sh); }
    WALK_DEV void reset() { w = 0u; set_pp(1u); backoff_reset(); }
    WALK_DEV uint32_t phase() const { return get(0, 2); }   WALK_DEV void set_phase(uint32_t v) { put(0, 2, v); }
    WALK_DEV uint32_t pk() const { return get(2, 5); }      WALK_DEV void set_pk(uint32_t v) { put(2, 5, v); }      WALK_DEV void inc_pk() { set_pk(pk() + 1u); }
    WALK_DEV uint32_t pp() const { return get(7, 5); }      WALK_DEV void set_pp(uint32_t v) { put(7, 5, v); }
    WALK_DEV uint32_t pmin() const { return get(12, 5); }   WALK_DEV void set_pmin(uint32_t v) { put(12, 5, v); }
    WALK_DEV uint32_t fails() const { return get(17, 3); }  WALK_DEV void set_fails(uint32_t v) { put(17, 3, v); }  WALK_DEV void inc_fails() { set_fails(fails() + 1u); }
    WALK_DEV uint32_t backoff() const { return 1u << get(20, 4); }
    WALK_DEV void backoff_reset() { put(20, 4, 3u); }
    WALK_DEV void backoff_double() { const uint32_t l = get(20, 4); put(20, 4, l < 12u ? l + 1u : l); }
    WALK_DEV uint32_t nper() const { return get(24, 4); }   WALK_DEV void set_nper(uint32_t v) { put(24, 4, v); }   WALK_DEV void inc_nper() { const uint32_t n = nper(); set_nper(n < 15u ? n + 1u : n); }
    WALK_DEV uint32_t chain() const { return get(28, 1); }  WALK_DEV void set_chain(uint32_t v) { put(28, 1, v); }
    WALK_DEV uint32_t nochain() const { return get(29, 1); } WALK_DEV void set_nochain(uint32_t v) { put(29, 1, v); }      // the last attempt of that kind failed: not in this region again
};

// The shapes of the lists of a lane's last 12 steps (a list's vnodes and how far ahead of the scan each entry is -- 0, 1, 2, more --, in list
// order, folded to 5 bits by the step): field k = the list k steps before the one the next step will read; bits 60..63 = how many fields are
// valid.  A probe with period p starts only when the shape repeated with lag p: no probe in the middle of a transient, and the period of the
// LIST (a multiple of the input's) is read off instead of being guessed.  A jump skips whole periods of a list whose shape has that period:
// the history stays valid across it.
constexpr uint32_t HIST_MAX_P = 10u, HIST_N = 12u;
// does q (1..8) divide p (0..15)?  A bit of two constants (an integer remainder is ~40 instructions, and this sits in every iteration of a lane that is
// waiting to probe)
WALK_DEV bool divides_small(uint32_t q, uint32_t p) {
    const uint64_t lo = 0xffffull | (0x5555ull << 16) | (0x9249ull << 32) | (0x1111ull << 48);      // q = 1, 2, 3, 4: bit p of a 16-bit field
    const uint64_t hi = 0x8421ull | (0x1041ull << 16) | (0x4081ull << 32) | (0x0101ull << 48);      // q = 5, 6, 7, 8
    return (((q <= 4u ? lo : hi) >> ((((q - 1u) & 3u) << 4) + (p & 15u))) & 1ull) != 0ull;
}
struct ShapeHist {
    uint64_t h;
    WALK_DEV void reset() { h = 0ull; }
    WALK_DEV void push(uint32_t shape) {
        uint32_t c = (uint32_t)(h >> 60);
        c = c < HIST_N ? c + 1u : HIST_N;
        const uint32_t x = (shape * 0x9e3779b1u) >> 27;
        h = ((uint64_t)c << 60) | ((h << 5) & 0x0fffffffffffffe0ull) | x;
    }
    WALK_DEV bool lag(uint32_t p) const {                    // did the shapes of the last two steps occur p steps earlier?
        const uint32_t c = (uint32_t)(h >> 60);
        if (p + 2u > c) return false;
        return ((h ^ (h >> (5u * p))) & 1023ull) == 0ull;
    }
    // the smallest multiple p of q with pmin <= p <= HIST_MAX_P that the history supports; 0: none (yet).  Straight-line: the lags 1..10 that
    // repeat as a bit mask, the multiples of q, the lags the history is long enough for, those not below pmin -- and the lowest bit left.
    WALK_DEV uint32_t period(uint32_t q, uint32_t pmin) const {
        const uint32_t c = (uint32_t)(h >> 60);
        uint32_t m = 0u;
#pragma unroll
        for (uint32_t p = 1; p <= HIST_MAX_P; p++)
            if (((h ^ (h >> (5u * p))) & 1023ull) == 0ull) m |= 1u << p;
        const uint64_t lo = 0xffffull | (0x5555ull << 16) | (0x9249ull << 32) | (0x1111ull << 48), hi = 0x8421ull | (0x1041ull << 16) | (0x4081ull << 32) | (0x0101ull << 48);
        m &= (uint32_t)((q <= 4u ? lo : hi) >> (((q - 1u) & 3u) << 4)) & 0xffffu;       // multiples of q
        m &= c >= 3u ? (1u << (c - 1u)) - 2u : 0u;                                        // p + 2 <= c, p >= 1
        m &= ~((1u << (pmin < 16u ? pmin : 16u)) - 1u);                                   // p >= pmin
        return m ? (uint32_t)__builtin_ctz(m) : 0u;
    }
};

template <int K, bool REV, class Feeder, class TP>
WALK_DEV void walk_wave(const Batch& b, TP T, const Store& st, WALK_LDS uint64_t* rt_cache, Feeder& feed, WaveStats* stats) {
    WIn in;
    in.bytes = b.bytes; in.total16 = (b.offsets[b.n] + 15u) & ~(uint64_t)15; in.regions = b.regions; in.rtc = rt_cache; in.gate = b.gate;
    w_reset(in, 0, 0, 0);
    in.w0 = in.w1 = in.w2 = in.w3 = 0;
    bool active = false, exhausted = false, accept = false;
    uint32_t i = 0, len = 0;      // (the string's number is in.sid: a launch holds fewer than 2^32 strings)
    // probes: P.phase() 0 idle, 1 = plain periods after saving the list, 2 = the dual period, 3 = the dual period is over and looked sound
    uint32_t probe_at = 0;
    ProbeCtl P;
    P.reset();
    ShapeHist hist;
    hist.reset();
    tb_t TBacc = tb_init();
    bool fits = true, stable = false, patient = false;
    uint32_t cur = 0, n_cur = 0, sb_n = 0, warm = 0;
    Aut au;
    aut_load(au, T, 0u);
    const unsigned long long tm_begin = stats ? wv_clock() : 0ull;
    for (;;) {
        unsigned long long tm = stats ? wv_clock() : 0ull;
#define WALK_LAP(field) do { if (stats) { const unsigned long long now_ = wv_clock(); stats->field += now_ - tm; tm = now_; } } while (0)
        for (;;) {   // hand strings to idle lanes (again, if every string this trip handed out went to the lean queue)
            // Idle lanes take new strings together: a string's start is three dependent trips to memory (ticket, offsets, first
            // bytes) that the whole wave waits for, so it is paid once per `refill` lanes, not once per lane that runs out
            const bool want = !active && !exhausted;
            const unsigned long long wantb = __ballot(want);
            if (wantb && ((uint32_t)__builtin_popcountll(wantb) >= b.refill || !__any(active))) {
                uint64_t s = 0;
                const bool got = feed.take(want, s);
                uint4 rta = make_uint4(0, 0, 0, 0), rtb = rta;
                if (want && got) rt_fetch(b.regions, s, rta, rtb);   // the table row and the offsets travel together
                if (b.gate != 0u) w_rt_await(in, want && got, s, rta, rtb);      // (the row is being written while this kernel runs: not without the call's stamp)
                if (want) {
                    if (!got) exhausted = true;
                    else {
                        const uint64_t sid = s;
                        const uint64_t o0 = b.offsets[sid], o1 = b.offsets[sid + 1];
                        if (o1 - o0 > MFA_DEV_MAX_LEN) b.results[sid] = 2;
                        else if (b.lean_queue != nullptr && b.regions != nullptr && (rta.x & 0x1ffu) == 0u && o1 - o0 >= WALK_LEAN_MIN_LEN) {
                            // A string whose row is empty has no periodic stretch: every one of its steps will be executed, nothing of the probe
                            // machinery is of use to it.  Such strings go to a queue and are walked afterwards by a kernel that has the plain step
                            // only -- fewer registers, less LDS, twice the waves per SIMD (walk_wave_lean).  (The lanes in this branch.)
                            const unsigned long long lb = __ballot(true);
                            const int leader = __builtin_ctzll(lb);
                            uint32_t at = 0;
                            if (wv_lane() == (uint32_t)leader) at = wv_atomic_add(b.lean_count, (uint32_t)__builtin_popcountll(lb));
                            at = wv_bcast32(at, leader);
                            b.lean_queue[at + (uint32_t)__builtin_popcountll(lb & ((1ull << wv_lane()) - 1ull))] = (uint32_t)sid;
                        } else {
                            len = (uint32_t)(o1 - o0);
                            w_reset(in, o0, len, (uint32_t)sid);
                            w_rt_attach(in, warm, rta, rtb);
                            uint32_t seg = 0;
                            for (uint32_t k = 1; k < b.n_seg; k++)
                                if (sid >= b.seg_first[k]) seg = k;
                            aut_load(au, T, b.n_seg ? b.seg_table[seg] : 0u);
                            i = 0; accept = false; active = true; probe_at = 0;
                            P.reset(); hist.reset();
#if WALK_NODE_MAP
                            for (uint32_t k = 0; k < st.nm_words; k++) reinterpret_cast<WALK_LDS uint32_t*>(st.nm)[k * WALK_WV + wv_lane()] = 0xffffffffu;
#endif
                            stable = false; patient = false;
                            n_cur = 1;                                // the list: (pos 0, start, no cells)  mfa.cpp:217-219
                            Ent<uint32_t, K> e0;
                            e0.P = 0u; e0.vid = aut_start(T, au);
#pragma unroll
                            for (int c = 0; c < K; c++) { e0.S[c] = 0u; e0.L[c] = 0u; e0.F[c] = 0u; }
                            bool f2 = true;
                            store_entry<uint32_t, K>(st, cur, 0u, e0, 0u, f2);
                        }
                    }
                }
            }
            if (__any(active) || __all(exhausted)) break;
        }
        if (!__any(active)) break;
        if (stats) stats->iters++;
        WALK_LAP(t_start);
        const bool final_pass = (i == len);
        uint32_t ch = 0x100u;
        if (active && !final_pass) ch = w_stream_byte<REV>(in, i);
        WALK_LAP(t_byte);
        // ---- a lane that has just jumped tries the same movement again one period later: a jump ends where a comparison is about to change its
        // outcome, and what follows is often the same movement of another state (a cell opened later takes over): the list is saved, gets the
        // movement of the last jump as its directions and goes through a dual period at once -- which proves the movement or refutes it
        {
            const bool again = b.accel && active && !final_pass && P.phase() == 0u && P.chain() != 0u && i == probe_at;
            const bool room = again && in.per_q != 0u && in.per_lo <= i && in.per_hi >= i + 1u + 2u * P.pp() && n_cur == sb_n;
            if (again) P.set_chain(0u);
            if (__any(room)) {
                image_save<K>(st, cur, room, n_cur);
                image_dirs<K>(st, cur, room, n_cur);
            }
            if (room) { P.set_phase(2u); P.set_pk(0u); P.set_nper(15u); TBacc = tb_init(); fits = true; if (stats) stats->probes++; }
        }
        // ---- does this lane sit in a stretch that repeats, with a list whose shape has repeated?  (probes run in epochs: all lanes that measure do
        // it together, with the same period)
        uint32_t q = 0u;
        const bool ep_busy = __any(P.phase() == 1u);
        if (b.accel && active && !final_pass && P.phase() == 0u && i >= probe_at && !ep_busy) {
            if (in.per_q != 0u && in.per_lo <= i && i < in.per_hi) q = in.per_q;      // still inside the region found last
            else if (in.regions != nullptr) {
                uint32_t rl, rh, rq, rn;
                if (w_rt_find<REV>(in, i, rl, rh, rq, rn)) {
                    if (in.per_q != 0u) { in.prev_lo = in.per_lo; in.prev_hi = in.per_hi; in.prev_q = in.per_q; }
                    in.per_lo = rl; in.per_hi = rh; in.per_q = rq; q = rq;
                    P.set_pmin(0u); P.set_nochain(0u); patient = false;
                } else probe_at = rn;                                        // look again where the next region starts (never, if there is none)
            } else probe_at = ~0u;                                           // no table: every step is executed
        }
        if (q == 1u) { in.run_lo = i; in.run_hi = in.per_hi; in.run_ch = ch; }
        uint32_t want_p = 0u;
        if (q != 0u) {
            if (in.per_hi - i < 4u * q + WALK_TAIL) probe_at = in.per_hi > i + 1u ? in.per_hi : i + 1u;      // too short to be worth a probe
            else {
                want_p = hist.period(q, P.pmin());
                if (want_p != 0u && in.per_hi - i < 4u * want_p + WALK_TAIL) want_p = 0u;
            }
        }
        const unsigned long long cand = __ballot(want_p != 0u);
        const uint32_t ep_pp = cand ? __shfl(want_p, __builtin_ctzll(cand)) : 0u;      // the first candidate's period leads the epoch
        {
            bool begin = false;
            if (want_p != 0u || (q != 0u && ep_pp != 0u)) {
                // (a lane whose own choice differs joins if the epoch's period suits its list as well)
                if (divides_small(q, ep_pp) && ep_pp >= P.pmin() && (want_p == ep_pp || hist.lag(ep_pp)) && in.per_hi - i >= 4u * ep_pp + WALK_TAIL) {
                    begin = true; P.set_pp(ep_pp); P.set_phase(1u); P.set_pk(0u); P.set_nper(0u); stable = false; sb_n = n_cur;
                    if (stats) stats->probes++;
                }
            }
            if (__any(begin)) image_save<K>(st, cur, begin, n_cur);
        }
        WALK_LAP(t_look);
        uint32_t n_next = 0, shape = 0;
        tb_t TB = tb_init();
        const bool p2 = P.phase() == 2u;
        const bool accept_before = accept, fits_before = fits;
        in.cq_n = 0u;
        for (;;) {                                                   // (again after the wave has answered what the step asked for)
            n_next = 0; TB = tb_init(); accept = accept_before; fits = fits_before;
            if (WALK_WITH_DUAL && __any(p2)) {
                // dual step: lanes in their dual period carry the list's directions, the others direction 0 (their TB is ignored)
                if (stats) stats->dual++;
                const Dual di{i, (int32_t)P.pp()}, dlen{len, 0};
                in.dual_p = p2 ? P.pp() : 0u;
                (void)lt(di, Dual{p2 ? in.per_hi : i + 1u, 0}, TB);     // the byte at this step of the period repeats while i is inside the region
                (void)eq(di, dlen, TB);
                walk_step<Dual, K, REV, TP>(st, T, au, in, cur, n_cur, n_next, di, dlen, ch, final_pass, active, p2, accept, fits, TB, shape);
                in.dual_p = 0u;
                WALK_LAP(t_dual);
            } else {
                bool f2 = true;
                walk_step<uint32_t, K, REV, TP>(st, T, au, in, cur, n_cur, n_next, i, len, ch, final_pass, active, false, accept, f2, TB, shape);
                WALK_LAP(t_plain);
            }
            if (!__any(active && in.rq != 0u)) break;
            answer_requests<REV>(st, in, active);
#if WALK_NODE_MAP
            map_forget<K>(st, au.vbits(), cur ^ 1u, active, n_next);
#endif
        }
        cur ^= 1u;
        if (active) { n_cur = n_next; hist.push(shape); }
        const bool any_next = n_next != 0u;
        if (stats && active && n_cur > st.C) stats->spills++;
#ifdef MFA_HOST_EMUL
        if (stats && active) stats->hist[n_cur < 79u ? n_cur : 79u]++;
#endif
        uint32_t skip = 0;
        const bool chained = p2 && P.nper() == 15u;                  // this dual period repeats the last jump's movement (no plain periods before it)
        // a probe that did not lead to a jump: the first failure after an optimistic start (one plain period) makes the lane patient (two equal
        // movements before the next dual period); after that the period is given up for this region (the list may move with a multiple of it)
        auto probe_failed = [&]() {
            P.inc_fails();
            if (!patient) patient = true;
            else P.set_pmin(P.pp() + 1u);
            if (P.fails() >= 6u || P.pmin() > HIST_MAX_P) { P.set_fails(0u); P.set_pmin(0u); P.backoff_double(); }
        };
        if (p2) {
            tb_min(TBacc, TB.a, TB.b);
            P.inc_pk();
            if (P.pk() == P.pp()) {
                const int64_t periods = tb_steps(TBacc);
                const bool same = !accept && any_next && periods > 1 && fits;
                P.set_phase(same ? 3u : 0u);                              // 3: the comparison with the image decides (wave-level code: below)
            } else if (tb_is_one(TBacc) || accept || !any_next) {
                P.set_phase(0u);                                          // cannot succeed any more: stop the probe here
                if (!chained) probe_failed(); else P.set_nochain(1u);
            }
        }
        if (__any(P.phase() == 3u)) {
            const bool same = image_same<K>(st, cur, P.phase() == 3u, n_cur, sb_n);
            if (P.phase() == 3u) {
                if (same) {
                    const int64_t periods = tb_steps(TBacc);
                    skip = (uint32_t)(periods - 1 < (int64_t)0x00ffffff ? periods - 1 : (int64_t)0x00ffffff);
                }
                P.set_phase(0u);
            }
        }
        if (p2 && P.pk() == P.pp() && P.phase() == 0u) {
            if (skip) { P.backoff_reset(); P.set_fails(0u); patient = false; if (stats) { stats->hits++; stats->skipped += (unsigned long long)skip * P.pp(); } }
            else if (!chained) probe_failed();
            else P.set_nochain(1u);
        }
        if (__any(skip != 0u)) image_advance<K>(st, cur, skip != 0u, n_cur, skip);
        if (skip) { i += skip * P.pp(); w_drop_window(in); probe_at = i + 1u + P.pp(); P.set_chain(P.nochain() ^ 1u); sb_n = n_cur; }      // (i is incremented below: the next period, then the same movement again)
        else if (p2 && P.phase() == 0u) probe_at = i + ((chained || P.fails()) ? 1u : P.backoff());
        if (P.phase() == 1u) P.inc_pk();
        // plain periods of a probe: after each one the movement of the list over the period is compared with the previous period's; a
        // lane is ready for the dual period after the first period if the list kept its nodes (optimistic: the dual period itself proves or
        // refutes the movement), a patient lane once two consecutive movements agree.  All lanes of an epoch reach their period boundaries in
        // the same iteration and go on together.
        {
            const bool at_end = P.phase() == 1u && P.pk() == P.pp();
            if (__any(at_end)) {
                bool moved, wide, vac;
                image_measure<K>(st, cur, at_end, n_cur, sb_n, moved, wide, vac);
                if (at_end) {
                    const bool eqd = P.nper() != 0u && !moved;
                    const bool occ = P.nper() == 0u && !patient && !vac;      // first period: the same nodes before and after it
                    fits = !wide;
                    sb_n = n_cur;
                    P.inc_nper(); P.set_pk(0u); stable = (eqd || occ) && fits;
                }
            }
        }
        {
            const bool at_b = P.phase() == 1u && P.pk() == 0u && P.nper() != 0u;
            if (__any(at_b)) {
                const bool room = in.per_hi >= i + 2u + 2u * P.pp();      // the dual period and at least one more to skip
                if (!__any(at_b && !stable && room && P.nper() < (P.pp() > 2u ? WALK_PROBE_PERIODS : 3u))) {
                    const bool go = at_b && stable && room;
                    if (__any(go)) image_dirs<K>(st, cur, go, n_cur);
                    if (go) { P.set_phase(2u); TBacc = tb_init(); P.set_pk(0u); if (P.nper() == 15u) P.set_nper(14u); }
                    else if (at_b) {
                        P.set_phase(0u);
                        if (!room) probe_at = in.per_hi > i + 1u ? in.per_hi : i + 1u;
                        else {                                        // never settled
                            probe_failed();
                            probe_at = i + 1u + (P.fails() ? 0u : P.backoff());
                        }
                    }
                }
            }
        }
        if (active) {
            const bool done = accept || final_pass || !any_next;      // mfa.cpp:224-225, 227-235
            i++;
            if (stats) stats->steps++;
            if (done) {
                b.results[in.sid] = (warm == 0x9e3779b9u && len == 0xffffffffu) ? 3 : (accept ? 1 : 0);      // (warm keeps the touches alive)
                active = false; P.set_phase(0u); P.set_chain(0u); n_cur = 0u;
                if (stats) stats->strings++;
            }
        }
        WALK_LAP(t_post);
    }
#undef WALK_LAP
    if (stats) stats->t_total += wv_clock() - tm_begin;
}

// The walk of strings without periodic stretches (the queue the walk above fills): the plain step and nothing else -- no probes, no
// images, no directions, no shape history.  Same lists, same step function, same answers; about 60 registers fewer, a third of the LDS.
template <int K, bool REV, class Feeder, class TP>
WALK_DEV void walk_wave_lean(const Batch& b, TP T, const Store& st, Feeder& feed) {
    WIn in;
    in.bytes = b.bytes; in.total16 = (b.offsets[b.n] + 15u) & ~(uint64_t)15; in.regions = nullptr; in.rtc = nullptr; in.gate = 0u;
    w_reset(in, 0, 0, 0);
    in.w0 = in.w1 = in.w2 = in.w3 = 0;
    bool active = false, exhausted = false, accept = false;
    uint32_t i = 0, len = 0, cur = 0, n_cur = 0;
    Aut au;
    aut_load(au, T, 0u);
    for (;;) {
        {
            const bool want = !active && !exhausted;
            const unsigned long long wantb = __ballot(want);
            if (wantb && ((uint32_t)__builtin_popcountll(wantb) >= b.refill || !__any(active))) {
                uint64_t s = 0;
                const bool got = feed.take(want, s);
                if (want) {
                    if (!got) exhausted = true;
                    else {
                        const uint64_t o0 = b.offsets[s], o1 = b.offsets[s + 1];
                        len = (uint32_t)(o1 - o0);
                        w_reset(in, o0, len, (uint32_t)s);
                        uint32_t seg = 0;
                        for (uint32_t k = 1; k < b.n_seg; k++)
                            if (s >= b.seg_first[k]) seg = k;
                        aut_load(au, T, b.n_seg ? b.seg_table[seg] : 0u);
                        i = 0; accept = false; active = true;
#if WALK_NODE_MAP
                        for (uint32_t k = 0; k < st.nm_words; k++) reinterpret_cast<WALK_LDS uint32_t*>(st.nm)[k * WALK_WV + wv_lane()] = 0xffffffffu;
#endif
                        n_cur = 1;                                    // the list: (pos 0, start, no cells)  mfa.cpp:217-219
                        Ent<uint32_t, K> e0;
                        e0.P = 0u; e0.vid = aut_start(T, au);
#pragma unroll
                        for (int c = 0; c < K; c++) { e0.S[c] = 0u; e0.L[c] = 0u; e0.F[c] = 0u; }
                        bool f2 = true;
                        store_entry<uint32_t, K>(st, cur, 0u, e0, 0u, f2);
                    }
                }
            }
        }
        if (!__any(active)) break;
        const bool final_pass = (i == len);
        uint32_t ch = 0x100u;
        if (active && !final_pass) ch = w_stream_byte<REV>(in, i);
        uint32_t n_next = 0, shape = 0;
        const bool accept_before = accept;
        in.cq_n = 0u;
        for (;;) {                                                   // (again after the wave has answered what the step asked for)
            n_next = 0; accept = accept_before;
            tb_t TB = tb_init();
            bool f2 = true;
            walk_step<uint32_t, K, REV, TP>(st, T, au, in, cur, n_cur, n_next, i, len, ch, final_pass, active, false, accept, f2, TB, shape);
            if (!__any(active && in.rq != 0u)) break;
            answer_requests<REV>(st, in, active);
#if WALK_NODE_MAP
            map_forget<K>(st, au.vbits(), cur ^ 1u, active, n_next);
#endif
        }
        cur ^= 1u;
        if (active) {
            n_cur = n_next;
            const bool done = accept || final_pass || n_next == 0u;      // mfa.cpp:224-225, 227-235
            i++;
            if (done) { b.results[in.sid] = accept ? 1 : 0; active = false; n_cur = 0u; }
        }
    }
}

}  // namespace mfa_walk

#endif  // MFA_WALK_CORE_H
