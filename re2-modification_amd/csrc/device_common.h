// Device helpers shared by the generic MFA kernel (kernels.hip) and the automaton-specific kernels
// (jit_gen.cpp embeds this file verbatim into every generated source).  gfx950, wave64, one input
// string per lane.
//
// Input bytes are consumed strictly in scan order, one per step, so each lane streams its string
// through two 16-byte register blocks: the block being consumed and its successor, whose load is
// issued a whole block (16 steps) before its first byte is needed.
#ifndef MFA_DEVICE_COMMON_H
#define MFA_DEVICE_COMMON_H

// 16-byte loads in flight per lane and side in the wave-cooperative span comparison.  Two, not eight: eight cost 32 more VGPRs in
// every kernel (their peak), and inside a step of the headline corpus every register a walk wave does not hold is room for
// region-pass waves on its SIMD (measured per step: depth 8 4.86 ms, 4 4.79, 2 4.70, 1 4.65).
#ifndef MFA_SCAN_DEPTH
#define MFA_SCAN_DEPTH 2
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MFA_EMPTY 0xffffffffu
#define F_PRESENT 1u   /* cell exists in the state's memory (automata.h:12)          */
#define F_OPEN    2u   /* Variable::is_open (variable.h:10)                          */
#define F_READ    4u   /* Variable::is_read (variable.h:12)                          */
#define F_UNI     8u   /* every byte of the value equals the byte in bits 8..15      */
#define MFA_DEV_MAX_LEN 0x00ffffffu

struct Input {
    const uint8_t* bytes;   // the whole batch
    uint64_t total16;       // batch size rounded up to 16: loads stay below this offset
    uint64_t base;          // offset of this lane's string
    uint32_t len;
    uint64_t blk;           // offset of the block in w0..w3 (multiple of 16), ~0 = none
    uint32_t w0, w1, w2, w3;
    uint64_t pblk;          // offset of the prefetched block in p0..p3, ~0 = none
    uint32_t p0, p1, p2, p3;
    // cached run of equal bytes: scan[run_lo, run_hi) == run_ch, maximal to the right
    uint32_t run_lo, run_hi, run_ch;
    // cached periodic region: scan[j] == scan[j + per_q] for per_lo <= j < per_hi - per_q (per_q = 0: none)
    uint32_t per_lo, per_hi, per_q;
    // the periodic region known before that one (a cell captured there may be read inside the current one)
    uint32_t prev_lo, prev_hi, prev_q;
    uint32_t dual_p;        // steps per period of the dual step in flight (slope of i), 0 in plain steps
    // region table of this string (regions.hip), nullptr = none: the lane measures regions itself
    const uint64_t* rt;
    uint32_t rt_cnt;
    uint64_t* rt_cache;     // this lane's column of the LDS copy of the header's neighbours (first MFA_RT_CACHED entries)
    uint32_t rt_stride;     // words between consecutive entries of that column
};

#define MFA_NO_WINDOW 0x8000000000000000ull          // addr - MFA_NO_WINDOW >= 16 for every real address
__device__ __forceinline__ void input_drop_window(Input& in) { in.blk = MFA_NO_WINDOW; in.pblk = MFA_NO_WINDOW; }

__device__ __forceinline__ void input_reset(Input& in, uint64_t base, uint32_t len) {
    in.base = base; in.len = len;
    input_drop_window(in);
    in.run_lo = in.run_hi = 0; in.run_ch = 0x100u;
    in.per_lo = in.per_hi = 0; in.per_q = 0; in.dual_p = 0;
    in.prev_lo = in.prev_hi = 0; in.prev_q = 0;
    in.rt = nullptr; in.rt_cnt = 0; in.rt_cache = nullptr; in.rt_stride = 0;
}

// ---- region table (regions.hip; layout in include/mfa_hip.h) -------------------------------------------
#define MFA_RT_WORDS 16u
#define MFA_RT_OVERFLOW 0x100ull

#define MFA_RT_CACHED 2u          /* entries kept in LDS beside the header */
// Attach the table of string `sid`.  Every entry is a true region; when the string had more regions than fit (overflow flag)
// the table holds the longest ones, which only means that the lane walks through the others step by step.
// The header and the first MFA_RT_CACHED entries (all there are, for most strings) are copied to this lane's column of
// `cache` (LDS, [word][column]) with two 16-byte loads that go out together with the loads of the string's offsets; while
// they are at hand the lane also touches the input around both ends of those regions and at both ends of the string:
// that is where it will need bytes next (a jump ends near the end of its region), and a touched line is an L2 hit then
// instead of a trip to HBM through a cold TLB.  Touched words are only xor-ed into `warm`, which nothing depends on.
// rt_fetch issues the two loads (call it before anything waits for the string's offsets), rt_attach uses them.
__device__ __forceinline__ void rt_fetch(const uint64_t* regions, uint64_t sid, uint4& a, uint4& b) {
    a = b = make_uint4(0, 0, 0, 0);
    if (regions == nullptr) return;
    const uint64_t* t = regions + sid * MFA_RT_WORDS;
    a = *reinterpret_cast<const uint4*>(t); b = *reinterpret_cast<const uint4*>(t + 2);
}
__device__ __forceinline__ void rt_attach(Input& in, const uint64_t* regions, uint64_t sid, uint64_t* cache, uint32_t stride, uint32_t& warm,
                                          const uint4 a, const uint4 b) {
    if (regions == nullptr) return;
    const uint64_t* t = regions + sid * MFA_RT_WORDS;
    in.rt = t; in.rt_cnt = a.x & 0xffu; in.rt_cache = cache; in.rt_stride = stride;
    const uint64_t e0 = ((uint64_t)a.w << 32) | a.z, e1 = ((uint64_t)b.y << 32) | b.x;
    cache[0] = e0; cache[stride] = e1;
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

// entry e in scan coordinates: scan[j] == scan[j + q] for lo <= j < hi - q
template <bool REV>
__device__ __forceinline__ void rt_entry(const Input& in, uint32_t e, uint32_t& lo, uint32_t& hi, uint32_t& q) {
    const uint64_t w = e < MFA_RT_CACHED ? in.rt_cache[e * in.rt_stride] : in.rt[1u + e];
    const uint32_t mlo = (uint32_t)w & 0x00ffffffu, mhi = (uint32_t)(w >> 24) & 0x00ffffffu;
    q = (uint32_t)(w >> 48) & 15u;
    lo = REV ? in.len - mhi : mlo;
    hi = REV ? in.len - mlo : mhi;
}

// The region with the smallest period that contains scan index i and leaves room for a probe (the caller's rule:
// hi - i >= 4 q mult + 24).  next = the nearest region start behind i, ~0u if there is none.
template <bool REV>
__device__ __forceinline__ bool rt_find(const Input& in, uint32_t i, uint32_t mult, uint32_t& lo, uint32_t& hi, uint32_t& q, uint32_t& next) {
    bool found = false;
    next = ~0u;
    for (uint32_t e = 0; e < in.rt_cnt; e++) {
        uint32_t l, h, qq;
        rt_entry<REV>(in, e, l, h, qq);
        if (l <= i && i < h) {
            const uint32_t m = qq * mult > 16u ? 1u : mult;
            if (h - i >= 4u * qq * m + 24u && (!found || qq < q)) { found = true; lo = l; hi = h; q = qq; }
        } else if (l > i && l < next) next = l;
    }
    return found;
}

// exclusive end of the run of equal bytes that contains scan index i, if the table has it (entries with q = 1 are
// maximal at both ends); after a miss the caller measures the run (it is short unless the table overflowed)
template <bool REV>
__device__ __forceinline__ bool rt_run(const Input& in, uint32_t i, uint32_t& hi) {
    for (uint32_t e = 0; e < in.rt_cnt; e++) {
        uint32_t l, h, qq;
        rt_entry<REV>(in, e, l, h, qq);
        if (qq == 1u && l <= i && i < h) { hi = h; return true; }
    }
    return false;
}

// scan index j -> byte offset (reversed automata scan the string backwards: mfa.cpp:163-166)
template <bool REV>
__device__ __forceinline__ uint64_t scan_addr(const Input& in, uint32_t j) {
    return in.base + (REV ? (uint64_t)(in.len - 1u - j) : (uint64_t)j);
}

__device__ __forceinline__ uint4 load16(const uint8_t* bytes, uint64_t blk) {
    return *reinterpret_cast<const uint4*>(bytes + blk);
}

// ---- the byte window -----------------------------------------------------------------------------------
// Every lane keeps 16 bytes of its string in registers (w0..w3, the bytes at [blk, blk + 16) of the batch, any alignment) and
// 16 more that it expects to need next (p0..p3 at pblk).  The windows are turned over by the whole wave at once, every 16th
// iteration of the main loop (window_turn): a lane that walks step by step then finds its next 16 bytes already there, asked
// for 16 iterations earlier, and no iteration in between issues a load.  (With windows that every lane renewed when it ran off
// its own, some lane did so in every iteration, and every iteration waited for the load issued in the one before: half of the
// time of a plain walk.)  A lane that starts a string or lands after a jump loads its window on the spot, and with it the one
// it will want at the next turn.

// address of the 16-byte window whose FIRST byte in scan order is scan index j; kept inside [0, total16)
template <bool REV>
__device__ __forceinline__ uint64_t window_addr(const Input& in, uint32_t j) {
    if (REV) { const uint64_t e = in.base + in.len; return e >= (uint64_t)j + 16u ? e - j - 16u : 0; }
    const uint64_t a = in.base + j;
    return a + 16u <= in.total16 ? a : in.total16 - 16u;
}

__device__ __forceinline__ uint4 load16u(const uint8_t* bytes, uint64_t a) {
    uint4 d;
    __builtin_memcpy(&d, bytes + a, 16);
    return d;
}

// every 16th iteration, all lanes: take the window asked for at the last turn, ask for the one after it.
// reading = the lane is inside a string (active, i < len)
template <bool REV>
__device__ __forceinline__ void window_turn(Input& in, uint32_t i, bool reading) {
    if (reading && scan_addr<REV>(in, i) - in.pblk < 16u) { in.w0 = in.p0; in.w1 = in.p1; in.w2 = in.p2; in.w3 = in.p3; in.blk = in.pblk; }
    in.pblk = MFA_NO_WINDOW;
    if (reading && i + 16u < in.len) {
        const uint64_t b = window_addr<REV>(in, i + 16u);
        const uint4 d = load16u(in.bytes, b);
        in.p0 = d.x; in.p1 = d.y; in.p2 = d.z; in.p3 = d.w; in.pblk = b;
    }
}

// the byte at scan index i (i < len); to_turn = iterations until the next window_turn, this one included (1..16)
template <bool REV>
__device__ __forceinline__ uint32_t stream_byte(Input& in, uint32_t i, uint32_t to_turn) {
    const uint64_t addr = scan_addr<REV>(in, i);
    uint64_t o = addr - in.blk;
    if (o >= 16u) {                                      // a string begins, or a jump has landed here
        const uint64_t a = window_addr<REV>(in, i);
        const uint4 d = load16u(in.bytes, a);
        in.pblk = MFA_NO_WINDOW;
        if (i + to_turn < in.len) {                      // where the lane will be at the next turn if it walks on step by step
            const uint64_t b = window_addr<REV>(in, i + to_turn);
            const uint4 e = load16u(in.bytes, b);
            in.p0 = e.x; in.p1 = e.y; in.p2 = e.z; in.p3 = e.w; in.pblk = b;
        }
        in.w0 = d.x; in.w1 = d.y; in.w2 = d.z; in.w3 = d.w; in.blk = a;
        o = addr - a;
    }
    const uint32_t ob = (uint32_t)o;
    const uint32_t lo = (ob & 4u) ? in.w1 : in.w0;
    const uint32_t hi = (ob & 4u) ? in.w3 : in.w2;
    const uint32_t w = (ob & 8u) ? hi : lo;
    return (w >> ((ob & 3u) * 8u)) & 0xffu;
}

// bit k set iff byte k of the 16-byte block differs from c
__device__ __forceinline__ uint32_t mismatch_mask16(uint4 d, uint32_t c) {
    const uint32_t cc = c * 0x01010101u;
    auto nz4 = [](uint32_t t) -> uint32_t {
        uint32_t x = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;   // bit 7 of every non-zero byte
        x >>= 7;
        return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
    };
    return nz4(d.x ^ cc) | (nz4(d.y ^ cc) << 4) | (nz4(d.z ^ cc) << 8) | (nz4(d.w ^ cc) << 12);
}

// smallest offset q in [p, e) with bytes[q] != c, or e
__device__ __forceinline__ uint64_t first_not_equal(const uint8_t* bytes, uint64_t p, uint64_t e, uint32_t c) {
    while (p < e) {
        const uint64_t blk = p & ~(uint64_t)63;
        uint4 d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = (blk + 16u * k < e) ? load16(bytes, blk + 16u * k) : make_uint4(0, 0, 0, 0);
        uint64_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) m |= (uint64_t)mismatch_mask16(d[k], c) << (16 * k);
        const uint32_t skip = (uint32_t)(p - blk);
        m &= ~(uint64_t)0 << skip;
        if (m) {
            const uint64_t q = blk + (uint64_t)__builtin_ctzll(m);
            return q < e ? q : e;
        }
        p = blk + 64u;
    }
    return e;
}

// largest offset q in [lo, hi] with bytes[q] != c, or lo - 1 (callers pass lo >= 1 or test for wrap)
__device__ __forceinline__ int64_t last_not_equal(const uint8_t* bytes, int64_t lo, int64_t hi, uint32_t c) {
    while (hi >= lo) {
        const int64_t blk = hi & ~(int64_t)63;
        uint4 d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = (blk + 16 * k <= hi) ? load16(bytes, (uint64_t)(blk + 16 * k)) : make_uint4(0, 0, 0, 0);
        uint64_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) m |= (uint64_t)mismatch_mask16(d[k], c) << (16 * k);
        const uint32_t top = (uint32_t)(hi - blk);                 // keep bits 0..top
        if (top < 63u) m &= (~(uint64_t)0) >> (63u - top);
        if (m) {
            const int64_t q = blk + 63 - (int64_t)__builtin_clzll(m);
            return q >= lo ? q : lo - 1;
        }
        hi = blk - 1;
    }
    return lo - 1;
}

// exclusive end (scan index) of the run of byte c that starts at scan index i
template <bool REV>
__device__ __forceinline__ uint32_t run_end_from(const Input& in, uint32_t i, uint32_t c) {
    if (i + 1u >= in.len) return in.len;
    if (!REV) {
        const uint64_t q = first_not_equal(in.bytes, in.base + i + 1u, in.base + in.len, c);
        return (uint32_t)(q - in.base);
    }
    const int64_t lo = (int64_t)in.base, hi = (int64_t)(in.base + in.len - 2u - i);
    const int64_t q = last_not_equal(in.bytes, lo, hi, c);
    return q < lo ? in.len : (uint32_t)((int64_t)in.len - 1 - (q - lo));
}

// The same, looking at no more than `limit` positions: false = the run goes on beyond them (the caller then measures it with
// the whole wave).  Runs the region table does not hold are short, so the lanes of a wave that need one measure theirs side
// by side instead of queueing for the wave-wide scan.
template <bool REV>
__device__ __forceinline__ bool run_end_bounded(const Input& in, uint32_t i, uint32_t c, uint32_t limit, uint32_t& end) {
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

// equality of scan[a, a+l) and scan[b, b+l): the two sides of a cell read (mfa.cpp:179-191)
template <bool REV>
__device__ __forceinline__ bool spans_equal(const Input& in, uint32_t a, uint32_t b, uint32_t l) {
    const uint8_t* pa = in.bytes + (REV ? in.base + (in.len - a - l) : in.base + a);   // ascending in memory either way
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

// ---- wave-cooperative scan ----------------------------------------------------------------------------
// Finding how far a run / periodic region extends is the one place where every input byte has to be
// looked at, so it is done at memory speed: all 64 lanes of the wave scan ONE lane's string together,
// 16 bytes per lane per load (1 KiB per wave instruction, contiguous), four loads in flight.
// Returns the exclusive end R (scan index, forward scan) of the q-periodic region starting at i0:
// scan[j] == scan[j+q] for i0 <= j < R - q.  Must be called by all 64 lanes with wave-uniform arguments.
__device__ __forceinline__ uint32_t nz16(uint4 d) {                     // bit k set iff byte k of d is non-zero
    auto nz4 = [](uint32_t t) -> uint32_t {
        uint32_t x = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
        x >>= 7;
        return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
    };
    return nz4(d.x) | (nz4(d.y) << 4) | (nz4(d.z) << 8) | (nz4(d.w) << 12);
}

// the whole blocks of a run: 16 * (index of the first block holding another byte) + the offset of that byte, or ~0.
// One register block per load (nothing to compare with but the run's byte).
template <int D>
__device__ __forceinline__ uint32_t run_blocks_fwd(const uint8_t* p, uint32_t nblk, uint32_t cc, uint32_t lane) {
    for (uint32_t t0 = 0; t0 < nblk; t0 += 64u * D) {
        uint4 x[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            const uint32_t t = t0 + 64u * k + lane;
            x[k] = make_uint4(0, 0, 0, 0);
            if (t < nblk) __builtin_memcpy(&x[k], p + 16u * t, 16);
        }
        bool any_diff = false;
#pragma unroll
        for (int k = 0; k < D; k++)
            any_diff = any_diff || (t0 + 64u * k + lane < nblk && ((x[k].x ^ cc) | (x[k].y ^ cc) | (x[k].z ^ cc) | (x[k].w ^ cc)) != 0u);
        if (__any(any_diff)) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const uint32_t m = t0 + 64u * k + lane < nblk ? nz16(make_uint4(x[k].x ^ cc, x[k].y ^ cc, x[k].z ^ cc, x[k].w ^ cc)) : 0u;
                const unsigned long long b = __ballot(m != 0u);
                if (b) {
                    const int L = __builtin_ctzll(b);
                    return 16u * (t0 + 64u * k + (uint32_t)L) + (uint32_t)__builtin_ctz(__shfl(m, L));
                }
            }
        }
    }
    return ~0u;
}

// Exclusive end (scan index) of the run of equal bytes that starts at scan index i0: all lanes look at
// consecutive 16-byte blocks, a shallow pipeline (MFA_RUN_DEPTH blocks per lane in flight).  The walk kernels call it for cell reads of one-byte-repeated
// values whose run the region table does not hold -- runs shorter than MFA_REGION_MIN_LEN, or any run when there is no table.
#ifndef MFA_RUN_DEPTH
#define MFA_RUN_DEPTH 2
#endif
template <bool REV>
__device__ __forceinline__ uint32_t coop_run_end(const uint8_t* bytes, uint64_t base, uint32_t len, uint32_t i0, uint32_t lane) {
    if (i0 + 1u >= len) return len;
    constexpr int D = MFA_RUN_DEPTH;
    const uint32_t nblk = (len - 1u - i0) >> 4;          // whole blocks of positions i0 .. len-2 (each compared with the run's byte)
    if (!REV) {
        const uint8_t* p = bytes + base;
        const uint32_t cc = (uint32_t)p[i0] * 0x01010101u;
        const uint32_t r = run_blocks_fwd<D>(p + i0, nblk, cc, lane);
        if (r != ~0u) return i0 + r;
        const uint32_t j = i0 + 16u * nblk + lane;       // fewer than 17 positions left
        const unsigned long long b1 = __ballot(lane < 17u && j < len && p[j] != (uint8_t)cc);
        return b1 ? i0 + 16u * nblk + (uint32_t)__builtin_ctzll(b1) : len;
    }
    const uint8_t* top = bytes + base + (len - 1u - i0);  // address of scan index i0; the scan runs towards lower addresses
    const uint32_t cc = (uint32_t)*top * 0x01010101u;
    for (uint32_t t0 = 0; t0 < nblk; t0 += 64u * D) {
        uint4 x[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            const uint32_t t = t0 + 64u * k + lane;
            x[k] = make_uint4(cc, cc, cc, cc);
            if (t < nblk) __builtin_memcpy(&x[k], top - 16u * t - 15u, 16);      // scan indices i0+16t .. i0+16t+15
        }
        bool any_diff = false;
#pragma unroll
        for (int k = 0; k < D; k++) any_diff = any_diff || ((x[k].x ^ cc) | (x[k].y ^ cc) | (x[k].z ^ cc) | (x[k].w ^ cc)) != 0u;
        if (__any(any_diff)) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const uint32_t m = nz16(make_uint4(x[k].x ^ cc, x[k].y ^ cc, x[k].z ^ cc, x[k].w ^ cc));
                const unsigned long long b = __ballot(m != 0u);
                if (b) {
                    const int L = __builtin_ctzll(b);
                    const uint32_t mm = __shfl(m, L);      // the highest address is the first scan index: highest set bit
                    return i0 + 16u * (t0 + 64u * k + (uint32_t)L) + (15u - (31u - (uint32_t)__builtin_clz(mm)));
                }
            }
        }
    }
    const uint32_t j = i0 + 16u * nblk + lane;           // fewer than 17 positions left
    const unsigned long long b1 = __ballot(lane < 17u && j < len && *(bytes + base + (len - 1u - j)) != (uint8_t)cc);
    return b1 ? i0 + 16u * nblk + (uint32_t)__builtin_ctzll(b1) : len;
}

// are the byte ranges [pa, pa+l) and [pb, pb+l) of the batch equal?  All 64 lanes, wave-uniform arguments.
__device__ __forceinline__ bool coop_mem_equal(const uint8_t* bytes, uint64_t pa, uint64_t pb, uint32_t l, uint32_t lane) {
    const uint8_t* a = bytes + pa;
    const uint8_t* b = bytes + pb;
    const uint32_t nblk = l >> 4;
    constexpr int D = MFA_SCAN_DEPTH;
    if (nblk <= 64u) {                                   // up to 1 KiB: one block per lane
        uint4 x = make_uint4(0, 0, 0, 0), y = x;
        if (lane < nblk) { __builtin_memcpy(&x, a + 16u * lane, 16); __builtin_memcpy(&y, b + 16u * lane, 16); }
        if (__any(((x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w)) != 0u)) return false;
    } else {
        for (uint32_t t0 = 0; t0 < nblk; t0 += 64u * D) {
            uint4 x[D], y[D];
#pragma unroll
            for (int k = 0; k < D; k++) {
                // lanes beyond the last block reload it: no branch around the loads, all 2 D of them are in flight together
                const uint32_t t = t0 + 64u * k + lane, tc = t < nblk ? t : nblk - 1u;
                __builtin_memcpy(&x[k], a + 16u * tc, 16); __builtin_memcpy(&y[k], b + 16u * tc, 16);
            }
            bool diff = false;
#pragma unroll
            for (int k = 0; k < D; k++) diff = diff || ((x[k].x ^ y[k].x) | (x[k].y ^ y[k].y) | (x[k].z ^ y[k].z) | (x[k].w ^ y[k].w)) != 0u;
            if (__any(diff)) return false;
        }
    }
    const uint32_t j = 16u * nblk + lane;
    return !__any(lane < 16u && j < l && a[j] != b[j]);
}

// ---- plain and dual values ---------------------------------------------------------------------------
// The automaton-specific kernels run their step function in two instantiations: on plain uint32_t, and
// on "dual" values (v, d) = value now and its change per step.  While a lane sits inside a long run of
// equal input bytes its slot configuration usually evolves affinely (pos and open-cell lengths grow by
// one per step, everything else is constant).  The dual step computes, next to the new state, (1) how
// the per-step change propagates through the step, and (2) for every comparison it makes, after how
// many steps its outcome would flip if the state kept moving along d.  If the change reproduces itself
// the lane jumps over all steps whose outcomes are already known (jit_gen.cpp, "run acceleration").
struct Dual { uint32_t v; int32_t d; };

template <class U> __device__ __forceinline__ U konst(uint32_t x);
template <> __device__ __forceinline__ uint32_t konst<uint32_t>(uint32_t x) { return x; }
template <> __device__ __forceinline__ Dual konst<Dual>(uint32_t x) { return Dual{x, 0}; }

__device__ __forceinline__ uint32_t val(uint32_t a) { return a; }
__device__ __forceinline__ uint32_t val(Dual a) { return a.v; }

// TB: for how many consecutive steps (this one included) do all comparisons made so far keep their
// outcome?  Every comparison contributes a bound of the form ceil(a / b), a, b > 0, and the minimum of
// ceilings is the ceiling of the minimum, so TB is kept as the smallest fraction seen (compared by cross
// multiplication) and divided once per step.
struct tb_t { int64_t a; uint32_t b; };                              // a / b, 0 < a < 2^34, 0 < b <= 2^20
__device__ __forceinline__ tb_t tb_init() { return tb_t{(int64_t)1 << 30, 1u}; }
__device__ __forceinline__ void tb_min(tb_t& TB, int64_t a, int64_t b) {
    if (b > ((int64_t)1 << 20) || a > ((int64_t)1 << 34)) {           // absurd slope or distance: give up on this step / no constraint
        if (b > ((int64_t)1 << 20)) { a = 1; b = 1; } else return;
    }
    // a / b < TB.a / TB.b  <=>  a * TB.b < TB.a * b   (64 x 32 bit products, < 2^54)
    if ((uint64_t)a * TB.b < (uint64_t)TB.a * (uint32_t)b) { TB.a = a; TB.b = (uint32_t)b; }
}
__device__ __forceinline__ void tb_min(tb_t& TB, int64_t t) { tb_min(TB, t, 1); }
__device__ __forceinline__ int64_t tb_steps(const tb_t& TB) { return (TB.a + TB.b - 1) / TB.b; }
__device__ __forceinline__ bool tb_is_one(const tb_t& TB) { return TB.a <= (int64_t)TB.b; }   // tb_steps(TB) <= 1, without dividing

// a < b now; bounds TB by the number of steps for which the outcome stays the same
__device__ __forceinline__ bool lt(uint32_t a, uint32_t b, tb_t&) { return a < b; }
__device__ __forceinline__ bool lt(Dual a, Dual b, tb_t& TB) {
    const bool o = a.v < b.v;
    const int64_t diff = (int64_t)b.v - (int64_t)a.v, s = (int64_t)b.d - (int64_t)a.d;   // f(t) = diff + t*s, o <=> f > 0
    if (o) { if (s < 0) tb_min(TB, diff, -s); }                       // first t with f(t) <= 0: ceil(diff / -s)
    else   { if (s > 0) tb_min(TB, 1 - diff, s); }                    // first t with f(t) > 0: floor(-diff / s) + 1 = ceil((1 - diff) / s)
    return o;
}
template <class U> __device__ __forceinline__ bool gt(U a, U b, tb_t& TB) { return lt(b, a, TB); }
template <class U> __device__ __forceinline__ bool ge(U a, U b, tb_t& TB) { return !lt(a, b, TB); }
template <class U> __device__ __forceinline__ bool le(U a, U b, tb_t& TB) { return !lt(b, a, TB); }

__device__ __forceinline__ bool eq(uint32_t a, uint32_t b, tb_t&) { return a == b; }
__device__ __forceinline__ bool eq(Dual a, Dual b, tb_t& TB) {
    const bool o = a.v == b.v;
    const int64_t diff = (int64_t)b.v - (int64_t)a.v, s = (int64_t)b.d - (int64_t)a.d;
    if (s != 0) {
        if (o) tb_min(TB, 1);
        else if ((diff < 0) == (s > 0)) tb_min(TB, diff < 0 ? -diff : diff, s < 0 ? -s : s);   // may meet at |diff|/|s| (conservative if it does not divide)
    }
    return o;
}
template <class U> __device__ __forceinline__ bool ne(U a, U b, tb_t& TB) { return !eq(a, b, TB); }

__device__ __forceinline__ uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
__device__ __forceinline__ Dual sel(bool c, Dual a, Dual b) { return Dual{c ? a.v : b.v, c ? a.d : b.d}; }
__device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ Dual add(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) { return a - b; }
__device__ __forceinline__ Dual sub(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }

// slot key P = pos << 4 | first cell name
__device__ __forceinline__ uint32_t mkp(uint32_t pos, uint32_t fname) { return (pos << 4) | fname; }
__device__ __forceinline__ Dual mkp(Dual pos, uint32_t fname) { return Dual{(pos.v << 4) | fname, pos.d * 16}; }
__device__ __forceinline__ uint32_t posof(uint32_t P, tb_t&) { return P >> 4; }
__device__ __forceinline__ Dual posof(Dual P, tb_t& TB) {
    if (P.d & 15) tb_min(TB, 1);                                      // the first-cell name would change: not affine
    return Dual{P.v >> 4, P.d / 16};
}
// flag words do not move affinely: a dual flag word must be constant
__device__ __forceinline__ uint32_t flagv(uint32_t F, tb_t&) { return F; }
__device__ __forceinline__ uint32_t flagv(Dual F, tb_t& TB) { if (F.d != 0) tb_min(TB, 1); return F.v; }

// does scan[i, i+l) equal the cell value (start, l, flags)?  ch = scan[i].   (mfa.cpp:178-187)
template <bool REV>
__device__ __forceinline__ bool read_matches(Input& in, uint32_t i, uint32_t ch, uint32_t start, uint32_t l, uint32_t fl) {
    if (in.len - i < l) return false;
    if (l == 0u) return true;
    if (fl & F_UNI) {                    // the value is one byte repeated: compare with the run of bytes at i
        const uint32_t c = (fl >> 8) & 0xffu;
        if (c != ch) return false;
        if (l == 1u) return true;
        if (!(in.run_ch == c && in.run_lo <= i && i < in.run_hi)) {
            in.run_hi = run_end_from<REV>(in, i, c);
            in.run_lo = i; in.run_ch = c;
        }
        return in.run_hi - i >= l;
    }
    if (((fl >> 8) & 0xffu) != ch) return false;          // first byte of the value (kept in the flags) against the byte at i
    return spans_equal<REV>(in, start, i, l);
}

// MFA::doMemoryWriteActions (mfa.cpp:80-105) for one cell (S,L,F) = (start, len, flags); the text just
// consumed is scan[ts, ts+tl), all of whose bytes equal tch iff tuni
__device__ __forceinline__ void act_open(uint32_t& S, uint32_t& L, uint32_t& F, uint32_t ts, uint32_t tl, bool tuni, uint32_t tch) {
    S = ts; L = tl; F = F_PRESENT | F_OPEN | (tuni ? F_UNI : 0u) | (tch << 8);      // create if absent, open(), write(t)
}
__device__ __forceinline__ void act_close(uint32_t& S, uint32_t& L, uint32_t& F) { F &= ~F_OPEN; }   // close(); absent stays absent
__device__ __forceinline__ void act_none(uint32_t& S, uint32_t& L, uint32_t& F, uint32_t ts, uint32_t tl, bool tuni, uint32_t tch) {
    const bool w = (F & (F_PRESENT | F_OPEN)) == (F_PRESENT | F_OPEN) && tl != 0u;   // write(t) when open
    const bool was_empty = L == 0u;
    const uint32_t fch = (F >> 8) & 0xffu;
    const bool uni = was_empty ? tuni : ((F & F_UNI) && tuni && fch == tch);
    const uint32_t nf = (F & (F_PRESENT | F_OPEN | F_READ)) | (uni ? F_UNI : 0u) | ((was_empty ? tch : fch) << 8);
    S = (w && was_empty) ? ts : S;
    L = w ? L + tl : L;
    F = w ? nf : F;
}

// ---- the same helpers over plain or dual values (generated kernels) ------------------------------------
// end of the run of equal bytes that contains i, as a value of type U.  In a dual step over a periodic
// region with period > 1 the run structure repeats every period: the end moves with i.
__device__ __forceinline__ uint32_t run_hi_u(const Input& in, uint32_t, tb_t&) { return in.run_hi; }
__device__ __forceinline__ Dual run_hi_u(const Input& in, Dual, tb_t& TB) {
    if (in.per_q <= 1u || in.dual_p == 0u) return Dual{in.run_hi, 0};
    Dual r{in.run_hi, (int32_t)in.dual_p};
    (void)lt(r, Dual{in.per_hi, 0}, TB);                 // exact only while the shifted run ends inside the periodic region
    return r;
}

// byte-wise comparison of a cell value with the text at i: is its outcome the same in every period?
__device__ __forceinline__ void spans_period_bound(const Input&, uint32_t, uint32_t, uint32_t, tb_t&) {}
__device__ __forceinline__ void spans_period_bound(const Input& in, Dual i, Dual start, Dual l, tb_t& TB) {
    const bool ok = in.dual_p != 0u && in.per_q != 0u && l.d == 0 && start.d >= 0 && (uint32_t)start.d % in.per_q == 0u &&
                    start.v >= in.per_lo && i.v >= in.per_lo;
    if (!ok) { tb_min(TB, 1); return; }
    (void)le(add(i, l), Dual{in.per_hi, 0}, TB);         // both sides stay inside the periodic region: same bytes every period
    if (start.d != 0) (void)le(add(start, l), Dual{in.per_hi, 0}, TB);
}

// does scan[s, s+l) lie inside one of the two periodic regions the lane knows?  q = that region's period
__device__ __forceinline__ bool span_in_region(const Input& in, uint32_t s, uint32_t l, uint32_t& q) {
    if (in.per_q != 0u && s >= in.per_lo && s + l <= in.per_hi) { q = in.per_q; return true; }
    if (in.prev_q != 0u && s >= in.prev_lo && s + l <= in.prev_hi) { q = in.prev_q; return true; }
    return false;
}

// would the cell read below have to look for the end of the run of bytes at i?  (then the caller finds it
// with the whole wave first, so that read_pre_u never scans memory from a single lane)
__device__ __forceinline__ bool uni_needs_run(const Input& in, uint32_t i, uint32_t ch, uint32_t l, uint32_t fl) {
    if (!(fl & F_UNI) || l <= 1u || in.len - i < l) return false;
    const uint32_t c = (fl >> 8) & 0xffu;
    return c == ch && !(in.run_ch == c && in.run_lo <= i && i < in.run_hi);
}

// First half of a cell read for the generated kernels: everything that needs no byte-wise comparison.
// Returns the outcome, or sets `need_cmp` when scan[i, i+l) has to be compared with scan[start, start+l)
// byte by byte -- the caller then runs that comparison with the whole wave (coop_mem_equal).
template <bool REV, class U>
__device__ __forceinline__ bool read_pre_u(Input& in, U i, uint32_t ch, U start, U l, uint32_t fl, tb_t& TB, bool& need_cmp) {
    if (lt(sub(konst<U>(in.len), i), l, TB)) return false;
    if (eq(l, konst<U>(0u), TB)) return true;
    if (fl & F_UNI) {
        const uint32_t c = (fl >> 8) & 0xffu;
        if (c != ch) return false;
        if (eq(l, konst<U>(1u), TB)) return true;
        const uint32_t iv = val(i);
        if (!(in.run_ch == c && in.run_lo <= iv && iv < in.run_hi)) {
            in.run_hi = run_end_from<REV>(in, iv, c);
            in.run_lo = iv; in.run_ch = c;
        }
        return ge(sub(run_hi_u(in, i, TB), i), l, TB);
    }
    // bits 8..15 of the flags always hold the FIRST byte of a non-empty value (act_open_u / act_none_u): a read whose
    // first byte already differs from the byte at i fails without touching memory
    if (((fl >> 8) & 0xffu) != ch) return false;
    spans_period_bound(in, i, start, l, TB);
    // the first 16 bytes per lane (all lanes at once); only reads that survive them go to the wave-wide comparison
    const uint32_t lv = val(l), head = lv < 16u ? lv : 16u;
    if (!spans_equal<REV>(in, REV ? val(start) + (lv - head) : val(start), REV ? val(i) + (lv - head) : val(i), head)) return false;
    if (lv <= 16u) return true;
    // two q-periodic spans (q <= 8) whose first 16 bytes agree are equal: nothing more to read.  Plain steps only
    // (lanes in a dual period have to bound the outcome over the periods to come: spans_period_bound above).
    if (in.dual_p == 0u) {
        uint32_t qa = 0, qb = 0;
        if (span_in_region(in, val(start), lv, qa) && span_in_region(in, val(i), lv, qb) && qa == qb) return true;
    }
    need_cmp = true;
    return false;
}

template <class U>
__device__ __forceinline__ void act_open_u(U& S, U& L, U& F, U ts, U tl, bool tuni, uint32_t tch) {
    S = ts; L = tl; F = konst<U>(F_PRESENT | F_OPEN | (tuni ? F_UNI : 0u) | (tch << 8));
}
template <class U>
__device__ __forceinline__ void act_close_u(U& S, U& L, U& F, tb_t& TB) { F = konst<U>(flagv(F, TB) & ~F_OPEN); }
template <class U>
__device__ __forceinline__ void act_none_u(U& S, U& L, U& F, U ts, U tl, bool tuni, uint32_t tch, tb_t& TB) {
    const uint32_t f = flagv(F, TB);
    const bool w = (f & (F_PRESENT | F_OPEN)) == (F_PRESENT | F_OPEN) && !eq(tl, konst<U>(0u), TB);
    if (!w) return;                                                   // callers pass lanes under a wave-level guard; per-lane skip
    const bool was_empty = eq(L, konst<U>(0u), TB);
    const uint32_t fch = (f >> 8) & 0xffu;
    const bool uni = was_empty ? tuni : ((f & F_UNI) && tuni && fch == tch);
    S = sel(was_empty, ts, S);
    L = add(L, tl);
    F = konst<U>((f & (F_PRESENT | F_OPEN | F_READ)) | (uni ? F_UNI : 0u) | ((was_empty ? tch : fch) << 8));
}

#endif  // MFA_DEVICE_COMMON_H
