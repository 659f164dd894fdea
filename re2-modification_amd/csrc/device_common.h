// Device helpers shared by the generic MFA kernel (kernels.hip) and the automaton-specific kernels
// (jit_gen.cpp embeds this file verbatim into every generated source).  gfx950, wave64, one input
// string per lane.
//
// Input bytes are consumed strictly in scan order, one per step, so each lane streams its string
// through two 16-byte register blocks: the block being consumed and its successor, whose load is
// issued a whole block (16 steps) before its first byte is needed.
#ifndef MFA_DEVICE_COMMON_H
#define MFA_DEVICE_COMMON_H

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
};

__device__ __forceinline__ void input_reset(Input& in, uint64_t base, uint32_t len) {
    in.base = base; in.len = len;
    in.blk = ~(uint64_t)0; in.pblk = ~(uint64_t)0;
    in.run_lo = in.run_hi = 0; in.run_ch = 0x100u;
}

// scan index j -> byte offset (reversed automata scan the string backwards: mfa.cpp:163-166)
template <bool REV>
__device__ __forceinline__ uint64_t scan_addr(const Input& in, uint32_t j) {
    return in.base + (REV ? (uint64_t)(in.len - 1u - j) : (uint64_t)j);
}

__device__ __forceinline__ uint4 load16(const uint8_t* bytes, uint64_t blk) {
    return *reinterpret_cast<const uint4*>(bytes + blk);
}

// the byte at scan index i; i advances by one per call
template <bool REV>
__device__ __forceinline__ uint32_t stream_byte(Input& in, uint32_t i) {
    const uint64_t addr = scan_addr<REV>(in, i);
    const uint64_t blk = addr & ~(uint64_t)15;
    if (blk != in.blk) {
        if (blk == in.pblk) { in.w0 = in.p0; in.w1 = in.p1; in.w2 = in.p2; in.w3 = in.p3; }
        else { uint4 d = load16(in.bytes, blk); in.w0 = d.x; in.w1 = d.y; in.w2 = d.z; in.w3 = d.w; }
        in.blk = blk;
        // issue the load of the block that follows in scan order; it is consumed 16 steps from now
        const uint64_t nb = REV ? blk - 16u : blk + 16u;
        const bool ok = REV ? (blk >= 16u && blk > (in.base & ~(uint64_t)15)) : (nb < in.total16 && nb < in.base + in.len);
        in.pblk = ~(uint64_t)0;
        if (ok) { uint4 d = load16(in.bytes, nb); in.p0 = d.x; in.p1 = d.y; in.p2 = d.z; in.p3 = d.w; in.pblk = nb; }
    }
    const uint32_t o = (uint32_t)addr & 15u;
    const uint32_t lo = (o & 4u) ? in.w1 : in.w0;
    const uint32_t hi = (o & 4u) ? in.w3 : in.w2;
    const uint32_t w = (o & 8u) ? hi : lo;
    return (w >> ((o & 3u) * 8u)) & 0xffu;
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

#endif  // MFA_DEVICE_COMMON_H
