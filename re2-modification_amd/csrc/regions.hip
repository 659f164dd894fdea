// Region pre-pass: one streaming read of a batch that finds every string's periodic regions.
//
// The specialised walk kernels (jit_gen.cpp) skip over stretches of input that repeat with a short
// period q <= 8 ("run / period acceleration"); to do that exactly they must know how far such a
// stretch extends, which is the one place on the path where every input byte has to be looked at
// (reference: the reads of str[i] in mfa.cpp:163-166 and the substr compares in mfa.cpp:179-191).
// In round 1 the walk kernels measured the stretches themselves, at 200-256 VGPRs per wave.  This
// kernel does that reading instead: few registers, full occupancy, every byte of the batch fetched
// once, and it leaves a small table per string that the walk kernels look regions up in.
//
// Definitions (offsets relative to the start of the string, memory order):
//   d_q[j] = (s[j] != s[j+q])  for 0 <= j < len - q,  q = 1..8
//   a maximal zero run [a, b) of d_q is the q-periodic REGION [lo, hi) = [a, b + q)
// Table of a string (MFA_REGION_WORDS u64 words): word 0 = header, then up to MFA_REGION_MAX entries
//   entry  = lo | hi << 24 | q << 48
//   header = count | MFA_REGION_OVERFLOW  (overflow: the string has more regions than fit and the table
//            holds the first three and the longest of the others, or the string is too long to be matched at all)
// Guarantees the walk kernels rely on:
//   (1) every entry is true: s[j] == s[j+q] for lo <= j < hi - q;
//   (2) entries with q = 1 are maximal at both ends (cell reads of one-byte-repeated values compare
//       a length with the exact extent of a run, device_common.h: read_pre_u);
//   (3) without the overflow flag, every maximal region that contains at least kCleanMin + 2 whole
//       16-byte blocks is in the table unless a region of a divisor period covers it (to within 16
//       bytes at either end) -- completeness only matters for speed, never for results.
//
// How: one wave per string; lane L of row r looks at the 16-byte block 64 r + L of the string (blocks are
// 16-byte aligned in the batch) plus the 8 bytes behind it, and decides per q whether the block is
// "dirty" (holds a j with d_q[j] = 1).  A ballot per q gives the dirty blocks of the row; scalar code
// keeps, per q, the last dirty block seen and records a candidate (q, X, Y) whenever at least kCleanMin
// clean blocks lie between two dirty ones X < Y.  The exact ends are resolved once per string, all
// candidates at once (lane e re-reads blocks X and Y of candidate e).  Blocks may be declared dirty
// without being so ("forced"): block 0 and the blocks that hold the last 8 bytes.  Forcing never breaks
// (1): the ends are resolved against the real bytes, and a forced block without a real mismatch just
// ends the region at a block boundary (or at the end of the string).
// The block test is twelve V_QSAD_PK_U16_U8 (all eight periods at once, see block_mask); a row whose
// 64 blocks all look like the previous row's costs one compare and one ballot more.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdlib>

#include "mfa_internal.h"

namespace mfa {

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) { set_last_hip_error((int)e_); return MFA_ERR_HIP; } \
    } while (0)

static constexpr int kCleanMin = 4;          // clean 16-byte blocks between two dirty ones that make a candidate
static constexpr uint32_t kMaxLen = 0x00ffffffu;

// bit k set iff byte k of d is non-zero
__device__ __forceinline__ uint32_t nz8(uint64_t d) {
    const uint64_t x = (((d & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | d) & 0x8080808080808080ull;
    return (uint32_t)(((x >> 7) * 0x0102040810204080ull) >> 56);
}

// bytes [s, s+8) of the 16 bytes (hi:lo), 0 <= s <= 8
__device__ __forceinline__ uint64_t shr_bytes(uint64_t lo, uint64_t hi, uint32_t s) {
    return s == 0u ? lo : (s >= 8u ? hi : (lo >> (8u * s)) | (hi << (64u - 8u * s)));
}

// ---- the per-block test ---------------------------------------------------------------------------------
// V_QSAD_PK_U16_U8 compares a 4-byte reference with the four 4-byte windows at byte offsets 0..3 of an 8-byte source and
// ADDS the four sums of absolute differences to four packed 16-bit accumulators.  With the reference = dword m of the
// block and the source = the bytes from dword m (or m + 1) on, one instruction yields, for that dword, the mismatch sums
// for the periods 0..3 (or 4..7); four of them cover a 16-byte block.  A sum is zero iff no byte of the block differs
// from the byte q behind it.  (Sums stay below 16 * 255: no carry between the 16-bit fields.)
__device__ __forceinline__ uint64_t qsad(uint32_t src_lo, uint32_t src_hi, uint32_t ref, uint64_t acc) {
    return __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)src_hi << 32) | src_lo, ref, acc);
}
__device__ __forceinline__ uint32_t pk_nz(uint32_t d) {               // both 16-bit halves -> 0 / 1
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(d), "v"(0x00010001u));
    return r;
}

// bit of period q in a lane's block mask (set = the block is dirty for q)
__device__ __forceinline__ constexpr uint32_t qbit(int q) {
    return q == 1 ? 1u << 16 : q == 2 ? 1u << 1 : q == 3 ? 1u << 17 : q == 4 ? 1u << 2 : q == 5 ? 1u << 18 : q == 6 ? 1u << 3 : q == 7 ? 1u << 19 : 1u << 4;
}
static constexpr uint32_t kAllDirty = 0x000f001eu;

__device__ __forceinline__ uint32_t block_mask(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5) {
    uint64_t a = 0, b = 0;
    a = qsad(w0, w1, w0, a); a = qsad(w1, w2, w1, a); a = qsad(w2, w3, w2, a); a = qsad(w3, w4, w3, a);      // periods 0..3
    b = qsad(w1, w2, w0, b); b = qsad(w2, w3, w1, b); b = qsad(w3, w4, w2, b); b = qsad(w4, w5, w3, b);      // periods 4..7
    const uint32_t d8 = (w0 ^ w2) | (w1 ^ w3) | (w2 ^ w4) | (w3 ^ w5);                                    // period 8
    const uint32_t z0 = pk_nz((uint32_t)a), z1 = pk_nz((uint32_t)(a >> 32)), z2 = pk_nz((uint32_t)b), z3 = pk_nz((uint32_t)(b >> 32));
    const uint32_t z8 = d8 < 1u ? d8 : 1u;
    return ((z0 | (z1 << 1)) | ((z2 << 2) | (z3 << 3)) | (z8 << 4)) & kAllDirty;      // & drops the period-0 field
}

struct RegionState {
    int32_t  last_dirty[9];      // per period: the last dirty block seen (scalar registers after unrolling)
    uint32_t ncand;              // candidates recorded so far (wave-uniform)
    uint32_t shortest;           // with 64 candidates held: the shortest one, length in blocks << 6 | lane (~0u: not known; wave-uniform)
    // candidate e lives in lane e
    uint32_t cq;
    int32_t  cx, cy;
};

// wave-wide minimum by DPP (six V_MIN_U32 with a row shift or a row broadcast as operand, the result read from lane 63): emit() is inlined
// for every period and row in flight, and the usual exchange through the LDS crossbar was 35 instructions at each of its 64 copies
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define MFA_DPP_MIN(ctrl, rows) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false); v = t < v ? t : v; }
    MFA_DPP_MIN(0x111, 0xf)      // row_shr:1   lane i: min over i-1 .. i   (lanes without a source keep their own value)
    MFA_DPP_MIN(0x112, 0xf)      // row_shr:2             i-3 .. i
    MFA_DPP_MIN(0x114, 0xf)      // row_shr:4             i-7 .. i
    MFA_DPP_MIN(0x118, 0xf)      // row_shr:8             i-15 .. i: lane 15 of every row of 16 holds the row's minimum
    MFA_DPP_MIN(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    MFA_DPP_MIN(0x143, 0xc)      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
#undef MFA_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// The candidates recorded first (lanes 0 .. kEarly - 1) are never given up: a walk that fails early -- the usual fate of an attack string
// under an automaton it does not fit -- only ever needs the first stretches, and stepping through them byte by byte is what a table
// without them costs.
static constexpr uint32_t kEarly = 3;
__device__ __forceinline__ uint32_t shortest_candidate(const RegionState& st, uint32_t lane) {
    return wave_min_u32(lane < kEarly ? 0xffffffc0u | lane : ((uint32_t)(st.cy - st.cx) << 6) | lane);          // length in blocks (< 2^20) | lane
}

// Records a candidate.  (A candidate that a candidate of a proper divisor covers says nothing new; most of those are never made -- row_same_as,
// row_inside -- and the others are dropped with the exact ends in hand, finish_string.  Checking here as well kept the last candidate of four
// periods in eight more scalar registers of a kernel that spills them.)
template <int Q>
__device__ __forceinline__ void emit(RegionState& st, uint32_t lane, int32_t x, int32_t y) {
    if (st.ncand < 64u) {
        if (lane == st.ncand) { st.cq = Q; st.cx = x; st.cy = y; }
        st.ncand++;
        st.shortest = ~0u;
        return;
    }
    // More candidates than lanes (text made of hundreds of medium runs): the new one takes the place of the shortest if it is longer,
    // so that a long periodic stretch BEHIND such text is still in the table (the walk jumps over what the table holds and executes
    // everything else step by step).  The table then carries the overflow flag (ncand stays at 64).  The shortest candidate held
    // (length in blocks << 6 | its lane) is kept as a scalar: most candidates of such text are no longer than it and cost one compare.
    if (st.shortest == ~0u) st.shortest = shortest_candidate(st, lane);
    if ((uint32_t)(y - x) <= (st.shortest >> 6)) return;
    if (lane == (st.shortest & 63u)) { st.cq = Q; st.cx = x; st.cy = y; }
    st.shortest = ~0u;
}

// book-keeping of one row for one period: dirty = ballot of dirty blocks, base = block number of lane 0
// returns whether a candidate was recorded (emit called, whatever emit then did with it)
template <int Q>
__device__ __forceinline__ bool book(RegionState& st, uint32_t lane, unsigned long long dirty, int32_t base) {
    int32_t& last = st.last_dirty[Q];
    if (dirty == ~0ull && last == base - 1) { last = base + 63; return false; }      // nothing clean anywhere: the common dirty case
    if (dirty == 0ull) return false;
    bool emitted = false;
    const int32_t first = (int32_t)__builtin_ctzll(dirty), top = 63 - (int32_t)__builtin_clzll(dirty);
    if (base + first - last - 1 >= kCleanMin) { emit<Q>(st, lane, last, base + first); emitted = true; }
    // clean stretches between dirty blocks of this row
    unsigned long long z = ~dirty;
    unsigned long long zz = z;
#pragma unroll
    for (int k = 1; k < kCleanMin; k++) zz &= z >> k;
    const unsigned long long inside = first < top ? ((~0ull >> (63 - top)) & (~0ull << first)) : 0ull;
    // bit s of zz: blocks s .. s + kCleanMin - 1 are clean.  The lowest such bit inside (first, top) starts a stretch (the block before it is dirty),
    // the first dirty block from there on ends it: one trip per stretch that makes a candidate, not one per dirty block
    for (unsigned long long w = zz & inside; w;) {
        const int32_t s = (int32_t)__builtin_ctzll(w);
        const int32_t e = s + (int32_t)__builtin_ctzll(dirty >> s);
        emit<Q>(st, lane, base + s - 1, base + e);
        emitted = true;
        w &= ~0ull << e;
    }
    last = base + top;
    return emitted;
}

// One row for one period.  A run of equal bytes is clean for every period, text of period 2 for 4, 6 and 8 as well: where a proper
// divisor D of Q saw the same dirty blocks in this row and stood at the same block before it, Q's book-keeping would repeat D's
// step by step and every candidate it found would be dropped as covered by D's (finish_string) -- D's position is copied instead.  (Rows
// at the ends of strings and of regions are the ones that get here; with eight full book-keepings each they were most of the
// pass's scalar instructions.)
struct RowBook { unsigned long long dirty[9]; int32_t before[9]; };

template <int Q, int D>
__device__ __forceinline__ bool row_same_as(RegionState& st, RowBook& rb) {
    if constexpr (D >= Q || Q % D != 0) return false;
    else {
        if (rb.dirty[Q] != rb.dirty[D] || rb.before[Q] != rb.before[D]) return false;
        st.last_dirty[Q] = st.last_dirty[D];
        return true;
    }
}

// Where every block that is clean for Q is clean for a proper divisor D as well (this row's, and the stretch reaching into it: D's last dirty
// block is no later than Q's), whatever Q's book-keeping recorded would lie inside D's candidate: Q only notes its last dirty block.  (Text of
// runs of one byte broken by single other bytes: the blocks dirty for period 1 are dirty for every period -- without this all eight periods
// went through their stretches, 8 x the scalar work of such rows, and filled the candidate lanes with covered copies.)
template <int Q, int D>
__device__ __forceinline__ bool row_inside(RegionState& st, RowBook& rb, int32_t base) {
    if constexpr (D >= Q || Q % D != 0) return false;
    else {
        if ((rb.dirty[D] & ~rb.dirty[Q]) != 0ull || rb.before[Q] < rb.before[D] || rb.dirty[Q] == 0ull) return false;
        st.last_dirty[Q] = base + 63 - (int32_t)__builtin_clzll(rb.dirty[Q]);
        return true;
    }
}

template <int Q>
__device__ __forceinline__ void row_prep(RegionState& st, uint32_t mask, uint32_t settled, int32_t base, RowBook& rb) {
    if (settled & qbit(Q)) st.last_dirty[Q] = base - 1;        // every block of the rows skipped before this one was dirty
    rb.dirty[Q] = __ballot((mask & qbit(Q)) != 0u);
    rb.before[Q] = st.last_dirty[Q];
}
template <int Q>
__device__ __forceinline__ void row_period(RegionState& st, uint32_t lane, int32_t base, RowBook& rb) {
    if (row_inside<Q, 1>(st, rb, base)) return;
    if (row_same_as<Q, 4>(st, rb) || row_same_as<Q, 3>(st, rb) || row_same_as<Q, 2>(st, rb) || row_same_as<Q, 1>(st, rb)) return;
    if (row_inside<Q, 2>(st, rb, base) || row_inside<Q, 3>(st, rb, base) || row_inside<Q, 4>(st, rb, base)) return;
    (void)book<Q>(st, lane, rb.dirty[Q], base);
}
// ---- one string ---------------------------------------------------------------------------------------------
struct Geo {                 // where a string lies: blocks are the 16-byte aligned blocks of the batch it touches
    uint64_t a0;             // offset of block 0 in the batch
    uint32_t off0, len;      // the string starts off0 bytes into block 0
    int32_t nblk, endblk, nrows;
    uint32_t lastoff, ylim;  // byte offsets from a0: of the last block, and the largest an 8-byte read behind a block may start at
};

__device__ __forceinline__ Geo make_geo(uint64_t b, uint64_t e, uint64_t ymax) {
    Geo g;
    g.len = (uint32_t)(e - b);
    g.a0 = b & ~(uint64_t)15;
    g.off0 = (uint32_t)(b - g.a0);
    g.nblk = (int32_t)((g.off0 + g.len + 15u) >> 4);
    g.endblk = g.len >= 8u ? (int32_t)((g.off0 + g.len - 8u) >> 4) : 0;
    g.nrows = (g.nblk + 63) >> 6;
    g.lastoff = g.nblk > 0 ? 16u * (uint32_t)(g.nblk - 1) : 0u;
    {   // min(ymax - a0, 0xfffffff0) without a 64-bit comparison (a vector instruction with its constant in two registers: see the kernel)
        const uint64_t d = ymax - g.a0;
        uint64_t top;
        asm("s_lshr_b64 %0, %1, 32" : "=s"(top) : "s"(d) : "scc");
        const uint32_t dl = (uint32_t)d;
        g.ylim = top != 0ull ? 0xfffffff0u : (dl < 0xfffffff0u ? dl : 0xfffffff0u);
    }
    return g;
}

struct Scan {
    RegionState st;
    // Rows whose blocks all carry the same mask as the row before (a stretch inside one periodic region, or text in which every
    // block is dirty for every period) need no book-keeping at all.  `settled` is that mask (~0u: the last row was not uniform),
    // vp the smallest clean period in it (0: none); the dirty periods' last_dirty is caught up when the stretch ends.
    uint32_t settled, vp;
};

__device__ __forceinline__ void scan_reset(Scan& sc) {
#pragma unroll
    for (int q = 0; q < 9; q++) sc.st.last_dirty[q] = -1;
    sc.st.ncand = 0; sc.st.shortest = ~0u; sc.st.cq = 0; sc.st.cx = 0; sc.st.cy = 0;
    sc.settled = ~0u; sc.vp = 0u;
}

// ---- rows in flight -----------------------------------------------------------------------------------------------
// A row is the lane's 16-byte block and the 8 bytes behind it (one load per lane, one more in the last lane).  Rows are read whole, lanes past the last
// block re-read it, and the 8 bytes behind the very last block of the batch come from inside it; what such lanes load is never
// looked at (their blocks are forced).  Addresses are the string's (wave-uniform) base plus a 32-bit offset per lane.
//
// Requests and waits are written out (inline assembly): the row after the one being looked at must stay in flight across the
// trips of the row loop -- eight waves per SIMD with ONE row each are 8 MB on the whole chip, which at 2 us a trip is 4 TB/s.
// (Headline shard: 4.4 ms with the compiler's waits, 4.05 ms like this.)
// Left to the compiler's wait counting, every form of the loop waited for the row just requested somewhere: before a copy of
// freshly loaded registers at the end of a trip, before an address temporary that shares a register with a destination it
// believed pending on some path, or at the loop's exit test.
// What the counted waits rely on (checked in the generated code of this file, and by tests/test_regions_gpu.py on the device):
// every wave has all 64 lanes (256-thread workgroups, no lane leaves early), so both loads of a request are always issued; the
// compiler puts no vector memory operation of its own between a request and the wait for the row before it (the row loop has
// none: no spills -- the kernel needs 38 of the 64 registers its launch bounds allow); and it never copies a row's registers
// between the request and the wait (two or more register sets with constant indices after unrolling: no copies at the loop's
// back edge).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct Row { u32x4 x; u32x2 y; };

__device__ __forceinline__ void row_request(const uint8_t* sbase, const Geo& g, int32_t row, uint32_t lane, Row& r) {
    row = row < g.nrows ? row : g.nrows - 1;
    uint32_t off = (uint32_t)row * 1024u + 16u * lane;
    off = off < g.lastoff ? off : g.lastoff;
    uint32_t yo = off + 16u;
    yo = yo < g.ylim ? yo : g.ylim;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r.x) : "v"(off), "s"(sbase) : "memory");
    // the 8 bytes behind a block are the head of the next lane's block (row_tail); only the last lane has to ask for them
    if (lane == 63u) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(r.y) : "v"(yo), "s"(sbase) : "memory");
}
// y of an arrived row: lane L takes the first two words of lane L + 1's block (DPP wave_shl:1), lane 63 keeps what it loaded.
// (The lane of the string's last block gets its own head this way -- lanes past the last block re-read it -- but the blocks from
// endblk on are forced dirty and their y is never looked at.)
__device__ __forceinline__ uint2 row_tail(const Row& r) {
    return make_uint2((uint32_t)__builtin_amdgcn_update_dpp((int)r.y[0], (int)r.x[0], 0x130, 0xf, 0xf, false),
                      (uint32_t)__builtin_amdgcn_update_dpp((int)r.y[1], (int)r.x[1], 0x130, 0xf, 0xf, false));
}
// r was requested before the YOUNGER rows requested last: once at most their loads (two each) are outstanding, r has arrived
template <int YOUNGER, bool SAFE>
__device__ __forceinline__ void row_wait_older(Row& r) {
    if (SAFE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.x), "+v"(r.y) :: "memory");      // development: wait for everything (MFA_REGION_SAFE_WAITS=1)
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.x), "+v"(r.y) : "n"(2 * YOUNGER) : "memory");
}
__device__ __forceinline__ void row_wait_all(Row& r) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.x), "+v"(r.y) :: "memory"); }

// one row: x = this lane's block, y = the 8 bytes behind it
__device__ __forceinline__ void scan_row(Scan& sc, uint32_t lane, const Geo& g, const int32_t row, const uint4 x, const uint2 y) {
    RegionState& st = sc.st;
    const int32_t base = row * 64, blk = base + (int32_t)lane;
    const bool forced = blk == 0 || blk >= g.endblk;
    // Inside a stretch with a clean period (vp = the smallest one) a row looks like the one before as soon as the 24 bytes every
    // lane holds are vp-periodic: the stretch then continues with the same word, and whether a block is dirty for a period
    // depends on that word only.
    const uint32_t vp = sc.vp;
    if (vp != 0u) {
        uint32_t s0, s1, s2, s3;
        if (vp < 4u) {
            s0 = __builtin_amdgcn_alignbyte(x.y, x.x, vp); s1 = __builtin_amdgcn_alignbyte(x.z, x.y, vp);
            s2 = __builtin_amdgcn_alignbyte(x.w, x.z, vp); s3 = __builtin_amdgcn_alignbyte(y.x, x.w, vp);
        } else if (vp < 8u) {
            s0 = __builtin_amdgcn_alignbyte(x.z, x.y, vp - 4u); s1 = __builtin_amdgcn_alignbyte(x.w, x.z, vp - 4u);
            s2 = __builtin_amdgcn_alignbyte(y.x, x.w, vp - 4u); s3 = __builtin_amdgcn_alignbyte(y.y, y.x, vp - 4u);
        } else { s0 = x.z; s1 = x.w; s2 = y.x; s3 = y.y; }
        uint32_t diff = (x.x ^ s0) | (x.y ^ s1) | (x.z ^ s2) | (x.w ^ s3);
        if (vp < 8u) {                                 // bytes 16..23 among themselves
            const uint64_t yy = ((uint64_t)y.y << 32) | y.x;
            const uint64_t dd = (yy ^ (yy >> (8u * vp))) << (8u * vp);      // the low 8 - vp bytes of the difference
            diff |= (uint32_t)dd | (uint32_t)(dd >> 32);
        }
        if (!__any(forced || diff != 0u)) return;
    }
    const uint32_t mask = forced ? kAllDirty : block_mask(x.x, x.y, x.z, x.w, y.x, y.y);
    if (__all(mask == sc.settled)) return;
    const uint32_t was = sc.settled != ~0u ? sc.settled : 0u;
    RowBook rb;
    row_prep<1>(st, mask, was, base, rb);
    (void)book<1>(st, lane, rb.dirty[1], base);
    // Rows of runs of one byte broken by other bytes (thousands per string in text of that kind, and the rows where a run of the attack
    // corpus ends): every block that is dirty for period 1 is dirty for all periods, and no period's clean stretch reaches further back
    // than period 1's -- row_inside<Q, 1> for the seven other periods, decided before any of them does its book-keeping: they only
    // note a last dirty block.
    // (one vector test instead of seven ballots: a block that is dirty for period 1 and clean for some other period)
    bool inside = rb.dirty[1] != 0ull && !__any((mask & qbit(1)) != 0u && mask != kAllDirty);
    if (inside) {
        int32_t least = 0x7fffffff;
#pragma unroll
        for (int q = 2; q <= 8; q++) { const int32_t bq = (was & qbit(q)) ? base - 1 : st.last_dirty[q]; least = bq < least ? bq : least; }
        inside = least >= rb.before[1];
    }
    if (inside) {
        // where their last dirty block lies inside this row does not matter: every clean block of theirs is clean for period 1, whose
        // candidates say more -- the end of the row is a safe (late) answer
#pragma unroll
        for (int q = 2; q <= 8; q++) st.last_dirty[q] = base + 63;
    } else {
        row_prep<2>(st, mask, was, base, rb); row_period<2>(st, lane, base, rb);
        row_prep<3>(st, mask, was, base, rb); row_period<3>(st, lane, base, rb);
        row_prep<4>(st, mask, was, base, rb); row_period<4>(st, lane, base, rb);
        row_prep<5>(st, mask, was, base, rb); row_period<5>(st, lane, base, rb);
        row_prep<6>(st, mask, was, base, rb); row_period<6>(st, lane, base, rb);
        row_prep<7>(st, mask, was, base, rb); row_period<7>(st, lane, base, rb);
        row_prep<8>(st, mask, was, base, rb); row_period<8>(st, lane, base, rb);
    }
    // What the next rows are compared with is the mask of this row's LAST block: rows that are uniformly like it continue its
    // stretch (a block clean for a period looks 8 bytes beyond itself, so the first bytes of the next row are covered), and a row
    // that is uniformly dirty for a period follows a block that was.  Waiting for a whole uniform row instead cost one more trip
    // through this book-keeping after every string start and every region boundary.
    const int32_t b63 = base + 63;
    sc.settled = (b63 < g.endblk) ? (uint32_t)__builtin_amdgcn_readlane((int)mask, 63) : ~0u;
    // the smallest clean period of that mask (odd periods sit at bits 16.., even ones at bits 1..: qbit)
    const uint32_t clean = ~sc.settled;
    const uint32_t vo = 2u * (uint32_t)__builtin_ctz(((clean >> 16) & 0xfu) | 0x10u) + 1u, ve = 2u * (uint32_t)__builtin_ctz(((clean >> 1) & 0xfu) | 0x10u) + 2u;
    const uint32_t v = vo < ve ? vo : ve;
    sc.vp = v <= 8u ? v : 0u;
}

// A table word written for a reader that may start BEFORE this kernel ends (GATE: the walks of a group are released by a counter, below):
// an agent-scope store goes through to memory (`sc1`) instead of staying dirty in this XCD's L2, which another XCD does not see.
// (written out: as an atomic store of the language it cost the kernel two registers; `tab` is wave-uniform, the word's index per lane)
template <bool GATE> __device__ __forceinline__ void row_store(uint64_t* tab, uint32_t word, uint64_t v) {
    if (GATE) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" :: "v"(word * 8u), "v"(v), "s"(tab) : "memory");
    else tab[word] = v;
}

// exact ends of the candidates, clean-up, table row
struct RowOut { bool entry; uint32_t rank; uint64_t word, header; };      // what a lane contributes to its string's table row
__device__ __forceinline__ RowOut finish_string(Scan& sc, uint32_t lane, const Geo& g, const uint8_t* bytes, uint64_t total16) {
    RegionState& st = sc.st;
    // ---- exact ends: lane c resolves candidate c against the real bytes
    const uint32_t ncand = st.ncand < 64u ? st.ncand : 64u;
    bool keep = false;
    uint32_t lo = 0, hi = 0;
    const uint32_t q = st.cq;
    if (lane < ncand) {
        // positions are byte offsets from a0 (32 bits: a string is shorter than 16 MB); valid j: vlo <= j < vhi
        const int32_t vlo = (int32_t)g.off0, vhi = (int32_t)g.off0 + (int32_t)g.len - (int32_t)q;
        const uint8_t* const sbase = bytes + g.a0;
        auto ld8 = [&](uint32_t off) -> uint64_t {                  // 8 bytes at a0 + off, 0 beyond the batch
            uint64_t v = 0;
            if (off <= g.ylim) v = *reinterpret_cast<const uint64_t*>(sbase + off);
            return v;
        };
        {   // the last real mismatch in block X, or the first valid position of X
            const int32_t w0 = 16 * st.cx;
            const uint64_t p0 = ld8((uint32_t)w0), p1 = ld8((uint32_t)w0 + 8u), p2 = ld8((uint32_t)w0 + 16u);
            uint32_t m = nz8(p0 ^ shr_bytes(p0, p1, q)) | (nz8(p1 ^ shr_bytes(p1, p2, q)) << 8);
            uint32_t valid = 0xffffu;
            if (vlo > w0) valid &= vlo - w0 >= 16 ? 0u : (0xffffu << (uint32_t)(vlo - w0));
            if (vhi < w0 + 16) valid &= vhi <= w0 ? 0u : (0xffffu >> (uint32_t)(w0 + 16 - vhi));
            m &= valid;
            const int32_t lo_abs = m ? w0 + (31 - __builtin_clz(m)) + 1 : (w0 > vlo ? w0 : vlo);
            lo = (uint32_t)(lo_abs - vlo);
        }
        {   // the first real mismatch in blocks Y, Y + 1
            const int32_t w0 = 16 * st.cy;
            const uint64_t p0 = ld8((uint32_t)w0), p1 = ld8((uint32_t)w0 + 8u), p2 = ld8((uint32_t)w0 + 16u), p3 = ld8((uint32_t)w0 + 24u),
                           p4 = ld8((uint32_t)w0 + 32u);
            uint32_t m = nz8(p0 ^ shr_bytes(p0, p1, q)) | (nz8(p1 ^ shr_bytes(p1, p2, q)) << 8) | (nz8(p2 ^ shr_bytes(p2, p3, q)) << 16) |
                         (nz8(p3 ^ shr_bytes(p3, p4, q)) << 24);
            uint32_t valid = 0xffffffffu;
            if (vhi < w0 + 32) valid = vhi <= w0 ? 0u : (0xffffffffu >> (uint32_t)(w0 + 32 - vhi));
            m &= valid;
            int32_t hi_abs;
            if (m) hi_abs = w0 + __builtin_ctz(m) + (int32_t)q;
            else if (vhi < w0 + 32) hi_abs = vlo + (int32_t)g.len;
            else hi_abs = w0 + 32 + (int32_t)q;
            hi = (uint32_t)(hi_abs - vlo);
        }
        keep = hi > lo && hi - lo >= MFA_REGION_MIN_LEN && hi <= g.len;
    }
    // ---- drop regions that a region of a divisor period covers
    for (uint32_t f = 0; f < ncand; f++) {
        // f is wave-uniform: v_readlane (a scalar result a few cycles later), not a trip through the LDS crossbar
        const uint32_t qf = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)f), lof = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)f),
                       hif = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)f);
        const bool kf = __builtin_amdgcn_readlane((int)keep, (int)f) != 0;
        if (!kf) continue;
        const bool divides = ((0x804020108824aaffull >> (((qf - 1u) * 8u + (q - 1u)) & 63u)) & 1ull) != 0ull;     // bit 8 (qf-1) + (q-1): qf divides q
        if (lane < ncand && lane != f && qf < q && divides && lof <= lo + 16u && hif + 16u >= hi) keep = false;
    }
    unsigned long long kb = __ballot(keep);
    const uint32_t total = (uint32_t)__builtin_popcountll(kb);
    if (total > MFA_REGION_MAX) {                          // more than fit: the first kEarly stay (see shortest_candidate), and the longest of the others
        const uint32_t pkey = (lo << 4) | q;
        uint32_t before = 0;
        for (unsigned long long m = kb; m; m &= m - 1ull) {
            const int f = __builtin_ctzll(m);
            const uint32_t kf = (uint32_t)__builtin_amdgcn_readlane((int)pkey, f);
            if (kf < pkey || (kf == pkey && (uint32_t)f < lane)) before++;
        }
        const bool early = keep && before < kEarly;
        const unsigned long long eb = __ballot(early);
        const uint32_t mine = hi - lo;
        uint32_t longer = 0;
        for (unsigned long long m = kb & ~eb; m; m &= m - 1ull) {
            const int f = __builtin_ctzll(m);
            const uint32_t lf = (uint32_t)__builtin_amdgcn_readlane((int)mine, f);
            if (lf > mine || (lf == mine && (uint32_t)f < lane)) longer++;
        }
        keep = keep && (early || longer < MFA_REGION_MAX - (uint32_t)__builtin_popcountll(eb));
        kb = __ballot(keep);
    }
    // entries go out in the order of their starts (the walk kernels copy the first few to LDS: the ones they meet first)
    uint32_t rank = 0;
    {
        const uint32_t key = (lo << 4) | q;
        for (unsigned long long m = kb; m; m &= m - 1ull) {
            const int f = __builtin_ctzll(m);
            const uint32_t kf = (uint32_t)__builtin_amdgcn_readlane((int)key, f);
            if (kf < key || (kf == key && (uint32_t)f < lane)) rank++;
        }
    }
    RowOut out;
    out.header = (uint64_t)(total < MFA_REGION_MAX ? total : MFA_REGION_MAX) | ((total > MFA_REGION_MAX || st.ncand >= 64u) ? MFA_REGION_OVERFLOW : 0ull);
    out.entry = keep && rank < MFA_REGION_MAX;
    out.rank = rank;
    out.word = (uint64_t)lo | ((uint64_t)hi << 24) | ((uint64_t)q << 48);
    return out;
}

// the row goes out.  GATE: in ONE store instruction -- the entries from their lanes, the header from the first lane that holds no entry (at
// most 15 do) -- that goes through to memory
template <bool GATE> __device__ __forceinline__ void row_out(uint64_t* tab, uint32_t lane, const RowOut& r, uint64_t stamp) {
    if (GATE) {
        const uint32_t hl = (uint32_t)__builtin_ctzll(~__ballot(r.entry));
        if (r.entry || lane == hl) row_store<true>(tab, r.entry ? 1u + r.rank : 0u, (r.entry ? r.word : r.header) | stamp);
    } else {
        if (r.entry) tab[1 + r.rank] = r.word;
        if (lane == 0) tab[0] = r.header;
    }
}

// GATE: the strings of a batch are cut into groups (mfa_match_mixed), ONE launch of this kernel scans the whole batch, and the walk
// launches of a group are released as soon as the group's strings are done -- while the kernel is still running.  Three parts:
//   * every word of a row carries the STAMP of the call that wrote it (bits 52..63: the call's number, 12 bits).  A reader takes a word
//     for what it says only if the stamp is the one it expects: a row that has not arrived yet, or a stale copy of the row an earlier call
//     wrote to the same place, is recognised (the walk polls its string's row until the stamp fits: walk_core.h).  So nothing here has to
//     WAIT for a row to reach memory before it says that the row is done (waiting for the store and a returning atomic cost every wave 3 us
//     of its ~20: the gated pass took 3.85 ms against 3.45 without, more than the seven launch tails it saves).
//   * a wave counts its string with ONE fire-and-forget atomic in the counter of (group, string number modulo 64): 64 counters per
//     group, each on a 64-byte line of its own (one counter per group took every string's atomic to ONE address: 12 ns each, 14.5 ms a step).
//   * gate_wait_kernel, one wave launched on the walk stream in front of a group's walk launches, polls the group's 64 counters and
//     ends when they are complete: the walks follow in stream order.  (The command processor can do that wait itself --
//     hipStreamWaitValue32 -- but only on signal memory, which lives on the host side of the link: an atomic there is 0.5 us, per
//     string 630 ms a step.)
// What the kernel needs sits in front of the table, so that no kernel argument stays alive through the row loop (a scalar more there is a
// VGPR more: spilled scalars live in VGPR lanes, and the kernel has 40).  u64 words, k < MFA_GATE_MAX_GROUPS:
//   table[-1 - k]   exclusive end of group k (string index; the last group's is n, unused ones ~0)
//   table[-33]      the call's stamp, already in place (stamp << 52)
//   below table[-96]: the counters, (k, r) at 64-byte line k * 64 + r + 1 going down; zeroed by the host before the launch
__device__ __forceinline__ uint32_t* gate_counter(const uint64_t* table, uint32_t k, uint32_t r) {
    return const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(table - MFA_GATE_FIXED_WORDS)) - 16u * (k * 64u + r + 1u);
}
// strings of [lo, hi) whose number is r modulo 64
__device__ __forceinline__ uint32_t gate_share(uint64_t lo, uint64_t hi, uint32_t r) { return (uint32_t)(((hi + 63u - r) >> 6) - ((lo + 63u - r) >> 6)); }

__device__ __forceinline__ void gate_signal(const uint64_t* table, uint64_t sid, uint32_t lane) {
    uint32_t k = 0;
    while (k + 1u < MFA_GATE_MAX_GROUPS && sid >= table[-1 - (int)k]) k++;      // (scalar loads; groups are few)
    if (lane == 0) __hip_atomic_fetch_add(gate_counter(table, k, (uint32_t)sid & 63u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave: ends when every string of group k has been counted (or after two seconds: a walk that then meets a row without the
// call's stamp polls it for a while and goes on without the row -- slower, never wrong)
__global__ void __launch_bounds__(64) gate_wait_kernel(const uint64_t* table, uint32_t k) {
    const uint32_t r = threadIdx.x & 63u;
    const uint64_t hi = table[-1 - (int)k], lo = k ? table[-(int)k] : 0ull;
    const uint32_t mine = gate_share(lo, hi, r);
    const uint32_t* const c = gate_counter(table, k, r);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= mine)) break;
        if (wall_clock64() - t0 > 200000000ull) break;                 // 100 MHz: 2 s
        __builtin_amdgcn_s_sleep(32);
    }
}

int launch_gate_wait(const uint64_t* d_table, uint32_t group, void* stream) {
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_table, group);
    HIP_TRY(hipGetLastError());
    return MFA_OK;
}

// ---- kernels ---------------------------------------------------------------------------------------------------
constexpr int kRegionDepth = 2;      // 3 and 4 are more robust alone at low occupancy (5 waves per SIMD: 4.26 against 4.87 ms) but slower inside a step: more registers, fewer waves beside a walk wave (5.18-5.32 / 5.33-5.44 / 5.63-5.69 ms)
// MODE 1: streaming phase only (development).
// One wave per string (the hardware dispatcher hands out strings), the next row's loads issued before the current row is looked
// at.  A persistent form (as many waves as the chip holds, strings by ticket, the next string's offsets and first row requested
// before the current string's candidates are resolved) was built and measured: 4.48 ms against 4.30 ms for this one on the
// headline shard, and worse beside the walk kernels -- with eight waves per SIMD the dispatcher's own refill hides a wave's
// start-up latencies as well as software pipelining does.
template <int MODE, int DEPTH, bool SAFE, bool GATE = false>
__global__ void __launch_bounds__(256, 8) region_scan_kernel(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n,
                                                             uint64_t* __restrict__ table, uint32_t rotate) {
    const uint32_t lane = threadIdx.x & 63u;
    // the wave's number as a scalar: string offsets and everything derived from them then live in scalar registers
    const uint32_t wpb = blockDim.x >> 6;                                // waves per workgroup (4; 1 or 2 as a development variant)
    const uint64_t wave = (uint64_t)blockIdx.x * wpb + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = (uint64_t)gridDim.x * wpb;
    const uint64_t total16 = (offsets[n] + 15u) & ~(uint64_t)15;        // the batch is readable below this offset
    const uint64_t ymax = total16 - 8u;
    // Which string a wave takes first is rotated within blocks of 64 by an amount that changes from block to block.  Workgroups go to
    // the eight XCDs in turn, so in a batch whose strings repeat a pattern of costs with a small period (BASELINE configs[2]: every
    // fourth string is text of many short runs, 4 x the scalar work) the expensive strings of EVERY round fell to the same XCDs:
    // 6.8 ms against 3.8 ms for the same batch with two-wave and four-wave workgroups.  (A wave's later strings are n_waves apart:
    // still every string exactly once.)
    uint64_t first = wave;
    if (rotate && (wave | 63ull) < n) first = (wave & ~63ull) | ((wave + 17ull * (wave >> 6)) & 63ull);
    // (with the gate the grid has a wave for every string: no second trip, and nothing that only the next trip needs stays alive)
    for (uint64_t sid = first; sid < n; sid += GATE ? n : n_waves) {
        const uint64_t b = offsets[sid], e = offsets[sid + 1];
        uint64_t* const tab = table + sid * MFA_REGION_WORDS;
        static_assert(kMaxLen == 0x00ffffffu, "the test below is a shift");
        // (the shift is written out: as `e - b > kMaxLen` the test is a 64-bit VECTOR comparison whose constant sits in two vector registers
        // through the whole kernel -- the scalar unit has no 64-bit less-than --, and the compiler turns a shift it can see back into that)
        uint64_t over;
        asm("s_lshr_b64 %0, %1, 24" : "=s"(over) : "s"(e - b) : "scc");
        if (!GATE && over != 0ull) { if (lane == 0) tab[0] = MFA_REGION_OVERFLOW; continue; }
        // (with the gate a string beyond the limit takes the common way out -- no rows, a header that says overflow --: its own store would
        // keep its constant operands in vector registers through the whole kernel)
        const bool too_long = GATE && over != 0ull;
        const Geo g = make_geo(b, e, ymax);
        const uint8_t* const sbase = bytes + g.a0;
        Scan sc;
        scan_reset(sc);
        // DEPTH register sets in rotation, the loop unrolled by DEPTH (all indices are constants after unrolling): while a row is
        // looked at, the DEPTH - 1 rows after it are in flight.  One exit, at the bottom.  (Every string is read to its end, also when it
        // has more candidates than lanes to hold them: emit() then keeps the longest.)
        Row r[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) { r[k].x = u32x4{0, 0, 0, 0}; r[k].y = u32x2{0, 0}; }
        bool more = g.nrows > 0 && !too_long;
        if (more) {
#pragma unroll
            for (int k = 0; k < DEPTH - 1; k++) row_request(sbase, g, k, lane, r[k]);
        }
        for (int32_t row = 0; more; row += DEPTH) {
            bool go = true;
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                if (go) {
                    row_request(sbase, g, row + k + DEPTH - 1, lane, r[(k + DEPTH - 1) % DEPTH]);
                    row_wait_older<DEPTH - 1, SAFE>(r[k]);
                    scan_row(sc, lane, g, row + k, make_uint4(r[k].x[0], r[k].x[1], r[k].x[2], r[k].x[3]), row_tail(r[k]));
                    go = row + k + 1 < g.nrows;
                }
            }
            more = go;
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) row_wait_all(r[k]);             // nothing of this string is in flight any more
        if (MODE == 1) { if (lane == 0) tab[0] = sc.st.ncand; continue; }
        RowOut ro = finish_string(sc, lane, g, bytes, total16);
        if (too_long) { ro.entry = false; ro.header = MFA_REGION_OVERFLOW; }
        row_out<GATE>(tab, lane, ro, GATE ? table[-33] : 0ull);
        if (GATE) gate_signal(table, sid, lane);
    }
}

int launch_region_scan(int n_cus, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t* d_table, void* stream, unsigned threads, bool gate, void* done_event) {
    if (n == 0) { if (done_event) HIP_TRY(hipEventRecord((hipEvent_t)done_event, (hipStream_t)stream)); return MFA_OK; }
    hipStream_t s = (hipStream_t)stream;
    const char* em = getenv("MFA_REGION_MODE");                   // development knobs
    const int mode = em ? atoi(em) : 0;
    const uint64_t cus = (uint64_t)(n_cus > 0 ? n_cus : 256);
    const char* eb = getenv("MFA_REGION_BLOCK");                  // development: threads per workgroup (64, 128 or 256)
    // 256 threads when the pass runs alone (its tuned form); 128 when walk kernels run beside it (mfa_match_mixed): workgroups of two
    // waves find room on SIMDs that walk waves share, and the pass loses less to them (3.80 against 3.92 ms per headline step)
    const unsigned block = eb && (atoi(eb) == 64 || atoi(eb) == 128 || atoi(eb) == 256) ? (unsigned)atoi(eb) : (threads == 128u ? 128u : 256u);
    const uint64_t wpb = block / 64u;
    uint64_t blocks = (n + wpb - 1) / wpb;
    // beyond this waves take several strings each.  Not with the gate: there a wave takes ONE string, so that strings are finished in about
    // the order of their numbers (the dispatcher hands out workgroups in order) and the groups' counters fill up one after the other
    const uint64_t cap = gate ? 0x7fffffffull : cus * 8u * 64u * (4u / wpb);
    if (blocks > cap) blocks = cap;
    const char* el = getenv("MFA_REGION_LDS");                    // development: unused dynamic LDS per workgroup, to lower the occupancy
    const unsigned lds = el ? (unsigned)atoi(el) : 0u;
    const char* ed = getenv("MFA_REGION_DEPTH");                  // development: rows per wave in rotation (2, 3 or 4)
    const int depth = ed ? atoi(ed) : kRegionDepth;
    // development: MFA_REGION_SAFE_WAITS=1 waits for every outstanding load before a row is looked at, instead of counting on the
    // order of the requests (tests/test_regions_gpu.py compares the tables of the two modes)
    const char* er = getenv("MFA_REGION_ROTATE");                 // development: 0 = wave w takes string w
    const uint32_t rotate = er && er[0] == '0' ? 0u : 1u;
    const char* es = getenv("MFA_REGION_SAFE_WAITS");
    const bool safe = es && es[0] == '1';
    if (gate) hipLaunchKernelGGL((region_scan_kernel<0, 2, false, true>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    else if (mode == 1) hipLaunchKernelGGL((region_scan_kernel<1, kRegionDepth, false>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    else if (safe) hipLaunchKernelGGL((region_scan_kernel<0, 2, true>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    // (done_event: the event is the launch's own completion signal -- no packet of its own between this launch and the next one on the stream)
    else if (depth == 2 && done_event) hipExtLaunchKernelGGL((region_scan_kernel<0, 2, false>), dim3((unsigned)blocks), dim3(block), lds, s, nullptr, (hipEvent_t)done_event, 0u, d_bytes, d_offsets, n, d_table, rotate);
    else if (depth == 2) hipLaunchKernelGGL((region_scan_kernel<0, 2, false>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    else if (depth == 4) hipLaunchKernelGGL((region_scan_kernel<0, 4, false>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    else if (depth == 3) hipLaunchKernelGGL((region_scan_kernel<0, 3, false>), dim3((unsigned)blocks), dim3(block), lds, s, d_bytes, d_offsets, n, d_table, rotate);
    else return MFA_ERR_UNSUPPORTED;
    HIP_TRY(hipGetLastError());
    if (done_event && !(depth == 2 && !gate && mode != 1 && !safe)) HIP_TRY(hipEventRecord((hipEvent_t)done_event, s));
    return MFA_OK;
}

}  // namespace mfa
