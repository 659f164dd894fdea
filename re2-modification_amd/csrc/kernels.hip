// HIP kernels for memory-less automata, written for gfx950 (MI355X): Automata::match (reference automata.cpp:177-210) on the
// tabulated step function (image_host.cpp: tabulate_nfa).  One input string per lane, 64 strings per wavefront.
// (Memory automata: regions.hip + walk.hip, or the kernels jit_gen.cpp generates.)
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

// Tabulated automata whose table does not fit LDS (more than 127 state sets): the table stays in global memory -- resident in L2
// up to a few MiB, 16-bit entries up to 65535 state sets and 32-bit entries beyond -- and only the byte classes go to LDS.  One
// string per lane, 32-bit state.
template <bool REV, class T>
__global__ void __launch_bounds__(256)
dfa_big_kernel(const T* __restrict__ trans, const uint8_t* __restrict__ accept_tab, const uint8_t* __restrict__ byte_class,
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
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, pk, (const uint16_t*)ds.d_dfa_trans, ds.d_dfa_accept, ds.d_byte_class,
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
#define BIG_GO(REVV)                                                                                                                                         \
    do {                                                                                                                                                     \
        if (img.dfa_states <= 0xffffu)                                                                                                                       \
            hipLaunchKernelGGL((dfa_big_kernel<REVV, uint16_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)ds.d_dfa_trans, ds.d_dfa_accept,  \
                               ds.d_byte_class, img.n_classes, d_bytes, d_offsets, n, d_results);                                                            \
        else                                                                                                                                                 \
            hipLaunchKernelGGL((dfa_big_kernel<REVV, uint32_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)ds.d_dfa_trans, ds.d_dfa_accept,  \
                               ds.d_byte_class, img.n_classes, d_bytes, d_offsets, n, d_results);                                                            \
    } while (0)
        if (img.h.is_reversed)
            BIG_GO(true);
        else
            BIG_GO(false);
#undef BIG_GO
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
        hipLaunchKernelGGL(dfa_walk_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, (const uint16_t*)ds.d_dfa_trans, ds.d_dfa_accept,
                           ds.d_byte_class, img.dfa_states, img.n_classes, d_bytes, d_offsets, n, d_results);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)dfa_walk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dfa_walk_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, (const uint16_t*)ds.d_dfa_trans, ds.d_dfa_accept,
                           ds.d_byte_class, img.dfa_states, img.n_classes, d_bytes, d_offsets, n, d_results);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)cx.ev_stop, s));
    return MFA_OK;
}

}  // namespace mfa
