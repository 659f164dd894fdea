// The live-list walk kernel (walk_core.h) for gfx950: one string per lane, 64 strings per wave, 4 waves per workgroup that share
// one copy of the automaton tables in LDS.  Table driven: nothing is compiled per automaton, and the lanes of one wave may walk
// DIFFERENT automata (a mixed batch is one launch; mfa_match_mixed).  Compiled once per cell count K (-DWALK_K=1..9).
//
// Persistent waves: a launch starts as many workgroups as the device holds (or the batch needs), the first 64 strings of a wave
// come from its position in the grid, the following ones from a ticket counter.  A lane that finishes its string takes the next
// ticket while the other lanes of the wave go on.
#include <hip/hip_runtime.h>

#include "mfa_internal.h"
#include "walk.h"
#include "walk_core.h"

#ifndef WALK_STATS
#define WALK_STATS 0
#endif
#ifndef WALK_K
#error "compile with -DWALK_K=<cells>"
#endif

namespace mfa {

using namespace mfa_walk;

struct TicketFeeder {
    unsigned long long* counter;
    uint64_t n, first_sid;
    bool first_round = true;
    __device__ __forceinline__ bool take(bool want, uint64_t& sid) {
        const uint32_t lane = threadIdx.x & 63u;
        const unsigned long long wb = __ballot(want);
        unsigned long long first = 0;
        if (first_round) first = first_sid;
        else {
            const int leader = __builtin_ctzll(wb);
            if (lane == (uint32_t)leader) first = atomicAdd(counter, (unsigned long long)__builtin_popcountll(wb));
            first = ((unsigned long long)__shfl((uint32_t)(first >> 32), leader) << 32) | __shfl((uint32_t)first, leader);
            first += (unsigned long long)gridDim.x * blockDim.x;
        }
        first_round = false;
        sid = first + (unsigned long long)__builtin_popcountll(wb & ((1ull << lane) - 1ull));
        return want && sid < n;
    }
};

// LDS of a workgroup: [tables] then per wave [lv][ld][sb][sa][rt_cache]
// TG: the tables stay in global memory (automata whose tables do not fit LDS beside the lists)
template <int K, bool REV, bool STATS, bool TG, int NKEYS>
__global__ void __launch_bounds__(256, WALK_MIN_WAVES)
walk_kernel(WalkArgs a) {
    extern __shared__ uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    if (!TG) {
        for (uint32_t k = threadIdx.x; k < a.table_words; k += blockDim.x) smem[k] = a.tables[k];
        __syncthreads();
    }
    constexpr uint32_t W = Lay<K>::W, DW = Lay<K>::DW;
    const uint32_t CI = a.images_global ? 0u : a.C, XI = a.CX + a.C - CI;      // image entries in LDS / in global memory
    const uint32_t nm_words = WALK_NODE_MAP ? a.nm_words : 0u;      // the lanes' node maps (long-list kernel)
    const uint32_t per_wave = a.C * 64u * (2u * W + 2u * DW) + CI * 64u * (W + DW) + 2u * 64u * MFA_RT_CACHED + nm_words * 64u;
    // the wave's number as a scalar: everything derived from it stays in scalar registers
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    WALK_LDS uint32_t* base = (WALK_LDS uint32_t*)smem + a.shared_words + wave_u * per_wave;
    Store st;
    st.C = a.C; st.CX = a.CX; st.CI = CI;
    st.lv = base; base += 2u * a.C * W * 64u;
    st.ld = base; base += 2u * a.C * DW * 64u;
    st.sb = base; base += CI * W * 64u;
    st.sa = base; base += CI * DW * 64u;
    WALK_LDS uint64_t* const rtc = (WALK_LDS uint64_t*)base; base += 2u * 64u * MFA_RT_CACHED;
    st.nm = (WALK_LDS uint8_t*)base; st.nm_words = nm_words;
    const uint64_t gwave = (uint64_t)blockIdx.x * 4u + wave_u;
    uint32_t* g = a.spill + gwave * ((uint64_t)a.CX * 64u * (2u * W + 2u * DW) + (uint64_t)XI * 64u * (W + DW) + CMP_CACHE * 4u * 64u);
    st.gv = g; g += 2u * a.CX * W * 64u;
    st.gd = g; g += 2u * a.CX * DW * 64u;
    st.gsb = g; g += XI * W * 64u;
    st.gsa = g; g += XI * DW * 64u;
    st.gq = g;
    Batch b{a.bytes, a.offsets, a.n, a.results, a.regions, a.accel, a.refill, a.n_seg, a.seg_first, a.seg_table, a.gate,
            a.lean_queue, reinterpret_cast<uint32_t*>(a.counter + 1)};
    TicketFeeder feed{a.counter, a.n, gwave * 64u};
#if WALK_STATS
    // development build: per-lane counts and per-wave cycle counts, added up in a.counter[8 ..]
    WaveStats ws;
    if (TG) walk_wave<K, REV, TicketFeeder, TablePtrG>(b, a.tables, st, rtc, feed, &ws);
    else walk_wave<K, REV, TicketFeeder, TablePtr>(b, (TablePtr)smem, st, rtc, feed, &ws);
    unsigned long long* out = a.counter + 8;
    const unsigned long long lanev[7] = {ws.dual * 0ull + ws.skipped, ws.probes, ws.hits, ws.steps, ws.spills, ws.strings, 0ull};
    for (int k = 0; k < 6; k++) atomicAdd(&out[k], lanev[k]);
    if ((threadIdx.x & 63u) == 0) {
        const unsigned long long wavev[9] = {ws.iters, ws.dual, ws.t_start, ws.t_byte, ws.t_look, ws.t_plain, ws.t_dual, ws.t_post, ws.t_total};
        for (int k = 0; k < 9; k++) atomicAdd(&out[8 + k], wavev[k]);
        atomicAdd(&out[17], 1ull);
    }
#else
    if (TG) walk_wave<K, REV, TicketFeeder, TablePtrG>(b, a.tables, st, rtc, feed, nullptr);
    else walk_wave<K, REV, TicketFeeder, TablePtr>(b, (TablePtr)smem, st, rtc, feed, nullptr);
#endif
}

// ---- the lean walk: strings without periodic stretches, from the queue the kernel above fills (walk_core.h: walk_wave_lean) -----------------
struct QueueFeeder {
    const uint32_t* queue;
    uint32_t count;
    unsigned long long* counter;
    __device__ __forceinline__ bool take(bool want, uint64_t& sid) {
        const uint32_t lane = threadIdx.x & 63u;
        const unsigned long long wb = __ballot(want);
        const int leader = __builtin_ctzll(wb);
        unsigned long long first = 0;
        if (lane == (uint32_t)leader) first = atomicAdd(counter, (unsigned long long)__builtin_popcountll(wb));
        const uint32_t at = (uint32_t)__shfl((uint32_t)first, leader) + (uint32_t)__builtin_popcountll(wb & ((1ull << lane) - 1ull));
        const bool got = want && at < count;
        sid = got ? queue[at] : 0u;
        return got;
    }
};

// LDS of a workgroup: [tables] then per wave the two lists (values only)
template <int K, bool REV, bool TG, int NKEYS>
__global__ void __launch_bounds__(256, 4)
walk_lean_kernel(WalkArgs a) {
    extern __shared__ uint32_t smem[];
    const uint32_t count = *reinterpret_cast<const uint32_t*>(a.counter + 1);      // final: the kernel that fills the queue has ended
    if (a.lean_seen != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.lean_seen, count + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (count == 0u) return;
    if (!TG) {
        for (uint32_t k = threadIdx.x; k < a.table_words; k += blockDim.x) smem[k] = a.tables[k];
        __syncthreads();
    }
    constexpr uint32_t W = Lay<K>::W;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Store st;
    st.C = a.C; st.CX = a.CX; st.CI = 0u;
    const uint32_t nm_words = WALK_NODE_MAP ? a.nm_words : 0u;
    st.lv = (WALK_LDS uint32_t*)smem + a.shared_words + wave_u * (2u * a.C * W * 64u + nm_words * 64u);
    st.nm = (WALK_LDS uint8_t*)(st.lv + 2u * a.C * W * 64u); st.nm_words = nm_words;
    st.ld = nullptr; st.sb = nullptr; st.sa = nullptr;
    const uint64_t gwave = (uint64_t)blockIdx.x * 4u + wave_u;
    uint32_t* g = a.spill + gwave * ((uint64_t)a.CX * 64u * 2u * W + CMP_CACHE * 4u * 64u);
    st.gv = g; g += 2u * a.CX * W * 64u;
    st.gd = nullptr; st.gsb = nullptr; st.gsa = nullptr;
    st.gq = g;
    Batch b{a.bytes, a.offsets, a.n, a.results, nullptr, 0u, a.refill, a.n_seg, a.seg_first, a.seg_table, 0u, nullptr, nullptr};
    QueueFeeder feed{a.lean_queue, count, a.counter + 2};
    if (TG) walk_wave_lean<K, REV, QueueFeeder, TablePtrG>(b, a.tables, st, feed);
    else walk_wave_lean<K, REV, QueueFeeder, TablePtr>(b, (TablePtr)smem, st, feed);
}

#define WALK_CAT2(a, b) a##b
#define WALK_CAT(a, b) WALK_CAT2(a, b)

// words of LDS one wave needs at capacity C
static size_t wave_words(uint32_t C, bool images_global, uint32_t nm_words = 0) {
    return (size_t)C * 64u * ((images_global ? 2u : 3u) * (Lay<WALK_K>::W + Lay<WALK_K>::DW)) + 2u * 64u * MFA_RT_CACHED + (WALK_NODE_MAP ? (size_t)nm_words * 64u : 0u);
}

#if WALK_STATS
int launch_walk_stats(const WalkLaunch& L, void* stream) {
#elif defined(WALK_LONG_LISTS)
int WALK_CAT(launch_walk_long_k, WALK_K)(const WalkLaunch& L, void* stream) {
#else
int WALK_CAT(launch_walk_k, WALK_K)(const WalkLaunch& L, void* stream) {
#endif
    WalkArgs a = L.args;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)a.shared_words + 4u * wave_words(a.C, a.images_global != 0u, a.nm_words)) * 4u;
    if (lds > 160u * 1024u) return MFA_ERR_UNSUPPORTED;
    hipError_t e = hipSuccess;
#define WALK_GO(REVV, TGV)                                                                                                                   \
    do {                                                                                                                                     \
        e = hipFuncSetAttribute((const void*)walk_kernel<WALK_K, REVV, WALK_STATS != 0, TGV, WALK_KEYS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((walk_kernel<WALK_K, REVV, WALK_STATS != 0, TGV, WALK_KEYS>), dim3(L.grid), dim3(256), lds, s, a);          \
    } while (0)
    if (L.tables_global) { if (L.reversed) WALK_GO(true, true); else WALK_GO(false, true); }
    else { if (L.reversed) WALK_GO(true, false); else WALK_GO(false, false); }
#undef WALK_GO
    if (e == hipSuccess) e = hipGetLastError();
#if !WALK_STATS
    if (e == hipSuccess && L.lean_grid != 0u && a.lean_queue != nullptr) {
        // behind it, on the same stream: the strings it has handed on (the kernel ends at once when there are none)
        WalkArgs al = a;
        al.C = L.lean_C;
        al.CX = a.CX + a.C > al.C ? a.CX + a.C - al.C : 1u;
        const size_t lds_l = ((size_t)a.shared_words + 4u * ((size_t)al.C * 64u * 2u * Lay<WALK_K>::W + (WALK_NODE_MAP ? (size_t)a.nm_words * 64u : 0u))) * 4u;
#define WALK_GO_LEAN(REVV, TGV)                                                                                                              \
    do {                                                                                                                                     \
        e = hipFuncSetAttribute((const void*)walk_lean_kernel<WALK_K, REVV, TGV, WALK_KEYS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l); \
        if (e == hipSuccess) hipLaunchKernelGGL((walk_lean_kernel<WALK_K, REVV, TGV, WALK_KEYS>), dim3(L.lean_grid), dim3(256), lds_l, s, al);          \
    } while (0)
        if (L.tables_global) { if (L.reversed) WALK_GO_LEAN(true, true); else WALK_GO_LEAN(false, true); }
        else { if (L.reversed) WALK_GO_LEAN(true, false); else WALK_GO_LEAN(false, false); }
#undef WALK_GO_LEAN
        if (e == hipSuccess) e = hipGetLastError();
    }
#endif
    if (e != hipSuccess) { set_last_hip_error((int)e); return MFA_ERR_HIP; }
    return MFA_OK;
}

#if !WALK_STATS && !defined(WALK_LONG_LISTS)
size_t WALK_CAT(walk_wave_words_k, WALK_K)(uint32_t C, bool images_global) { return wave_words(C, images_global); }
#endif

}  // namespace mfa
