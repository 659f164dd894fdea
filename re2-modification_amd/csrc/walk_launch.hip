// Host side of the live-list walk: launch planning, and the mixed-batch object of the C-ABI (mfa_mixed_*).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "mfa_internal.h"
#include "walk.h"

namespace mfa {

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) { set_last_hip_error((int)e_); return MFA_ERR_HIP; } \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

int walk_mode() {
    const char* e = getenv("MFA_WALK");
    return !e ? 0 : e[0] == 't' ? 1 : e[0] == 'j' ? 2 : 0;     // "table" / "jit"; anything else: automatic
}

static size_t wave_words(uint32_t K, uint32_t C, bool ig, uint32_t nm_words = 0) {
    if (nm_words != 0u && K == 1u) return walk_wave_words_k1(C, ig) + (size_t)nm_words * 64u;      // the long-list kernel's node maps
    switch (K) {
        case 1: return walk_wave_words_k1(C, ig); case 2: return walk_wave_words_k2(C, ig); case 3: return walk_wave_words_k3(C, ig);
        case 4: return walk_wave_words_k4(C, ig); case 5: return walk_wave_words_k5(C, ig); case 6: return walk_wave_words_k6(C, ig);
        case 7: return walk_wave_words_k7(C, ig); case 8: return walk_wave_words_k8(C, ig); default: return walk_wave_words_k9(C, ig);
    }
}

// One launch over the sub-batch [d_offsets[0], d_offsets[n]).  The LDS capacity C of the lists: enough for the automata's
// longest possible list if that leaves room for two workgroups per CU, else what does (longer lists spill to `d_spill`).
int launch_walk(const WalkPlanInput& p, const uint32_t* d_tables, int n_cus, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                uint8_t* d_results, const uint64_t* d_regions, uint32_t n_seg, const uint32_t* seg_first, const uint32_t* seg_table,
                uint32_t** d_spill, size_t* spill_bytes, unsigned long long* d_counter, void* stream, uint32_t gate, LeanHint* lean, void* wait_event) {
    if (n == 0) { if (wait_event) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_event, 0)); return MFA_OK; }
    if (n_seg == 0 || n_seg > WALK_MAX_SEG || n > 0xffffffffull) return MFA_ERR_INVALID_ARG;
    WalkLaunch L;
    WalkArgs& a = L.args;
    std::memset(&a, 0, sizeof a);
    a.bytes = d_bytes; a.offsets = d_offsets; a.n = n; a.results = d_results; a.regions = d_regions; a.tables = d_tables;
    a.counter = d_counter;
    a.gate = gate;
    a.table_words = p.table_words;
    a.shared_words = (p.table_words + 63u) & ~63u;
    // tables beyond a third of the LDS (or forced: development) stay in global memory
    L.tables_global = a.shared_words > 160u * 1024u / 4u / 3u || getenv("MFA_WALK_TABLES_GLOBAL") != nullptr;
    if (L.tables_global) a.shared_words = 0;
    a.n_seg = n_seg;
    for (uint32_t k = 0; k <= n_seg; k++) a.seg_first[k] = seg_first[k];
    for (uint32_t k = 0; k < n_seg; k++) a.seg_table[k] = seg_table[k];
    const char* ae = getenv("MFA_ACCEL");
    a.accel = (ae && ae[0] == '0') ? 0u : 1u;
    a.refill = (uint32_t)std::max(1, std::min(64, env_int("MFA_WALK_REFILL", 1)));
    // capacity
    const bool ig = env_int("MFA_WALK_IMAGES_GLOBAL", 0) != 0;
    a.images_global = ig ? 1u : 0u;
    const size_t lds_max = 160u * 1024u / 4u;                 // words
    const int want_c = env_int("MFA_WALK_C", 0);
    const uint32_t c_cap = want_c > 0 ? (uint32_t)want_c : 8u;
    uint32_t C = std::min(p.max_live, c_cap);
    if (C < 1) C = 1;
    // two workgroups per CU when the batch fills the device; a batch that does not even give every CU one workgroup leaves the LDS to that
    // one: longer lists stay in LDS (the 77-node automata: lists of 10, three entries of them in LDS at two workgroups per CU)
    const uint64_t cus_ = (uint64_t)(n_cus > 0 ? n_cus : 256);
    const uint32_t wgs_goal = (uint32_t)env_int("MFA_WALK_WGS", (n + 255) / 256 <= cus_ ? 1 : 2);
    // one-cell automata with long lists (the 77-node ex. 8 -bnf / -reverse) have a kernel of their own, which finds a node's entry through a
    // per-lane map in LDS: one byte per node and lane (automata of up to 128 nodes; beyond them the general kernel's key search)
    const bool long_lists = p.K == 1 && p.max_live > 16u && p.max_live < 128u && env_int("MFA_WALK_LONG", 1) != 0 && getenv("MFA_WALK_STATS") == nullptr;
    a.nm_words = long_lists ? (p.max_live + 1u + 3u) / 4u : 0u;
    while (C > 1 && want_c <= 0 && a.shared_words + 4u * wave_words(p.K, C, ig, a.nm_words) > lds_max / wgs_goal) C--;
    while (C > 1 && a.shared_words + 4u * wave_words(p.K, C, ig, a.nm_words) > lds_max) C--;
    if (a.shared_words + 4u * wave_words(p.K, C, ig, a.nm_words) > lds_max) return MFA_ERR_UNSUPPORTED;      // the tables alone fill the LDS
    a.C = C;
    a.CX = p.max_live > C ? p.max_live - C : 1u;
    const size_t lds_words = a.shared_words + 4u * wave_words(p.K, C, ig, a.nm_words);
    uint64_t per_cu = lds_max / lds_words;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int cap_waves = env_int("MFA_WALK_WAVES_PER_CU", 0);      // development knob (multiples of 4)
    if (cap_waves > 0 && (uint64_t)(cap_waves + 3) / 4 < per_cu) per_cu = (uint64_t)(cap_waves + 3) / 4;
    uint64_t grid = (uint64_t)(n_cus > 0 ? n_cus : 256) * per_cu, want = (n + 255) / 256;
    {   // development: a fraction of the workgroups the device holds (fewer walk waves beside the region pass, each taking more tickets)
        const int pct = env_int("MFA_WALK_GRID_PCT", 100);
        if (pct > 0 && pct < 100) grid = std::max<uint64_t>(1, grid * (uint64_t)pct / 100u);
    }
    if (grid > want) grid = want;
    // What a wave may spill (list entries and probe images beyond the LDS capacity) is sized for the worst case -- every node of the launch's
    // largest automaton alive at once -- per wave of the grid: 4.7 MB per wave for 1024 nodes and one cell.  The grid shrinks (the waves are
    // persistent: fewer of them take more tickets each) until that fits a budget, 2 GiB by default; what does not fit with ONE workgroup is MFA_ERR_NOMEM.
    const uint32_t W = 2 + 2 * p.K, DW = (1 + 2 * p.K + 1) / 2;
    const size_t per_wave = ((size_t)(a.CX + a.C) * 64u * (3u * W + 3u * DW) + 4u * 4u * 64u) * sizeof(uint32_t);      // + the comparison answers (walk_core.h: CMP_CACHE)
    const size_t budget = (size_t)std::max(1, env_int("MFA_WALK_SPILL_MB", 2048)) << 20;
    while (grid > 1 && grid * 4u * per_wave > budget) grid = (grid + 1) / 2;
    if (grid * 4u * per_wave > budget) return MFA_ERR_NOMEM;
    L.grid = (unsigned)grid;
    L.reversed = p.reversed;
    // the lean kernel behind it (strings without periodic stretches: walk.hip): the plain step only, lists of the same capacity, four
    // workgroups per CU where the LDS allows; its waves' spill areas and the queue of string numbers share the buffer with this launch's
    L.lean_grid = 0; L.lean_C = a.C;
    size_t lean_bytes = 0, queue_at = 0;
    // (An EMPTY lean launch is not free beside a region pass: its workgroups -- 128 VGPRs, LDS for the tables and the lists -- queue for room like any
    // other; 0.1-0.35 ms of the walk stream's time were measured for launches that had nothing to do.  So a launch whose slot last reported an empty
    // queue leaves the lean kernel and the queue out, and looks again every 32nd time only, with a quarter of the grid.)
    bool want_lean = d_regions != nullptr && a.accel != 0u && env_int("MFA_WALK_LEAN", 1) != 0;
    bool lean_probe = false;
    if (want_lean && lean != nullptr) {
        if (lean->h_seen == nullptr && hipHostMalloc((void**)&lean->h_seen, sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) *lean->h_seen = 0u;
        if (lean->h_seen == nullptr) (void)hipGetLastError();
        else {
            const uint32_t seen = *(volatile uint32_t*)lean->h_seen;      // what the last lean kernel of this slot that has ended found (+ 1; 0: none has reported yet)
            lean->quiet = seen == 1u ? lean->quiet + 1u : 0u;
            if (lean->quiet >= 1u && env_int("MFA_WALK_LEAN", 1) != 2) {      // (MFA_WALK_LEAN=2: always)
                if ((lean->launches & 31u) != 0u) want_lean = false; else lean_probe = true;
            }
            if (getenv("MFA_VERBOSE")) fprintf(stderr, "mfa_hip: table walk of %llu strings: last lean queue seen %d, launch %u: lean kernel %s\n",
                                               (unsigned long long)n, (int)seen - 1, lean->launches, want_lean ? (lean_probe ? "on (a look)" : "on") : "left out");
            lean->launches++;
            a.lean_seen = lean->h_seen;
        }
    }
    if (want_lean) {
        const size_t lean_wave_words = (size_t)a.C * 64u * 2u * W + (size_t)a.nm_words * 64u;
        uint64_t lean_per_cu = lds_max / (a.shared_words + 4u * lean_wave_words);
        if (lean_per_cu > 4) lean_per_cu = 4;
        if (lean_per_cu >= 1) {
            uint64_t lg = (uint64_t)(n_cus > 0 ? n_cus : 256) * lean_per_cu;
            if (lean_probe) lg = std::max<uint64_t>(1, lg / 4u);
            if (lg > want) lg = want;
            const size_t lean_per_wave = ((size_t)(a.CX + a.C) * 64u * 2u * W + 4u * 4u * 64u) * sizeof(uint32_t);
            while (lg > 1 && lg * 4u * lean_per_wave > budget) lg = (lg + 1) / 2;
            L.lean_grid = (unsigned)lg;
            lean_bytes = (size_t)lg * 4u * lean_per_wave;
        }
    }
    const size_t need = std::max((size_t)grid * 4u * per_wave, lean_bytes);
    queue_at = (need + 255u) & ~(size_t)255u;
    int rc = ctx_reserve((void**)d_spill, spill_bytes, queue_at + (L.lean_grid ? (size_t)n * sizeof(uint32_t) : 0));
    if (rc != MFA_OK) return rc;
    a.spill = *d_spill;
    a.lean_queue = L.lean_grid ? reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(*d_spill) + queue_at) : nullptr;
    const bool stats = getenv("MFA_WALK_STATS") != nullptr && p.K == 1;
    if (stats) { a.lean_queue = nullptr; L.lean_grid = 0; }
    HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long) * (stats ? 32 : 3), (hipStream_t)stream));      // ticket counter, queue length, the lean kernel's tickets
    // (what the launch waits for -- its group's regions -- comes AFTER its own preparations on the stream: they are done when the event arrives)
    if (wait_event) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_event, 0));
    if (stats) {
        rc = launch_walk_stats(L, stream);
        if (rc == MFA_OK) { (void)hipStreamSynchronize((hipStream_t)stream); walk_print_stats(d_counter, "walk"); }
        return rc;
    }
    if (long_lists) return launch_walk_long_k1(L, stream);
    switch (p.K) {
        case 1: return launch_walk_k1(L, stream); case 2: return launch_walk_k2(L, stream); case 3: return launch_walk_k3(L, stream);
        case 4: return launch_walk_k4(L, stream); case 5: return launch_walk_k5(L, stream); case 6: return launch_walk_k6(L, stream);
        case 7: return launch_walk_k7(L, stream); case 8: return launch_walk_k8(L, stream); case 9: return launch_walk_k9(L, stream);
    }
    return MFA_ERR_UNSUPPORTED;
}

void lean_hint_free(LeanHint& h) {
    if (h.h_seen) (void)hipHostFree(h.h_seen);
    h.h_seen = nullptr;
}

void walk_print_stats(unsigned long long* d_counter, const char* tag) {
    unsigned long long h[32] = {0};
    if (hipMemcpy(h, d_counter, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    const unsigned long long* o = h + 8;
    const double tot = (double)o[16] > 0 ? (double)o[16] : 1.0;
    fprintf(stderr, "%s stats: %llu waves, %llu strings, wave-iterations %llu (dual %llu), lane steps %llu, skipped %llu, probes %llu (hits %llu), spill steps %llu\n", tag,
            o[17], o[5], o[8], o[9], o[3], o[0], o[1], o[2], o[4]);
    fprintf(stderr, "%s stats: cycles per wave-iteration %.0f; share: string start %.1f%%, byte %.1f%%, region look-up %.1f%%, plain step %.1f%%, dual step %.1f%%, book-keeping %.1f%%\n", tag,
            tot / (double)(o[8] ? o[8] : 1), 100.0 * o[10] / tot, 100.0 * o[11] / tot, 100.0 * o[12] / tot, 100.0 * o[13] / tot, 100.0 * o[14] / tot, 100.0 * o[15] / tot);
}

}  // namespace mfa

using namespace mfa;

// ---- mixed batches ---------------------------------------------------------------------------------------------------------------------
// One batch, several automata.  The region pass runs over groups of consecutive segments on one internal stream; what walks a group
// starts as soon as the group's regions are known and runs beside the next group's region pass:
//   * table engine (MFA_WALK=table): ONE launch of the table-driven walk kernel per group, any lane any automaton;
//   * generated kernels (default while they are the faster walk on small automata): one launch per segment, spread over a few walk
//     streams by measured cost -- the first call on a device runs the walks one after the other and times them, later calls give
//     each walk to the stream that can start it first (list scheduling with the groups' region times as release times).
constexpr uint32_t MIX_MAX_GROUPS = 12, MIX_MAX_STREAMS = 4, MIX_MAX_LAUNCHES = 24, MIX_TIMINGS = 32;

struct mfa_mixed {
    std::vector<mfa_image*> images;
    std::vector<uint32_t>   words;          // the images' table blocks, back to back (table engine)
    std::vector<uint32_t>   block_at;       // word offset of image k's block
    uint32_t K = 1, max_live = 1;
    bool reversed = false, table_ok = true;
    std::mutex mu;
    std::map<uint64_t, uint64_t> bytes_of;                     // string count of a batch -> its bytes (read back once, see mfa_match_mixed)
    struct Dev {
        uint32_t* d_tables = nullptr;
        hipStream_t rs = nullptr;                              // region stream (the gate; MFA_MIXED_REGION_ON_CALLER=0)
        hipStream_t last_cs = nullptr;                         // the caller's stream of the last call
        hipStream_t ws[MIX_MAX_STREAMS] = {nullptr};           // walk streams
        hipEvent_t ev_g[MIX_MAX_GROUPS] = {nullptr};           // group g's regions are known (timed)
        hipEvent_t ev_w[MIX_MAX_STREAMS] = {nullptr};          // end of a walk stream's work
        hipEvent_t ev_in = nullptr;
        // timing of the last MIX_TIMINGS calls (a ring): first region launch, end of the last region launch, end of the call
        hipEvent_t ev_r0[MIX_TIMINGS] = {nullptr}, ev_r1[MIX_TIMINGS] = {nullptr}, ev_end[MIX_TIMINGS] = {nullptr};
        uint64_t calls = 0;
        uint64_t* d_regions = nullptr; size_t region_bytes = 0;
        uint32_t* d_spill[MIX_MAX_LAUNCHES] = {nullptr}; size_t spill_bytes[MIX_MAX_LAUNCHES] = {0};
        LeanHint lean[MIX_MAX_LAUNCHES];
        unsigned long long* d_counters = nullptr;
        int n_cus = 0;
        // the gate: ONE region launch per call; in front of a group's walk launches, on their stream, one wave that ends when the region
        // kernel has counted every string of the group (regions.hip: GATE)
        int gate_state = 0;                                    // 0 not tried, 1 usable, -1 not
        uint64_t* h_hdr = nullptr;                             // pinned: the gate headers of the last MIX_TIMINGS calls
        hipEvent_t ev_clear = nullptr;
        uint32_t last_region_launches = 0, last_walk_launches = 0, last_groups = 0;
        bool last_gated = false;
        bool timed = false, calibrated = false;
        std::vector<float> cost;                               // per segment: its walk alone, ms
        float ready[MIX_MAX_GROUPS] = {0};                     // per group: end of its region launch, ms from the start of the call
        uint32_t ng_last = 0;
    };
    std::map<int, Dev> dev;
};

extern "C" {

int mfa_mixed_create(mfa_image_t* const* images, uint32_t n_images, mfa_mixed_t** out) {
    if (!images || !out || n_images == 0) return MFA_ERR_INVALID_ARG;
    *out = nullptr;
    mfa_mixed* mx = new (std::nothrow) mfa_mixed();
    if (!mx) return MFA_ERR_NOMEM;
    for (uint32_t k = 0; k < n_images; k++) {
        mfa_image* img = images[k];
        if (!img) { delete mx; return MFA_ERR_INVALID_ARG; }
        if (img->host.h.kind != MFA_KIND_MFA) { delete mx; return MFA_ERR_UNSUPPORTED; }      // memory automata (tabulated ones have no regions to share)
        if (!img->walk_ok) mx->table_ok = false;
        if (k == 0) mx->reversed = img->walk.reversed;
        else if (mx->reversed != img->walk.reversed) mx->table_ok = false;                      // one scan direction per table launch
        mx->images.push_back(img);
        mx->K = std::max(mx->K, img->walk.K);
        mx->max_live = std::max(mx->max_live, img->walk.max_live);
    }
    if (mx->table_ok)
        for (mfa_image* img : mx->images) {
            mx->block_at.push_back((uint32_t)mx->words.size());
            if (mx->K > 6 && img->walk.K <= 6) {                  // a kernel for more than 6 cells reads 3-word edges
                WalkTables wide;
                if (build_walk_tables(img->host, wide, true) != MFA_OK) { mx->table_ok = false; break; }
                mx->words.insert(mx->words.end(), wide.words.begin(), wide.words.end());
            } else mx->words.insert(mx->words.end(), img->walk.words.begin(), img->walk.words.end());
        }
    *out = mx;
    return MFA_OK;
}

void mfa_mixed_destroy(mfa_mixed_t* mx) {
    if (!mx) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto& kv : mx->dev) {
        (void)hipSetDevice(kv.first);
        mfa_mixed::Dev& d = kv.second;
        if (d.rs) (void)hipStreamSynchronize(d.rs);
        for (hipStream_t w : d.ws) if (w) (void)hipStreamSynchronize(w);
        if (d.calls > 0) (void)hipEventSynchronize(d.ev_end[(d.calls - 1) % MIX_TIMINGS]);      // (region launches on a caller's stream)
        if (d.d_tables) (void)hipFree(d.d_tables);
        if (d.d_regions) (void)hipFree(d.d_regions);
        for (uint32_t* p : d.d_spill) if (p) (void)hipFree(p);
        for (LeanHint& h : d.lean) lean_hint_free(h);
        if (d.d_counters) (void)hipFree(d.d_counters);
        if (d.h_hdr) (void)hipHostFree(d.h_hdr);
        if (d.ev_clear) (void)hipEventDestroy(d.ev_clear);
        for (hipEvent_t e : d.ev_g) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : d.ev_w) if (e) (void)hipEventDestroy(e);
        if (d.ev_in) (void)hipEventDestroy(d.ev_in);
        for (uint32_t k = 0; k < MIX_TIMINGS; k++) for (hipEvent_t e : {d.ev_r0[k], d.ev_r1[k], d.ev_end[k]}) if (e) (void)hipEventDestroy(e);
        if (d.rs) (void)hipStreamDestroy(d.rs);
        for (hipStream_t w : d.ws) if (w) (void)hipStreamDestroy(w);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    delete mx;
}

static int mixed_device(mfa_mixed* mx, int device, mfa_mixed::Dev** out) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MFA_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    auto it = mx->dev.find(device);
    if (it != mx->dev.end()) { *out = &it->second; return MFA_OK; }
    mfa_mixed::Dev d;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d.n_cus = prop.multiProcessorCount;
    if (mx->table_ok) {
        HIP_TRY(hipMalloc((void**)&d.d_tables, mx->words.size() * 4));
        HIP_TRY(hipMemcpy(d.d_tables, mx->words.data(), mx->words.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&d.d_counters, 64 * MIX_MAX_LAUNCHES * sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreateWithFlags(&d.rs, hipStreamNonBlocking));
    // Streams are made when a call first needs them (walk_stream / region_stream2): every stream beyond the hardware queues of the process (four
    // by default, the caller's included) shares a queue with another one, and work on streams that share a queue is serialised
    for (hipEvent_t& e : d.ev_g) HIP_TRY(hipEventCreate(&e));
    for (hipEvent_t& e : d.ev_w) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.ev_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.ev_clear, hipEventDisableTiming));
    for (uint32_t k = 0; k < MIX_TIMINGS; k++) { HIP_TRY(hipEventCreate(&d.ev_r0[k])); HIP_TRY(hipEventCreate(&d.ev_r1[k])); HIP_TRY(hipEventCreate(&d.ev_end[k])); }
    d.cost.assign(mx->images.size(), 0.0f);
    auto ins = mx->dev.emplace(device, d);
    *out = &ins.first->second;
    return MFA_OK;
}

// Is the gate usable on this device?  Decided once.
static bool gate_ready(mfa_mixed::Dev& d) {
    if (d.gate_state != 0) return d.gate_state > 0;
    d.gate_state = -1;
    if (hipHostMalloc((void**)&d.h_hdr, MIX_TIMINGS * MFA_GATE_FIXED_WORDS * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { d.h_hdr = nullptr; (void)hipGetLastError(); return false; }
    d.gate_state = 1;
    return true;
}

// What a call will launch, decided before anything is put on a stream (an error found here leaves the streams untouched).
struct MixLaunch { uint32_t g, s0, s1, ml, Kc, w0, w1; uint64_t a, b; int k; };

// seg_first: HOST array of n_images + 1 string indices, seg_first[0] = 0, seg_first[n_images] = n: strings seg_first[s] ..
// seg_first[s+1]-1 are matched against images[s] (the order of mfa_mixed_create).  `stream` sees the call as one operation.
// total_bytes: offsets[n] - offsets[0] if the caller knows it, else 0 (then it is read back once per string count: see the header).
static int match_mixed_impl(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const uint64_t* seg_first,
                            uint8_t* d_results, int device, void* stream, uint64_t total_bytes) {
    if (!mx || !d_offsets || !seg_first || (!d_results && n)) return MFA_ERR_INVALID_ARG;
    const uint32_t ns = (uint32_t)mx->images.size();
    if (seg_first[0] != 0 || seg_first[ns] != n) return MFA_ERR_INVALID_ARG;
    for (uint32_t s = 0; s < ns; s++)
        if (seg_first[s] > seg_first[s + 1]) return MFA_ERR_INVALID_ARG;
    if (n == 0) return MFA_OK;
    std::lock_guard<std::mutex> lk(mx->mu);
    mfa_mixed::Dev* d = nullptr;
    int rc = mixed_device(mx, device, &d);
    if (rc != MFA_OK) return rc;
    hipStream_t cs = (hipStream_t)stream;
    const bool table = walk_mode() != 2 && mx->table_ok;      // the table engine unless the generated kernels are asked for
    // groups: ranges of strings, cut at fractions of the batch (a segment may straddle a cut).  Decreasing sizes: the walk of the
    // last group is what the call ends with.
    std::vector<uint64_t> cut{0};
    {
        // How many groups pays depends on the batch's BYTES: a group's region launch should take about as long as a walk launch needs anyway
        // (a walk is latency-bound: ~0.3-0.5 ms for 20 000 strings as for 200 000): 1.3 GB per group, eight groups at most (measured: 10.7 GB of
        // 64 KiB strings 2.02 ms in eight groups, 2.28 ms in four; the 19.4 GB headline batch eight).  Cut finer, a small batch pays the walks'
        // latency once per group (a 1.9 GB batch of one automaton: 1.26 ms in eight groups against 0.52 ms in one; 34 ms against 9.8 ms
        // for the 77-node automaton).  The bytes of a batch are device data (offsets[n] - offsets[0]): a caller that knows them says so
        // (mfa_match_mixed_sized); otherwise they are read back ONCE per string count this object meets -- that call waits for the
        // caller's stream -- and remembered (a later batch with the same count and other bytes gets the same grouping: a matter of speed only).
        const char* spec = getenv("MFA_MIXED_CUTS");
        std::string made;
        if (!spec) {
            uint64_t bytes = total_bytes;
            auto known = mx->bytes_of.find(n);
            if (bytes != 0) { /* the caller's word */ }
            else if (known != mx->bytes_of.end()) bytes = known->second;
            else if (n >= 65536) {
                uint64_t ends[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(&ends[0], d_offsets, sizeof(uint64_t), hipMemcpyDeviceToHost, cs));
                HIP_TRY(hipMemcpyAsync(&ends[1], d_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, cs));
                HIP_TRY(hipStreamSynchronize(cs));
                bytes = ends[1] - ends[0];
                if (mx->bytes_of.size() >= 64) mx->bytes_of.clear();
                mx->bytes_of[n] = bytes;
            }
            const double per_group = table ? 1.3e9 : 2.0e9;
            const uint32_t most = table ? 8u : 5u;
            uint32_t want = (uint32_t)std::min<double>(most, std::max(1.0, std::floor((double)bytes / per_group + 0.5)));
            if (n < 65536) want = 1;
            // sizes: equal, the last three groups 0.8 / 0.53 / 0.33 of that (two groups: 1, 0.6; three: 1, 0.8, 0.4) -- the walk of the last
            // group is what the call ends with
            std::vector<double> w(want, 1.0);
            if (want == 2) w[1] = 0.6;
            else if (want == 3) { w[1] = 0.8; w[2] = 0.4; }
            else if (want >= 4) { w[want - 3] = 0.8; w[want - 2] = 0.53; w[want - 1] = 0.33; }
            double total = 0, acc = 0;
            for (double x : w) total += x;
            for (uint32_t k = 0; k + 1 < want; k++) { acc += w[k]; made += (k ? "," : "") + std::to_string(acc / total); }
            spec = made.c_str();
        }
        for (const char* q = spec; *q && cut.size() < MIX_MAX_GROUPS;) {
            const uint64_t at = (uint64_t)((double)n * atof(q));
            if (at > cut.back() && at < n) cut.push_back(at);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        cut.push_back(n);
    }
    const uint32_t ng = (uint32_t)cut.size() - 1;
    const char* re = getenv("MFA_REGIONS");
    const char* ae = getenv("MFA_ACCEL");
    const bool with_regions = !(re && re[0] == '0') && !(ae && ae[0] == '0');
    int NW = env_int("MFA_MIXED_WALK_STREAMS", table ? 2 : 3);
    if (NW < 1) NW = 1;
    if (NW > (int)MIX_MAX_STREAMS) NW = MIX_MAX_STREAMS;
    // ONE region launch for the whole batch and the walks released group by group (the gate): table engine, more than one group.
    // Built, checked (tests/test_regions_gpu.py::test_gated_walks_see_fresh_tables) and measured in round 4 -- and NOT the default: it saves
    // seven region launches, and the step takes as long or longer (4.29 against 3.84-4.18 ms on one box; kernel traces in
    // profiles/r04d_trace_*.txt).  What a launch's ramp and tail leave idle, the walk kernels beside it use: with one launch the walks take
    // 5.3 ms of stream time instead of 4.2, the region pass 3.83 ms instead of 3.9 for its eight launches.  MFA_MIXED_GATE=1 turns it on.
    const bool gate = table && with_regions && ng > 1 && ng <= MFA_GATE_MAX_GROUPS && env_int("MFA_MIXED_GATE", 0) != 0 && gate_ready(*d);
    const uint32_t stamp = (uint32_t)(d->calls & 0xfffu);     // what every word of this call's table rows carries (the table buffer is rewritten by every call)

    // One automaton, one group: exactly the single-automaton call (mfa_match_batch: region pass, then the walk, on the caller's stream, with the
    // engine that call would choose) -- the hops to the internal streams and back cost such a batch 0.03-0.06 ms and buy it nothing.  (Cutting a
    // 1.9 GB batch of ONE automaton into two or three groups was measured in round 4, configs[4]: 0.517 ms in one piece, 0.61 / 0.69 / 0.74
    // ms in two / three / four groups: a walk launch is latency-bound, its 0.15 ms are paid per group and hide behind nothing that short.)
    if (ns == 1 && ng == 1 && env_int("MFA_MIXED_SINGLE_DIRECT", 1) != 0) {
        const uint32_t slot1 = (uint32_t)(d->calls % MIX_TIMINGS);
        if (d->calls > 0 && d->last_cs != cs) HIP_TRY(hipStreamWaitEvent(cs, d->ev_end[(d->calls - 1) % MIX_TIMINGS], 0));
        d->last_cs = cs;
        HIP_TRY(hipEventRecord(d->ev_r0[slot1], cs));
        rc = mfa_match_batch(mx->images[0], d_bytes, d_offsets, n, d_results, device, stream);
        HIP_TRY(hipEventRecord(d->ev_r1[slot1], cs));
        HIP_TRY(hipEventRecord(d->ev_end[slot1], cs));
        if (rc != MFA_OK) return rc;
        d->timed = true; d->calls++; d->ng_last = 1;
        d->last_region_launches = with_regions ? 1u : 0u; d->last_walk_launches = 1; d->last_groups = 1; d->last_gated = false;
        return MFA_OK;
    }

    // ---- the plan (table engine): one launch per group and run of consecutive segments whose automata have the same number of cells (a
    // launch's kernel and its LDS footprint are those of its largest cell count); groups alternate between the walk streams, so that a
    // group's walk may start while the one before it drains
    std::vector<MixLaunch> plan;
    if (table)
        for (uint32_t g = 0; g < ng; g++) {
            const uint64_t lo = cut[g], hi = cut[g + 1];
            uint32_t sa = 0;
            while (sa + 1 < ns && seg_first[sa + 1] <= lo) sa++;
            uint32_t sb = sa;
            while (sb < ns && seg_first[sb] < hi) sb++;
            for (uint32_t s0 = sa; s0 < sb;) {
                uint32_t s1 = s0 + 1;
                const uint32_t Kc = mx->K > 6 ? mx->K : mx->images[s0]->walk.K;
                while (s1 < sb && s1 - s0 < WALK_MAX_SEG && (mx->K > 6 || mx->images[s1]->walk.K == Kc)) s1++;
                const uint64_t a = std::max(seg_first[s0], lo), b = std::min(seg_first[s1], hi);
                if (b > a) {
                    uint32_t ml = 1;
                    for (uint32_t j = s0; j < s1; j++) ml = std::max(ml, mx->images[j]->walk.max_live);
                    // the launch gets the blocks of ITS automata only (they lie back to back): less LDS per workgroup
                    plan.push_back(MixLaunch{g, s0, s1, ml, Kc, mx->block_at[s0], s1 < ns ? mx->block_at[s1] : (uint32_t)mx->words.size(), a, b, (int)(g % (uint32_t)NW)});
                }
                s0 = s1;
            }
        }
    if (plan.size() > MIX_MAX_LAUNCHES) return MFA_ERR_UNSUPPORTED;      // (more runs of equal cell count than the object has launch slots: nothing was started)
    uint64_t* d_table = nullptr;
    if (with_regions) {
        const size_t head = gate ? MFA_GATE_HEADER_WORDS : 0;
        rc = ctx_reserve((void**)&d->d_regions, &d->region_bytes, ((size_t)n * MFA_REGION_WORDS + MFA_GATE_HEADER_WORDS) * sizeof(uint64_t));
        if (rc != MFA_OK) return rc;
        d_table = d->d_regions + head;
    }
    // which stream walks which segment (generated kernels): first call one after the other (timed), then by cost
    std::vector<int> where(ns, 0);
    const bool calibrating = !table && !d->calibrated;
    if (!table && d->calibrated && d->ng_last == ng) {
        // a segment's walk is released when the group that holds its first string is scanned (segments that straddle a cut are rare)
        float free_at[MIX_MAX_STREAMS] = {0};
        for (uint32_t s = 0, g = 0; s < ns; s++) {
            while (g + 1 < ng && cut[g + 1] <= seg_first[s]) g++;
            int best = 0;
            for (int k = 1; k < NW; k++)
                if (std::max(free_at[k], d->ready[g]) < std::max(free_at[best], d->ready[g])) best = k;
            free_at[best] = std::max(free_at[best], d->ready[g]) + 1.5f * d->cost[s];      // beside the region pass a walk takes about 1.5 x its time alone
            where[s] = best;
        }
    }
    for (int k = 0; k < NW; k++)
        if (!d->ws[k]) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (env_int("MFA_MIXED_WALK_PRIORITY", 0) != 0) HIP_TRY(hipStreamCreateWithPriority(&d->ws[k], hipStreamNonBlocking, greatest));
            else HIP_TRY(hipStreamCreateWithFlags(&d->ws[k], hipStreamNonBlocking));
        }
    const uint32_t slot_t = (uint32_t)(d->calls % MIX_TIMINGS);
    // The region launches go to the CALLER's stream (MFA_MIXED_REGION_ON_CALLER=0, or the gate: to a stream of the object's own): back-to-back calls
    // then pass from the last walk of one to the first region launch of the next through ONE event (walk stream -> caller's stream) instead of
    // three (walk stream -> region stream -> caller's stream -> region stream): 0.02 ms a call.
    hipStream_t const rs = (!gate && env_int("MFA_MIXED_REGION_ON_CALLER", 1) != 0) ? cs : d->rs;

    // ---- from here on work goes to the internal streams.  Whatever happens, the caller's stream is made to wait for all of it before this
    // function returns: a caller that gets an error may free or reuse its buffers in stream order like one that gets MFA_OK.
    struct Join {
        mfa_mixed::Dev* d; hipStream_t cs, rs; int NW; uint32_t slot_t; bool used[MIX_MAX_STREAMS] = {false}; bool started = false; int err = MFA_OK;
        void run() {
            if (!started) return;
            started = false;
            for (int k = 0; k < NW; k++)
                if (used[k]) {
                    if (hipEventRecord(d->ev_w[k], d->ws[k]) != hipSuccess || hipStreamWaitEvent(rs, d->ev_w[k], 0) != hipSuccess) { err = MFA_ERR_HIP; (void)hipStreamSynchronize(d->ws[k]); }
                }
            if (hipEventRecord(d->ev_end[slot_t], rs) != hipSuccess || (rs != cs && hipStreamWaitEvent(cs, d->ev_end[slot_t], 0) != hipSuccess)) { err = MFA_ERR_HIP; (void)hipStreamSynchronize(rs); }
        }
        ~Join() { run(); }
    } join{d, cs, rs, NW, slot_t};
    // (the object's buffers -- table, counters, spill areas -- are shared by its calls: with the region launches on an internal stream the calls
    // follow each other there; on callers' streams a call starts behind the end of the one before it, whichever stream that one came on)
    if (d->calls > 0 && d->last_cs != cs) HIP_TRY(hipStreamWaitEvent(cs, d->ev_end[(d->calls - 1) % MIX_TIMINGS], 0));      // (the walk streams start behind ev_in: below)
    d->last_cs = cs;
    HIP_TRY(hipEventRecord(d->ev_in, cs));
    if (rs != cs) HIP_TRY(hipStreamWaitEvent(rs, d->ev_in, 0));
    join.started = true;
    if (gate) {
        // the gate header in front of the table (regions.hip: gate_signal), the counters back to zero; the walk streams start behind both
        uint64_t* h = d->h_hdr + (size_t)slot_t * MFA_GATE_FIXED_WORDS;
        for (uint32_t k = 0; k < MFA_GATE_FIXED_WORDS; k++) h[k] = 0ull;
        for (uint32_t k = 0; k < MFA_GATE_MAX_GROUPS; k++) h[MFA_GATE_FIXED_WORDS - 1 - k] = k < ng ? cut[k + 1] : ~0ull;
        h[MFA_GATE_FIXED_WORDS - 33] = (uint64_t)stamp << 52;
        HIP_TRY(hipMemsetAsync(d->d_regions, 0, (MFA_GATE_HEADER_WORDS - MFA_GATE_FIXED_WORDS) * sizeof(uint64_t), rs));      // the counters
        HIP_TRY(hipMemcpyAsync(d->d_regions + (MFA_GATE_HEADER_WORDS - MFA_GATE_FIXED_WORDS), h, MFA_GATE_FIXED_WORDS * sizeof(uint64_t), hipMemcpyHostToDevice, rs));
        HIP_TRY(hipEventRecord(d->ev_clear, rs));
    }
    for (int k = 0; k < NW; k++) HIP_TRY(hipStreamWaitEvent(d->ws[k], gate ? d->ev_clear : d->ev_in, 0));
    HIP_TRY(hipEventRecord(d->ev_r0[slot_t], rs));
    uint32_t region_launches = 0;
    const bool ext_events = table && !gate && env_int("MFA_MIXED_EXT_EVENTS", 1) != 0;      // a group's event = its region launch's completion signal
    if (gate) {
        rc = launch_region_scan(d->n_cus, d_bytes, d_offsets, n, d_table, rs, 128u, true);
        if (rc != MFA_OK) return rc;
        region_launches = 1;
    }
    uint32_t slot = 0;
    for (uint32_t g = 0; g < ng; g++) {
        const uint64_t lo = cut[g], hi = cut[g + 1];
        if (with_regions && !gate) {
            rc = launch_region_scan(d->n_cus, d_bytes, d_offsets + lo, hi - lo, d_table + lo * MFA_REGION_WORDS, rs, table ? 128u : 256u, false, ext_events ? d->ev_g[g] : nullptr);
            if (rc != MFA_OK) return rc;
            region_launches++;
        }
        if (!gate && !(with_regions && ext_events)) HIP_TRY(hipEventRecord(d->ev_g[g], rs));
        bool waits[MIX_MAX_STREAMS] = {false};
        // a stream's first launch of this group waits for the group's regions: for the event behind its region launch, or -- with the gate -- for
        // one wave, launched in front of it, that ends when the region kernel has counted every string of the group
        auto release = [&](int k) -> int {
            if (waits[k]) return MFA_OK;
            waits[k] = true;
            if (gate) { join.used[k] = true; return launch_gate_wait(d_table, g, d->ws[k]); }
            HIP_TRY(hipStreamWaitEvent(d->ws[k], d->ev_g[g], 0));
            return MFA_OK;
        };
        if (table) {
            for (const MixLaunch& L : plan) {
                if (L.g != g) continue;
                void* wait_for = nullptr;                       // (without the gate the launch itself waits, behind its own preparations on the stream)
                if (!gate && !waits[L.k]) { waits[L.k] = true; wait_for = d->ev_g[g]; }
                else {
                    rc = release(L.k);
                    if (rc != MFA_OK) return rc;
                }
                uint32_t sf[WALK_MAX_SEG + 1], stb[WALK_MAX_SEG];
                for (uint32_t j = 0; j <= L.s1 - L.s0; j++) sf[j] = (uint32_t)(std::min(std::max(seg_first[L.s0 + j], L.a), L.b) - L.a);
                for (uint32_t j = 0; j < L.s1 - L.s0; j++) stb[j] = mx->block_at[L.s0 + j] - L.w0;
                const WalkPlanInput pk{L.Kc, L.ml, mx->reversed, L.w1 - L.w0};
                join.used[L.k] = true;
                rc = launch_walk(pk, d->d_tables + L.w0, d->n_cus, d_bytes, d_offsets + L.a, L.b - L.a, d_results + L.a, d_table ? d_table + L.a * MFA_REGION_WORDS : nullptr,
                                 L.s1 - L.s0, sf, stb, &d->d_spill[slot], &d->spill_bytes[slot], d->d_counters + 64 * slot, d->ws[L.k], gate ? (0x1000u | stamp) : 0u, &d->lean[slot], wait_for);
                if (rc != MFA_OK) return rc;
                slot++;
            }
        } else {
            uint32_t sa = 0;
            while (sa + 1 < ns && seg_first[sa + 1] <= lo) sa++;
            uint32_t sb = sa;
            while (sb < ns && seg_first[sb] < hi) sb++;
            for (uint32_t s = sa; s < sb; s++) {
                const uint64_t a = std::max(seg_first[s], lo), b = std::min(seg_first[s + 1], hi);
                if (b <= a) continue;
                const int k = where[s];
                rc = release(k);
                if (rc != MFA_OK) return rc;
                join.used[k] = true;
                rc = mfa_match_batch_regions(mx->images[s], d_bytes, d_offsets + a, b - a, d_results + a, d_table ? d_table + a * MFA_REGION_WORDS : nullptr, device, d->ws[k]);
                if (rc != MFA_OK) return rc;
                slot++;
            }
        }
    }
    HIP_TRY(hipEventRecord(d->ev_r1[slot_t], rs));
    // the caller's stream (and the call's end event, on the region stream) wait for every stream that was given work
    join.run();
    if (join.err != MFA_OK) return join.err;
    d->timed = true;
    d->calls++;
    d->ng_last = ng;
    d->last_region_launches = region_launches; d->last_walk_launches = slot; d->last_groups = ng; d->last_gated = gate;
    if (calibrating) {                                        // once per device: the walks' costs and the groups' region times
        HIP_TRY(hipEventSynchronize(d->ev_end[slot_t]));
        for (uint32_t s = 0; s < ns; s++) {
            float ms = 0.0f;
            if (seg_first[s + 1] > seg_first[s] && mfa_last_kernel_ms(mx->images[s], device, &ms) == MFA_OK) d->cost[s] = ms;
        }
        for (uint32_t g = 0; g < ng; g++) {
            // the calibration pass runs a group's walks before the next group's region launch is reached by nothing: region launches
            // follow each other on their own stream, so the elapsed time between two group events is the later group's region time
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev_r0[slot_t], d->ev_g[g]));
            d->ready[g] = ms;
        }
        d->calibrated = true;
    }
    return MFA_OK;
}

int mfa_match_mixed(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const uint64_t* seg_first,
                    uint8_t* d_results, int device, void* stream) {
    return match_mixed_impl(mx, d_bytes, d_offsets, n, seg_first, d_results, device, stream, 0);
}

int mfa_match_mixed_sized(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, const uint64_t* seg_first,
                          uint8_t* d_results, int device, void* stream) {
    return match_mixed_impl(mx, d_bytes, d_offsets, n, seg_first, d_results, device, stream, total_bytes);
}

// what the last call on `device` launched (any pointer may be NULL): region launches (1 with the gate), walk launches, groups of strings, and
// whether the walks were released by counters (1) or by events behind per-group region launches (0)
int mfa_mixed_last_launches(mfa_mixed_t* mx, int device, uint32_t* region_launches, uint32_t* walk_launches, uint32_t* groups, uint32_t* gated) {
    if (!mx) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(mx->mu);
    auto it = mx->dev.find(device);
    if (it == mx->dev.end() || !it->second.timed) return MFA_ERR_INVALID_ARG;
    if (region_launches) *region_launches = it->second.last_region_launches;
    if (walk_launches) *walk_launches = it->second.last_walk_launches;
    if (groups) *groups = it->second.last_groups;
    if (gated) *gated = it->second.last_gated ? 1u : 0u;
    return MFA_OK;
}

// The same with HOST pointers: copies the batch to the device, matches, copies the results back, synchronises (the host mirror's
// match_mixed and the `diploma -match-mixed` command line; throughput is then bounded by the host link).
int mfa_match_mixed_host(mfa_mixed_t* mx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n, const uint64_t* seg_first, uint8_t* results, int device) {
    if (!mx || !offsets || !seg_first || (!results && n)) return MFA_ERR_INVALID_ARG;
    if (n == 0) return MFA_OK;
    for (uint64_t k = 0; k < n; k++) {
        if (offsets[k + 1] < offsets[k]) return MFA_ERR_INVALID_ARG;
        if (offsets[k + 1] - offsets[k] > MFA_MAX_STRING_BYTES) return MFA_ERR_TOO_LONG;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MFA_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    const uint64_t total = offsets[n] - offsets[0];
    uint8_t* d_bytes = nullptr; uint64_t* d_off = nullptr; uint8_t* d_res = nullptr;
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t k = 0; k <= n; k++) rel[k] = offsets[k] - offsets[0];
    int rc = MFA_OK;
    hipError_t e = hipMalloc((void**)&d_bytes, total + 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, (n + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_res, n);
    if (e == hipSuccess && total) e = hipMemcpy(d_bytes, bytes + offsets[0], total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, rel.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_last_hip_error((int)e); rc = MFA_ERR_HIP; }
    if (rc == MFA_OK) rc = match_mixed_impl(mx, d_bytes, d_off, n, seg_first, d_res, device, nullptr, total);
    if (rc == MFA_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(results, d_res, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_last_hip_error((int)e); rc = MFA_ERR_HIP; }
    }
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_off) (void)hipFree(d_off);
    if (d_res) (void)hipFree(d_res);
    return rc;
}

// Device time of a recent mfa_match_mixed on `device`: from its first region launch to the end of its last region launch, and to the
// end of its last walk (either pointer may be NULL).  back = 0: the last call, 1: the one before, ... (the library keeps the events
// of its last 32 calls, so a caller can time a sequence of calls without synchronising between them).  Synchronises on that call's end.
int mfa_mixed_timing(mfa_mixed_t* mx, int device, uint32_t back, float* region_ms, float* span_ms) {
    if (!mx) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(mx->mu);
    auto it = mx->dev.find(device);
    if (it == mx->dev.end() || !it->second.timed || back >= MIX_TIMINGS || back >= it->second.calls) return MFA_ERR_INVALID_ARG;
    mfa_mixed::Dev& d = it->second;
    const uint32_t k = (uint32_t)((d.calls - 1 - back) % MIX_TIMINGS);
    HIP_TRY(hipEventSynchronize(d.ev_end[k]));
    if (region_ms) HIP_TRY(hipEventElapsedTime(region_ms, d.ev_r0[k], d.ev_r1[k]));
    if (span_ms) HIP_TRY(hipEventElapsedTime(span_ms, d.ev_r0[k], d.ev_end[k]));
    return MFA_OK;
}

int mfa_mixed_last_ms(mfa_mixed_t* mx, int device, float* region_ms, float* span_ms) { return mfa_mixed_timing(mx, device, 0u, region_ms, span_ms); }

}  // extern "C"
