// Host side of the live-list walk: launch planning, and the mixed-batch object of the C-ABI (mfa_mixed_*).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

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

bool walk_selected() {
    const char* e = getenv("MFA_WALK");
    return e && e[0] == 't';                                  // "table"; anything else: the generated kernels
}

static size_t wave_words(uint32_t K, uint32_t C) {
    switch (K) {
        case 1: return walk_wave_words_k1(C); case 2: return walk_wave_words_k2(C); case 3: return walk_wave_words_k3(C);
        case 4: return walk_wave_words_k4(C); case 5: return walk_wave_words_k5(C); case 6: return walk_wave_words_k6(C);
        case 7: return walk_wave_words_k7(C); case 8: return walk_wave_words_k8(C); default: return walk_wave_words_k9(C);
    }
}

// One launch over the sub-batch [d_offsets[0], d_offsets[n]).  The LDS capacity C of the lists: enough for the automata's
// longest possible list if that leaves room for two workgroups per CU, else what does (longer lists spill to `d_spill`).
int launch_walk(const WalkPlanInput& p, const uint32_t* d_tables, int n_cus, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                uint8_t* d_results, const uint64_t* d_regions, uint32_t n_seg, const uint32_t* seg_first, const uint32_t* seg_table,
                uint32_t** d_spill, size_t* spill_bytes, unsigned long long* d_counter, void* stream) {
    if (n == 0) return MFA_OK;
    if (n_seg == 0 || n_seg > WALK_MAX_SEG || n > 0xffffffffull) return MFA_ERR_INVALID_ARG;
    WalkLaunch L;
    WalkArgs& a = L.args;
    std::memset(&a, 0, sizeof a);
    a.bytes = d_bytes; a.offsets = d_offsets; a.n = n; a.results = d_results; a.regions = d_regions; a.tables = d_tables;
    a.counter = d_counter;
    a.table_words = p.table_words;
    a.shared_words = (p.table_words + 63u) & ~63u;
    a.n_seg = n_seg;
    for (uint32_t k = 0; k <= n_seg; k++) a.seg_first[k] = seg_first[k];
    for (uint32_t k = 0; k < n_seg; k++) a.seg_table[k] = seg_table[k];
    const char* ae = getenv("MFA_ACCEL");
    a.accel = (ae && ae[0] == '0') ? 0u : 1u;
    // capacity
    const size_t lds_max = 160u * 1024u / 4u;                 // words
    const int want_c = env_int("MFA_WALK_C", 0);
    const uint32_t c_cap = want_c > 0 ? (uint32_t)want_c : 8u;
    uint32_t C = std::min(p.max_live, c_cap);
    if (C < 1) C = 1;
    const uint32_t wgs_goal = (uint32_t)env_int("MFA_WALK_WGS", 2);
    while (C > 1 && want_c <= 0 && a.shared_words + 4u * wave_words(p.K, C) > lds_max / wgs_goal) C--;
    while (C > 1 && a.shared_words + 4u * wave_words(p.K, C) > lds_max) C--;
    if (a.shared_words + 4u * wave_words(p.K, C) > lds_max) return MFA_ERR_UNSUPPORTED;      // the tables alone fill the LDS
    a.C = C;
    a.CX = p.max_live > C ? p.max_live - C : 1u;
    const size_t lds_words = a.shared_words + 4u * wave_words(p.K, C);
    uint64_t per_cu = lds_max / lds_words;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int cap_waves = env_int("MFA_WALK_WAVES_PER_CU", 0);      // development knob (multiples of 4)
    if (cap_waves > 0 && (uint64_t)(cap_waves + 3) / 4 < per_cu) per_cu = (uint64_t)(cap_waves + 3) / 4;
    uint64_t grid = (uint64_t)(n_cus > 0 ? n_cus : 256) * per_cu, want = (n + 255) / 256;
    if (grid > want) grid = want;
    L.grid = (unsigned)grid;
    L.reversed = p.reversed;
    const uint32_t W = 2 + 2 * p.K, DW = (1 + 2 * p.K + 1) / 2;
    const size_t need = (size_t)grid * 4u * a.CX * 64u * (3u * W + 3u * DW) * sizeof(uint32_t);
    int rc = ctx_reserve((void**)d_spill, spill_bytes, need);
    if (rc != MFA_OK) return rc;
    a.spill = *d_spill;
    HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), (hipStream_t)stream));
    switch (p.K) {
        case 1: return launch_walk_k1(L, stream); case 2: return launch_walk_k2(L, stream); case 3: return launch_walk_k3(L, stream);
        case 4: return launch_walk_k4(L, stream); case 5: return launch_walk_k5(L, stream); case 6: return launch_walk_k6(L, stream);
        case 7: return launch_walk_k7(L, stream); case 8: return launch_walk_k8(L, stream); case 9: return launch_walk_k9(L, stream);
    }
    return MFA_ERR_UNSUPPORTED;
}

}  // namespace mfa

using namespace mfa;

// ---- mixed batches ---------------------------------------------------------------------------------------------------------------------
struct mfa_mixed {
    std::vector<mfa_image*> images;
    std::vector<uint32_t>   words;          // the images' table blocks, back to back
    std::vector<uint32_t>   block_at;       // word offset of image k's block
    uint32_t K = 1, max_live = 1;
    bool reversed = false;
    std::mutex mu;
    struct Dev {
        uint32_t* d_tables = nullptr;
        hipStream_t rs = nullptr, ws = nullptr;                // region stream, walk stream
        std::vector<hipEvent_t> ev;                            // one per group: its regions are known
        hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_r0 = nullptr, ev_r1 = nullptr, ev_w1 = nullptr;
        uint64_t* d_regions = nullptr; size_t region_bytes = 0;
        uint32_t* d_spill[8] = {nullptr}; size_t spill_bytes[8] = {0};
        unsigned long long* d_counters = nullptr;
        int n_cus = 0;
        bool timed = false;
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
        if (!img || img->host.h.kind != MFA_KIND_MFA || !img->walk_ok) { delete mx; return img ? MFA_ERR_UNSUPPORTED : MFA_ERR_INVALID_ARG; }
        if (k == 0) mx->reversed = img->walk.reversed;
        else if (mx->reversed != img->walk.reversed) { delete mx; return MFA_ERR_UNSUPPORTED; }      // one scan direction per mixed batch
        mx->images.push_back(img);
        mx->K = std::max(mx->K, img->walk.K);
        mx->max_live = std::max(mx->max_live, img->walk.max_live);
    }
    for (mfa_image* img : mx->images) {
        mx->block_at.push_back((uint32_t)mx->words.size());
        if (mx->K > 6 && img->walk.K <= 6) {                  // a kernel for more than 6 cells reads 3-word edges
            WalkTables wide;
            if (build_walk_tables(img->host, wide, true) != MFA_OK) { delete mx; return MFA_ERR_UNSUPPORTED; }
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
        if (d.ws) (void)hipStreamSynchronize(d.ws);
        if (d.d_tables) (void)hipFree(d.d_tables);
        if (d.d_regions) (void)hipFree(d.d_regions);
        for (uint32_t* p : d.d_spill) if (p) (void)hipFree(p);
        if (d.d_counters) (void)hipFree(d.d_counters);
        for (hipEvent_t e : d.ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {d.ev_in, d.ev_out, d.ev_r0, d.ev_r1, d.ev_w1}) if (e) (void)hipEventDestroy(e);
        if (d.rs) (void)hipStreamDestroy(d.rs);
        if (d.ws) (void)hipStreamDestroy(d.ws);
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
    HIP_TRY(hipMalloc((void**)&d.d_tables, mx->words.size() * 4));
    HIP_TRY(hipMemcpy(d.d_tables, mx->words.data(), mx->words.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&d.d_counters, 64 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreateWithFlags(&d.rs, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&d.ws, hipStreamNonBlocking));
    for (hipEvent_t* e : {&d.ev_in, &d.ev_out}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (hipEvent_t* e : {&d.ev_r0, &d.ev_r1, &d.ev_w1}) HIP_TRY(hipEventCreate(e));
    auto ins = mx->dev.emplace(device, d);
    *out = &ins.first->second;
    return MFA_OK;
}

// One batch, n_images segments: strings seg_first[s] .. seg_first[s+1]-1 are matched against images[s] (the order of
// mfa_mixed_create).  seg_first is a HOST array of n_images + 1 indices, seg_first[0] = 0, seg_first[n_images] = n.
// The region pass runs group by group (runs of consecutive segments) on an internal stream, each group's walk on a second one as
// soon as the group's regions are known: walks run beside the next group's region pass.  `stream` sees the call as one operation.
int mfa_match_mixed(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, const uint64_t* seg_first,
                    uint8_t* d_results, int device, void* stream) {
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
    // groups: runs of consecutive segments, about n / G strings each, at most WALK_MAX_SEG segments
    int G = env_int("MFA_MIXED_GROUPS", 3);
    if (G < 1) G = 1;
    std::vector<uint32_t> gb{0};                              // group g = segments gb[g] .. gb[g+1]-1
    for (uint32_t s = 1; s < ns; s++) {
        const uint32_t g = (uint32_t)gb.size();
        const bool cut = seg_first[s] >= (n * g + G - 1) / G && g < (uint32_t)G;
        if (cut || s - gb.back() >= WALK_MAX_SEG) gb.push_back(s);
    }
    gb.push_back(ns);
    const uint32_t ng = (uint32_t)gb.size() - 1;
    if (ng > 8) return MFA_ERR_UNSUPPORTED;
    while (d->ev.size() < ng) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->ev.push_back(e);
    }
    const char* re = getenv("MFA_REGIONS");
    const bool with_regions = !(re && re[0] == '0');
    if (with_regions) {
        rc = ctx_reserve((void**)&d->d_regions, &d->region_bytes, (size_t)n * MFA_REGION_WORDS * sizeof(uint64_t));
        if (rc != MFA_OK) return rc;
    }
    HIP_TRY(hipEventRecord(d->ev_in, cs));
    HIP_TRY(hipStreamWaitEvent(d->rs, d->ev_in, 0));
    HIP_TRY(hipStreamWaitEvent(d->ws, d->ev_in, 0));
    HIP_TRY(hipEventRecord(d->ev_r0, d->rs));
    WalkPlanInput p{mx->K, mx->max_live, mx->reversed, (uint32_t)mx->words.size()};
    for (uint32_t g = 0; g < ng; g++) {
        const uint64_t lo = seg_first[gb[g]], hi = seg_first[gb[g + 1]];
        if (hi == lo) continue;
        if (with_regions) {
            rc = launch_region_scan(d->n_cus, d_bytes, d_offsets + lo, hi - lo, d->d_regions + lo * MFA_REGION_WORDS, d->rs);
            if (rc != MFA_OK) return rc;
        }
        HIP_TRY(hipEventRecord(d->ev[g], d->rs));
        HIP_TRY(hipStreamWaitEvent(d->ws, d->ev[g], 0));
        uint32_t sf[WALK_MAX_SEG + 1], stb[WALK_MAX_SEG];
        const uint32_t nseg = gb[g + 1] - gb[g];
        for (uint32_t k = 0; k <= nseg; k++) sf[k] = (uint32_t)(seg_first[gb[g] + k] - lo);
        for (uint32_t k = 0; k < nseg; k++) stb[k] = mx->block_at[gb[g] + k];
        rc = launch_walk(p, d->d_tables, d->n_cus, d_bytes, d_offsets + lo, hi - lo, d_results + lo,
                         with_regions ? d->d_regions + lo * MFA_REGION_WORDS : nullptr, nseg, sf, stb, &d->d_spill[g], &d->spill_bytes[g],
                         d->d_counters + g, d->ws);
        if (rc != MFA_OK) return rc;
    }
    HIP_TRY(hipEventRecord(d->ev_r1, d->rs));
    HIP_TRY(hipEventRecord(d->ev_w1, d->ws));
    HIP_TRY(hipEventRecord(d->ev_out, d->ws));
    HIP_TRY(hipStreamWaitEvent(cs, d->ev_out, 0));
    HIP_TRY(hipStreamWaitEvent(cs, d->ev_r1, 0));
    d->timed = true;
    return MFA_OK;
}

// Device time of the last mfa_match_mixed on `device`: the region launches (first to last, on their stream) and the whole call
// (first region launch to last walk).  Synchronises on the call's last events.
int mfa_mixed_last_ms(mfa_mixed_t* mx, int device, float* region_ms, float* span_ms) {
    if (!mx) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(mx->mu);
    auto it = mx->dev.find(device);
    if (it == mx->dev.end() || !it->second.timed) return MFA_ERR_INVALID_ARG;
    mfa_mixed::Dev& d = it->second;
    HIP_TRY(hipEventSynchronize(d.ev_w1));
    HIP_TRY(hipEventSynchronize(d.ev_r1));
    if (region_ms) HIP_TRY(hipEventElapsedTime(region_ms, d.ev_r0, d.ev_r1));
    if (span_ms) HIP_TRY(hipEventElapsedTime(span_ms, d.ev_r0, d.ev_w1));
    return MFA_OK;
}

}  // extern "C"
