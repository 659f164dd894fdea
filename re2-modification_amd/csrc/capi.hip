// C-ABI of libmfa_hip.so (include/mfa_hip.h): glue between plain-C callers and the kernels.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "mfa_internal.h"

namespace mfa {

static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) { set_last_hip_error((int)e_); return MFA_ERR_HIP; } \
    } while (0)

static void ctx_free(LaunchCtx& cx) {
    if (cx.d_counter) (void)hipFree(cx.d_counter);
    lean_hint_free(cx.lean);
    if (cx.d_scratch) (void)hipFree(cx.d_scratch);
    if (cx.d_regions) (void)hipFree(cx.d_regions);
    for (void* ev : {cx.ev_start, cx.ev_stop, cx.ev_r0, cx.ev_r1, cx.ev_done})
        if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    cx = LaunchCtx{};
}

// grow *buf to `need` bytes (contents are not kept).  Frees the old buffer first: hipFree waits for the device, so a
// kernel of an earlier launch on the same stream that still uses it has finished by then.
int ctx_reserve(void** buf, size_t* have, size_t need) {
    if (need <= *have) return MFA_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr; *have = 0;
    const size_t want = need + need / 4;                      // head room: batches of similar size do not reallocate
    hipError_t e = hipMalloc(buf, want);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {      // not with head room: exactly what is needed, or an honest "no memory"
        (void)hipGetLastError();
        e = hipMalloc(buf, need);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) { (void)hipGetLastError(); *buf = nullptr; set_last_hip_error((int)e); return MFA_ERR_NOMEM; }
        if (e == hipSuccess) { *have = need; return MFA_OK; }
    }
    if (e != hipSuccess) { *buf = nullptr; set_last_hip_error((int)e); return MFA_ERR_HIP; }
    *have = want;
    return MFA_OK;
}

// A context for a launch on `stream`: the one this stream used last (its work is ordered before ours), else one whose
// last launch has completed, else a new one.  The caller holds the image mutex and has set the device.
int ctx_acquire(DeviceState& ds, void* stream, LaunchCtx** out) {
    LaunchCtx* pick = nullptr;
    for (LaunchCtx* cx : ds.ctxs)
        if (cx->used && cx->stream == stream) { pick = cx; break; }
    if (!pick)
        for (LaunchCtx* cx : ds.ctxs)
            if (!cx->used || hipEventQuery((hipEvent_t)cx->ev_done) == hipSuccess) { pick = cx; break; }
    if (!pick && ds.ctxs.size() >= 64) {                       // far more overlapping launches than a device runs at once: wait for the oldest
        pick = ds.ctxs.front();
        HIP_TRY(hipEventSynchronize((hipEvent_t)pick->ev_done));
    }
    if (!pick) {
        LaunchCtx* cx = new (std::nothrow) LaunchCtx();
        if (!cx) return MFA_ERR_NOMEM;
        hipError_t e = hipMalloc((void**)&cx->d_counter, 256 + (getenv("MFA_STATS") ? (4u << 20) : 0));
        if (e == hipSuccess) e = hipEventCreate((hipEvent_t*)&cx->ev_start);
        if (e == hipSuccess) e = hipEventCreate((hipEvent_t*)&cx->ev_stop);
        if (e == hipSuccess) e = hipEventCreate((hipEvent_t*)&cx->ev_r0);
        if (e == hipSuccess) e = hipEventCreate((hipEvent_t*)&cx->ev_r1);
        if (e == hipSuccess) e = hipEventCreateWithFlags((hipEvent_t*)&cx->ev_done, hipEventDisableTiming);
        if (e != hipSuccess) { set_last_hip_error((int)e); ctx_free(*cx); delete cx; return MFA_ERR_HIP; }
        ds.ctxs.push_back(cx);
        pick = cx;
    }
    pick->used = true; pick->stream = stream; pick->ran_regions = false;
    ds.last = pick;
    *out = pick;
    return MFA_OK;
}

void device_release(DeviceState& ds) {
    if (ds.device < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    (void)hipSetDevice(ds.device);
    if (ds.d_dfa_trans) (void)hipFree(ds.d_dfa_trans);
    if (ds.d_dfa_accept) (void)hipFree(ds.d_dfa_accept);
    if (ds.d_byte_class) (void)hipFree(ds.d_byte_class);
    if (ds.d_walk) (void)hipFree(ds.d_walk);
    for (LaunchCtx* cx : ds.ctxs) { ctx_free(*cx); delete cx; }
    jit_unload(ds);
    ds = DeviceState{};
    (void)hipSetDevice(cur);
}

// caller holds img->mu
int device_prepare(mfa_image* img, int device, DeviceState** out) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MFA_ERR_NO_DEVICE;
    auto it = img->dev.find(device);
    if (it != img->dev.end()) { *out = &it->second; return MFA_OK; }
    HIP_TRY(hipSetDevice(device));
    DeviceState ds;
    ds.device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    ds.n_cus = prop.multiProcessorCount;
    const HostImage& h = img->host;
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 4));
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return MFA_OK;
    };
    int rc = MFA_OK;
    if (h.h.kind == MFA_KIND_MFA) {
        if (img->walk_ok) rc = up((void**)&ds.d_walk, img->walk.words.data(), img->walk.words.size() * 4);
    } else {
        if (h.dfa_states <= 0xffffu) {
            std::vector<uint16_t> t16(h.dfa_trans.begin(), h.dfa_trans.end());
            rc = up((void**)&ds.d_dfa_trans, t16.data(), t16.size() * 2);
        } else rc = up((void**)&ds.d_dfa_trans, h.dfa_trans.data(), h.dfa_trans.size() * 4);
        if (rc == MFA_OK) rc = up((void**)&ds.d_dfa_accept, h.dfa_accept.data(), h.dfa_accept.size());
        if (rc == MFA_OK) rc = up((void**)&ds.d_byte_class, h.byte_class, 256);
    }
    if (rc != MFA_OK) { device_release(ds); return rc; }
    auto ins = img->dev.emplace(device, ds);
    *out = &ins.first->second;
    return MFA_OK;
}

}  // namespace mfa

using namespace mfa;

extern "C" {

int mfa_image_create(const void* blob, size_t n_bytes, mfa_image_t** out) {
    if (!blob || !out) return MFA_ERR_INVALID_ARG;
    *out = nullptr;
    mfa_image* img = new (std::nothrow) mfa_image();
    if (!img) return MFA_ERR_NOMEM;
    int rc = parse_blob(blob, n_bytes, img->host);
    if (rc == MFA_OK) rc = img->host.h.kind == MFA_KIND_MFA ? check_mfa_invariants(img->host) : tabulate_nfa(img->host);
    if (rc != MFA_OK) { delete img; return rc; }
    if (img->host.h.kind == MFA_KIND_MFA) {
        img->walk_ok = build_walk_tables(img->host, img->walk) == MFA_OK;
        if (!img->walk_ok && !jit_enabled(img->host)) { delete img; return MFA_ERR_UNSUPPORTED; }      // no kernel could walk it
    }
    *out = img;
    return MFA_OK;
}

void mfa_image_destroy(mfa_image_t* img) {
    if (!img) return;
    for (auto& kv : img->dev) device_release(kv.second);
    delete img;
}

int mfa_image_get_info(const mfa_image_t* img, mfa_image_info* out) {
    if (!img || !out) return MFA_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->kind = img->host.h.kind; out->is_reversed = img->host.h.is_reversed;
    out->n_nodes = img->host.h.n_nodes; out->n_edges = img->host.h.n_edges; out->n_cells = img->host.h.n_cells;
    out->dfa_states = img->host.dfa_states; out->byte_classes = img->host.n_classes;
    out->last_kernel = img->last_kernel;
    return MFA_OK;
}

int mfa_image_prepare(mfa_image_t* img, int device) {
    if (!img) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(img->mu);
    DeviceState* ds = nullptr;
    int rc = device_prepare(img, device, &ds);
    if (rc == MFA_OK && img->host.h.kind == MFA_KIND_MFA && walk_mode() != 1 && (walk_mode() == 2 || !img->walk_ok || (jit_lanes(img->host) == 64u && jit_cached(img->host))))
        (void)jit_load(img->host, *ds);
    return rc;
}

int mfa_image_specialize(mfa_image_t* img) {
    if (!img) return MFA_ERR_INVALID_ARG;
    if (!jit_enabled(img->host)) return MFA_ERR_UNSUPPORTED;
    std::string err;
    if (jit_compile(img->host, &err).empty()) {
        fprintf(stderr, "mfa_hip: %s\n", err.c_str());
        return MFA_ERR_JIT;
    }
    return MFA_OK;
}

static bool regions_enabled() {
    const char* e = getenv("MFA_REGIONS");                      // MFA_REGIONS=0: no region pass and no table: every step is executed (A/B runs)
    const char* a = getenv("MFA_ACCEL");
    return !(e && e[0] == '0') && !(a && a[0] == '0');
}

// own_regions: run the region pass into the context's table; else use d_table (may be NULL)
static int match_impl(mfa_image_t* img, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                      const uint64_t* d_table, bool own_regions, int device, void* stream) {
    if (!img || !d_offsets || (!d_results && n)) return MFA_ERR_INVALID_ARG;
    if (n == 0) return MFA_OK;
    std::lock_guard<std::mutex> lk(img->mu);
    DeviceState* ds = nullptr;
    int rc = device_prepare(img, device, &ds);
    if (rc != MFA_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    const bool is_mfa = img->host.h.kind == MFA_KIND_MFA;
    // Which walk: the table-driven kernel (walk.hip) works for every automaton at once; a kernel generated for this automaton
    // (jit_gen.cpp) steps faster on small automata but has to be compiled first.  Automatic: the generated kernel if its code object
    // is in the cache already (mfa_image_specialize builds it ahead of time), is not of the "huge" kind, and MFA_JIT != 0 -- a
    // match call never waits for a compiler.  MFA_WALK=table / MFA_WALK=jit force one or the other (jit compiles on demand).
    const int mode = walk_mode();
    bool want_jit = false;
    if (is_mfa) {
        if (mode == 2 || !img->walk_ok) want_jit = jit_enabled(img->host);
        else if (mode == 0) {
            if (!ds->jit_tried && !ds->jit_probed) { ds->jit_probed = true; ds->jit_in_cache = jit_lanes(img->host) == 64u && jit_cached(img->host); }
            want_jit = ds->jit_fn != nullptr || ds->jit_in_cache;
        }
    }
    const bool jit = want_jit && jit_load(img->host, *ds);
    const bool table_walk = is_mfa && !jit && img->walk_ok;
    if (is_mfa && !jit && !table_walk) return MFA_ERR_JIT;      // only a generated kernel could walk it, and none could be built
    // MFA_REQUIRE_JIT=1: a caller that counts on the specialised kernel's step rate gets an error instead of the other engine;
    // MFA_VERBOSE=1: which kernel walks, on stderr
    if (is_mfa && !jit) { const char* rq = getenv("MFA_REQUIRE_JIT"); if (rq && rq[0] == '1') return MFA_ERR_JIT; }
    if (const char* vb = getenv("MFA_VERBOSE"))
        if (vb[0] == '1') fprintf(stderr, "mfa_hip: %s\n", !is_mfa ? "table walk of the tabulated automaton (dfa_*_kernel)" : jit ? "kernel generated for this automaton (mfa_jit_kernel)" : "table-driven walk (walk_kernel)");
    LaunchCtx* cx = nullptr;
    rc = ctx_acquire(*ds, stream, &cx);
    if (rc != MFA_OK) return rc;
    // whatever happens below, the context's `done` event is recorded behind what was enqueued: a launch on another stream must not
    // be handed this context (its counter, scratch area and table) while a kernel of this call may still be using it
    struct DoneGuard {
        LaunchCtx* cx; void* stream;
        ~DoneGuard() { (void)hipEventRecord((hipEvent_t)cx->ev_done, (hipStream_t)stream); }
    } done_guard{cx, stream};
    if (table_walk) {
        img->last_kernel = MFA_KERNEL_WALK;
        if (own_regions && regions_enabled()) {
            rc = ctx_reserve((void**)&cx->d_regions, &cx->region_bytes, (size_t)n * MFA_REGION_WORDS * sizeof(uint64_t));
            if (rc != MFA_OK) return rc;
            HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_r0, (hipStream_t)stream));
            rc = launch_region_scan(ds->n_cus, d_bytes, d_offsets, n, cx->d_regions, stream);
            if (rc != MFA_OK) return rc;
            HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_r1, (hipStream_t)stream));
            cx->ran_regions = true;
            d_table = cx->d_regions;
        }
        const WalkPlanInput p{img->walk.K, img->walk.max_live, img->walk.reversed, (uint32_t)img->walk.words.size()};
        const uint32_t sf[2] = {0u, (uint32_t)n}, stb[1] = {0u};
        if (n > 0xffffffffull) return MFA_ERR_INVALID_ARG;
        HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_start, (hipStream_t)stream));
        rc = launch_walk(p, ds->d_walk, ds->n_cus, d_bytes, d_offsets, n, d_results, d_table, 1u, sf, stb, &cx->d_scratch, &cx->scratch_bytes,
                         cx->d_counter, stream, 0u, &cx->lean);
        HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_stop, (hipStream_t)stream));
    } else if (jit) {
        img->last_kernel = MFA_KERNEL_SPECIALISED;
        if (own_regions && regions_enabled()) {
            rc = ctx_reserve((void**)&cx->d_regions, &cx->region_bytes, (size_t)n * MFA_REGION_WORDS * sizeof(uint64_t));
            if (rc != MFA_OK) return rc;
            HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_r0, (hipStream_t)stream));
            rc = launch_region_scan(ds->n_cus, d_bytes, d_offsets, n, cx->d_regions, stream);
            if (rc != MFA_OK) return rc;
            HIP_TRY(hipEventRecord((hipEvent_t)cx->ev_r1, (hipStream_t)stream));
            cx->ran_regions = true;
            d_table = cx->d_regions;
        }
        rc = launch_mfa_jit(*ds, *cx, d_bytes, d_offsets, n, d_results, d_table, stream);
    } else {
        img->last_kernel = MFA_KERNEL_TABLE;
        rc = launch_dfa_walk(img->host, *ds, *cx, d_bytes, d_offsets, n, d_results, stream);
    }
    return rc;
}

int mfa_match_batch(mfa_image_t* img, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                    int device, void* stream) {
    return match_impl(img, d_bytes, d_offsets, n, d_results, nullptr, true, device, stream);
}

int mfa_match_batch_regions(mfa_image_t* img, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                            const uint64_t* d_table, int device, void* stream) {
    return match_impl(img, d_bytes, d_offsets, n, d_results, d_table, false, device, stream);
}

int mfa_region_scan(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t* d_table, int device, void* stream) {
    if (!d_offsets || (!d_table && n)) return MFA_ERR_INVALID_ARG;
    if (n == 0) return MFA_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MFA_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    static int cus[64] = {0};
    if (device < 64 && cus[device] == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        cus[device] = prop.multiProcessorCount;
    }
    return launch_region_scan(device < 64 ? cus[device] : 256, d_bytes, d_offsets, n, d_table, stream);
}

int mfa_match_batch_host(mfa_image_t* img, const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* results,
                         int device) {
    if (!img || !offsets || (!results && n)) return MFA_ERR_INVALID_ARG;
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
    int rc = MFA_OK;
    hipError_t e = hipMalloc((void**)&d_bytes, total + 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, (n + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_res, n);
    if (e == hipSuccess && total) e = hipMemcpy(d_bytes, bytes + offsets[0], total, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (offsets[0] == 0) e = hipMemcpy(d_off, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
        else {
            uint64_t* tmp = new (std::nothrow) uint64_t[n + 1];
            if (!tmp) rc = MFA_ERR_NOMEM;
            else {
                for (uint64_t k = 0; k <= n; k++) tmp[k] = offsets[k] - offsets[0];
                e = hipMemcpy(d_off, tmp, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
                delete[] tmp;
            }
        }
    }
    if (e != hipSuccess) { set_last_hip_error((int)e); rc = MFA_ERR_HIP; }
    if (rc == MFA_OK) rc = mfa_match_batch(img, d_bytes, d_off, n, d_res, device, nullptr);
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

int mfa_last_kernel_ms(mfa_image_t* img, int device, float* ms) {
    if (!img || !ms) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(img->mu);
    auto it = img->dev.find(device);
    if (it == img->dev.end() || !it->second.last) return MFA_ERR_INVALID_ARG;
    LaunchCtx& cx = *it->second.last;
    HIP_TRY(hipEventSynchronize((hipEvent_t)cx.ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)cx.ev_start, (hipEvent_t)cx.ev_stop));
    jit_print_stats(cx, "last kernel");
    return MFA_OK;
}

int mfa_last_region_ms(mfa_image_t* img, int device, float* ms) {
    if (!img || !ms) return MFA_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(img->mu);
    auto it = img->dev.find(device);
    if (it == img->dev.end() || !it->second.last) return MFA_ERR_INVALID_ARG;
    LaunchCtx& cx = *it->second.last;
    *ms = 0.0f;
    if (!cx.ran_regions) return MFA_OK;
    HIP_TRY(hipEventSynchronize((hipEvent_t)cx.ev_r1));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)cx.ev_r0, (hipEvent_t)cx.ev_r1));
    return MFA_OK;
}

int mfa_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return MFA_ERR_NO_DEVICE;
    return count;
}

int mfa_last_hip_error(void) { return g_last_hip_error; }

const char* mfa_strerror(int code) {
    switch (code) {
        case MFA_OK: return "ok";
        case MFA_ERR_INVALID_ARG: return "invalid argument";
        case MFA_ERR_BAD_BLOB: return "malformed automaton image blob";
        case MFA_ERR_UNSUPPORTED: return "automaton outside the limits of the device kernels";
        case MFA_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
        case MFA_ERR_HIP: return "HIP runtime error";
        case MFA_ERR_NOMEM: return "out of host memory";
        case MFA_ERR_TOO_LONG: return "string longer than MFA_MAX_STRING_BYTES";
        case MFA_ERR_JIT: return "compiling the specialised kernel failed";
    }
    return "unknown error";
}

const char* mfa_version(void) { return "mfa_hip 0.1 (gfx950)"; }

}  // extern "C"
