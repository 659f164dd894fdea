// Specialised kernels: source from jit_gen.cpp, compiled for gfx950 with hipcc --genco into a code
// object that is cached on disk (keyed by the automaton image), loaded with hipModuleLoad and launched
// with hipModuleLaunchKernel.  Everything here is optional: if generation, compilation or loading
// fails the launcher uses the generic kernel of kernels.hip (still on the GPU).
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "mfa_internal.h"

extern char** environ;

namespace mfa {

static const char* kCompileFlags[] = {"--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Rpass-analysis=kernel-resource-usage"};

static std::string cache_dir() {
    if (const char* e = getenv("MFA_JIT_CACHE")) return e;
    Dl_info info;
    if (dladdr((void*)&cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t slash = p.rfind('/');
        return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/jit_cache";
    }
    char buf[64];
    snprintf(buf, sizeof buf, "/tmp/mfa_jit_cache_%u", (unsigned)getuid());      // per user, created 0700
    return buf;
}

static uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t k = 0; k < n; k++) { h ^= p[k]; h *= 1099511628211ull; }
    return h;
}

static bool stats_build() {
    const char* e = getenv("MFA_STATS");
    return e && e[0] != '0' && e[0] != 0;
}

// The key covers everything the code object depends on: the generated source itself (which embeds
// device_common.h, the automaton and the development knobs) and the compile flags.
static std::string source_key(const std::string& src) {
    uint64_t h = 1469598103934665603ull, g = 0x9e3779b97f4a7c15ull;
    h = fnv1a(src.data(), src.size(), h);
    for (const char* f : kCompileFlags) { h = fnv1a(f, strlen(f), h); g = fnv1a(f, strlen(f), g); }
    if (stats_build()) { h = fnv1a("stats", 5, h); g = fnv1a("stats", 5, g); }
    g = fnv1a(src.data(), src.size(), g);
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%08x", (unsigned long long)h, (unsigned)(g >> 32));
    return buf;
}

static bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && st.st_size > 0;
}

bool jit_enabled(const HostImage& img) {
    const char* e = getenv("MFA_JIT");
    if (e && e[0] == '0') return false;
    // up to 272 slot registers in VGPRs, beyond that slot sets in LDS; the generated code keeps node sets in 128-bit masks
    // ... and unrolls every edge, through every chain of absent-cell edges: beyond a few hundred edges, or 1.5 MB of generated
    // source, the compiler takes longer than any batch (ex. 15 -reverse: 939 edges; ex. 15 -bnf: 69 edges, 4 MB of source)
    if (!(img.h.kind == MFA_KIND_MFA && img.h.n_nodes <= 128u && img.h.n_edges <= 400u && jit_lanes(img) != 0)) return false;
    if (img.jit_source_ok < 0) img.jit_source_ok = jit_generate_source(img).size() <= 1536u * 1024u ? 1 : 0;
    return img.jit_source_ok == 1;
}

bool jit_cached(const HostImage& img) {
    if (!jit_enabled(img)) return false;
    return file_exists(cache_dir() + "/" + source_key(jit_generate_source(img)) + ".hsaco");
}


// generate + compile into the cache (host-only, no GPU needed); returns the code-object path or ""
std::string jit_compile(const HostImage& img, std::string* err) {
    if (!jit_enabled(img)) { if (err) *err = "automaton too large for the specialised kernel"; return ""; }
    const std::string text = jit_generate_source(img);
    const std::string dir = cache_dir(), key = source_key(text);
    const std::string obj = dir + "/" + key + ".hsaco";
    mkdir(dir.c_str(), 0700);
    {   // the cache holds code that is loaded onto the GPU: it must be a directory of ours that nobody else can write to
        struct stat st;
        if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 022) != 0) {
            if (err) *err = "cache directory " + dir + " is not a directory owned by this user and closed to others";
            return "";
        }
    }
    if (file_exists(obj)) return obj;
    static std::atomic<unsigned> serial{0};
    char tmpl[64];
    snprintf(tmpl, sizeof tmpl, ".tmp%d_%u", (int)getpid(), serial.fetch_add(1));
    // every compiler run works on files of its own (source, object, log) and renames them into place when done:
    // ranks and threads that meet on a cold cache never read each other's half-written files
    const std::string src = dir + "/" + key + tmpl + ".hip", tmp = obj + tmpl, log = tmp + ".log";
    {
        std::ofstream f(src);
        if (!f.is_open()) { if (err) *err = "cannot write " + src; return ""; }
        f << text;
    }
    const char* hipcc = getenv("MFA_HIPCC");
    const std::string cc = hipcc ? hipcc : "/opt/rocm/bin/hipcc";
    std::vector<std::string> words = {cc};
    for (const char* f : kCompileFlags) words.push_back(f);
    if (stats_build()) words.push_back("-DMFA_STATS_BUILD=1");
    words.push_back("-o"); words.push_back(tmp); words.push_back(src);
    std::vector<char*> argv;
    for (auto& w : words) argv.push_back(const_cast<char*>(w.c_str()));
    argv.push_back(nullptr);
    // the resource-usage remarks (VGPRs, LDS, scratch, occupancy) of a successful build are kept next to the object
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid;
    const int sp = posix_spawn(&pid, cc.c_str(), &fa, nullptr, argv.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (sp != 0) {
        unlink(src.c_str());
        if (err) *err = "cannot start the compiler " + cc;
        return "";
    }
    int status = 0;
    if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0 || !file_exists(tmp)) {
        unlink(tmp.c_str());
        if (file_exists(obj)) { unlink(log.c_str()); unlink(src.c_str()); return obj; }      // another thread or process got there first
        rename(src.c_str(), (dir + "/" + key + ".failed.hip").c_str());
        if (err) *err = "hipcc failed, see " + log;
        return "";
    }
    rename(src.c_str(), (dir + "/" + key + ".hip").c_str());
    rename(log.c_str(), (dir + "/" + key + ".log").c_str());
    rename(tmp.c_str(), obj.c_str());
    return obj;
}

// caller holds the image mutex and has set the device
bool jit_load(const HostImage& img, DeviceState& ds) {
    if (ds.jit_tried) return ds.jit_fn != nullptr;
    ds.jit_tried = true;
    std::string err;
    std::string obj = jit_compile(img, &err);
    if (obj.empty()) {
        if (jit_enabled(img)) fprintf(stderr, "mfa_hip: specialised kernel unavailable (%s); using the table-driven walk\n", err.c_str());
        return false;
    }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoad(&mod, obj.c_str()) != hipSuccess || hipModuleGetFunction(&fn, mod, "mfa_jit_kernel") != hipSuccess) {
        fprintf(stderr, "mfa_hip: cannot load %s; using the table-driven walk\n", obj.c_str());
        if (mod) (void)hipModuleUnload(mod);
        return false;
    }
    int blocks = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 64, 0) != hipSuccess || blocks <= 0) blocks = 4;
    ds.jit_mod = mod; ds.jit_fn = fn; ds.jit_waves_per_cu = blocks > 32 ? 32 : blocks;
    ds.jit_words = jit_slot_registers(img) / 2;
    ds.jit_lanes = jit_lanes(img);
    return true;
}

int launch_mfa_jit(DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                   const uint64_t* d_regions, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(cx.d_counter, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) { set_last_hip_error((int)e); return MFA_ERR_HIP; }
    uint64_t grid = (uint64_t)ds.n_cus * ds.jit_waves_per_cu, want = (n + ds.jit_lanes - 1) / ds.jit_lanes;
    if (const char* g = getenv("MFA_WALK_WAVES_PER_CU")) {            // development knob: waves per CU of the walk launch
        const int w = atoi(g);
        if (w > 0 && w < ds.jit_waves_per_cu) grid = (uint64_t)ds.n_cus * w;
    }
    if (grid > want) grid = want;
    if (grid == 0) grid = 1;
    const char* ae = getenv("MFA_ACCEL");
    uint32_t accel = (ae && ae[0] == '0') ? 0u : 1u;                  // MFA_ACCEL=0: execute every step (A/B testing)
    // probe storage: three slot-set images per wave (jit_gen.cpp)
    int rc = ctx_reserve((void**)&cx.d_scratch, &cx.scratch_bytes, (size_t)grid * 3u * ds.jit_words * 64u * sizeof(uint32_t));
    if (rc != MFA_OK) return rc;
    // MFA_STATS=1: the kernel adds its iteration / probe counters to 6 words behind the ticket counter
    unsigned long long* stats = getenv("MFA_STATS") ? cx.d_counter + 1 : nullptr;      // + per-string executed steps at stats+8 (u32, first 1M strings)
    if (stats) (void)hipMemsetAsync(stats, 0, 16 * sizeof(unsigned long long), s);
    void* args[] = {(void*)&d_bytes, (void*)&d_offsets, (void*)&n, (void*)&d_results, (void*)&cx.d_counter, (void*)&accel,
                    (void*)&cx.d_scratch, (void*)&stats, (void*)&d_regions};
    (void)hipEventRecord((hipEvent_t)cx.ev_start, s);
    e = hipModuleLaunchKernel((hipFunction_t)ds.jit_fn, (unsigned)grid, 1, 1, 64, 1, 1, 0, s, args, nullptr);
    if (e != hipSuccess) { set_last_hip_error((int)e); return MFA_ERR_HIP; }
    (void)hipEventRecord((hipEvent_t)cx.ev_stop, s);
    return MFA_OK;
}

// debugging aid: copy the counters of the last launch (MFA_STATS=1) to the host and print them
void jit_print_stats(LaunchCtx& cx, const char* tag) {
    if (!getenv("MFA_STATS") || !cx.d_counter) return;
    unsigned long long h[17] = {0};
    if (hipMemcpy(h, cx.d_counter, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (const char* path = getenv("MFA_STATS_FILE")) {
        std::vector<uint32_t> steps(1u << 20);
        if (hipMemcpy(steps.data(), cx.d_counter + 17, steps.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(path, "wb");
            if (f) { fwrite(steps.data(), 4, steps.size(), f); fclose(f); }
        }
    }
    fprintf(stderr, "mfa_hip stats %s: wave-iterations %llu (dual %llu), lane-steps skipped %llu, probes %llu (hits %llu), scans %llu\n", tag,
            h[1], h[2], h[3], h[4], h[5], h[6]);
    fprintf(stderr, "mfa_hip stats %s: failed probes -- never settled %llu, dual period %llu, out of periodic input %llu\n", tag, h[11], h[12], h[13]);
    if (h[10]) fprintf(stderr, "mfa_hip stats %s: share of wave time -- tickets + string start %.1f%%, input byte %.1f%%, region look-up %.1f%%, period scans %.1f%%, plain steps %.1f%%, dual steps %.1f%% (steps include cell-read scans); %.1f M wave-cycles in all\n", tag,
                      100.0 * h[14] / h[10], 100.0 * h[15] / h[10], 100.0 * h[16] / h[10], 100.0 * h[7] / h[10], 100.0 * h[8] / h[10], 100.0 * h[9] / h[10], h[10] / 1e6);
}

void jit_unload(DeviceState& ds) {
    if (ds.jit_mod) (void)hipModuleUnload((hipModule_t)ds.jit_mod);
    ds.jit_mod = nullptr; ds.jit_fn = nullptr;
}

}  // namespace mfa
