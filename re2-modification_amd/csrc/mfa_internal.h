// Internal declarations shared by the host-side image code, the kernels' launchers
// and the C-ABI glue of libmfa_hip.so.  Not installed.
#ifndef MFA_INTERNAL_H
#define MFA_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mfa_hip.h"
#include "walk_tables.h"

namespace mfa {

// ---- host image ---------------------------------------------------------------------------

struct HostImage {
    mfa_blob_header            h{};
    std::vector<uint32_t>      edge_begin;  // n_nodes + 1
    std::vector<mfa_blob_edge> edges;

    // MFA_KIND_NFA only: the reference's step function (automata.cpp:98-128) tabulated
    // over its reachable state sets.  State 0 = the empty set (absorbing, rejects),
    // state 1 = {start}.
    uint32_t              dfa_states = 0;
    uint32_t              n_classes  = 0;
    uint8_t               byte_class[256] = {0};
    std::vector<uint32_t> dfa_trans;   // [dfa_states][n_classes]
    std::vector<uint8_t>  dfa_accept;  // [dfa_states]: finish is in the set after the final pass
    mutable int           jit_source_ok = -1;      // MFA kind: the generated kernel's source is of a size a compiler finishes (jit.hip; -1 = not looked at)
};

int parse_blob(const void* blob, size_t n_bytes, HostImage& out);   // MFA_OK / MFA_ERR_*
int check_mfa_invariants(const HostImage& img);                      // MFA_OK / MFA_ERR_UNSUPPORTED
int tabulate_nfa(HostImage& img);                                    // fills the dfa_* members

// ---- per-device state -----------------------------------------------------------------------

// Workspace of ONE launch: ticket counter, scratch, region table, events.  Every launch takes one from the
// pool of its (image, device); a context is reused by a later launch on the SAME stream (stream order makes
// that safe) or once its `done` event has completed, so launches of one image that overlap on different
// streams or from different host threads never share a counter, a scratch buffer or an event.
// Whether the lean kernel behind a table walk (walk.hip: strings without periodic stretches) has had anything to do lately: the kernel
// reports the length of its queue (+ 1) to a word of pinned host memory, and a launch whose slot last saw an empty queue leaves the lean
// kernel and the queue out (an empty launch beside a region pass costs the stream 0.1-0.35 ms: its workgroups queue for room) -- except
// every 32nd time, with a quarter of the grid, to notice when the input changes.
struct LeanHint {
    uint32_t* h_seen = nullptr;      // pinned, device-visible; 0 = nothing reported yet
    uint32_t quiet = 0, launches = 0;
};
void lean_hint_free(LeanHint& h);

struct LaunchCtx {
    LeanHint lean;
    unsigned long long* d_counter = nullptr;   // next-string ticket (+ MFA_STATS words)
    uint32_t*           d_scratch = nullptr;   // slot arrays / probe images that do not fit LDS
    size_t              scratch_bytes = 0;
    uint64_t*           d_regions = nullptr;   // region table of the batch (regions.hip)
    size_t              region_bytes = 0;
    void*               ev_start = nullptr;    // hipEvent_t: around the match kernel
    void*               ev_stop  = nullptr;
    void*               ev_r0 = nullptr;       // around the region pass
    void*               ev_r1 = nullptr;
    void*               ev_done = nullptr;     // after the last kernel of the launch (no timing)
    void*               stream = nullptr;      // stream of the launch it was last used for
    bool                used = false;
    bool                ran_regions = false;
};

struct DeviceState {
    int       device = -1;
    // NFA kind
    void*     d_dfa_trans  = nullptr;   // [dfa_states][n_classes], 16-bit entries up to 65535 state sets, 32-bit beyond
    uint8_t*  d_dfa_accept = nullptr;
    uint8_t*  d_byte_class = nullptr;
    // launch workspaces
    std::vector<LaunchCtx*> ctxs;
    LaunchCtx*              last = nullptr;    // context of the most recent launch (mfa_last_kernel_ms)
    int                 n_cus = 0;
    // specialised kernel (jit.hip), if one is loaded for this device
    bool                jit_tried = false;
    bool                jit_probed = false, jit_in_cache = false;      // automatic mode: looked whether the code object is cached
    void*               jit_mod = nullptr;     // hipModule_t
    void*               jit_fn  = nullptr;     // hipFunction_t
    int                 jit_waves_per_cu = 0;
    uint32_t            jit_words = 0;         // words per slot set of the specialised kernel
    uint32_t            jit_lanes = 64;        // string-carrying lanes per wave
    // live-list walk (walk.hip): the image's tables on this device
    uint32_t*           d_walk = nullptr;
};

}  // namespace mfa

struct mfa_image {
    mfa::HostImage                    host;
    std::mutex                        mu;
    std::map<int, mfa::DeviceState>   dev;
    uint32_t                          last_kernel = 0;   // MFA_KERNEL_*
    mfa::WalkTables                   walk;              // MFA kind: tables of the live-list walk (walk_tables.h)
    bool                              walk_ok = false;   //   ... built (false: the automaton exceeds the table format)
};

namespace mfa {

// launchers (kernels.hip); all asynchronous on `stream`
int launch_dfa_walk(const HostImage& img, DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                    uint64_t n, uint8_t* d_results, void* stream);
// regions.hip
// gate: the table has the gate header in front of it (regions.hip: gate_signal) and every finished string is counted for its group
constexpr uint32_t MFA_GATE_MAX_GROUPS = 32;
constexpr size_t MFA_GATE_FIXED_WORDS = 3 * MFA_GATE_MAX_GROUPS;                                  // the groups' ends, the call's stamp (regions.hip: gate_signal)
constexpr size_t MFA_GATE_HEADER_WORDS = MFA_GATE_FIXED_WORDS + MFA_GATE_MAX_GROUPS * 64 * 8;      // + a 64-byte line per (group, residue mod 64)      // u64 words in front of the table (a multiple of 16: the rows stay 128-byte aligned)
int launch_region_scan(int n_cus, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t* d_table, void* stream, unsigned threads = 256, bool gate = false, void* done_event = nullptr);
int launch_gate_wait(const uint64_t* d_table, uint32_t group, void* stream);      // one wave that ends when the gated region launch has counted every string of `group`
// launch contexts (capi.hip); the caller holds the image mutex
int  ctx_acquire(DeviceState& ds, void* stream, LaunchCtx** out);
int  ctx_reserve(void** buf, size_t* have, size_t need);
int device_prepare(mfa_image* img, int device, DeviceState** out);
// specialised kernels (jit_gen.cpp, jit.hip)
uint32_t    jit_slot_registers(const HostImage& img);
uint32_t    jit_lanes(const HostImage& img);
std::string jit_generate_source(const HostImage& img);
bool        jit_enabled(const HostImage& img);
bool        jit_cached(const HostImage& img);      // its code object is in the cache already
std::string jit_compile(const HostImage& img, std::string* err);
bool        jit_load(const HostImage& img, DeviceState& ds);
void        jit_unload(DeviceState& ds);
void        jit_print_stats(LaunchCtx& cx, const char* tag);
int         launch_mfa_jit(DeviceState& ds, LaunchCtx& cx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                           const uint64_t* d_regions, void* stream);
void device_release(DeviceState& ds);
// live-list walk (walk_launch.hip).  seg_first: n_seg + 1 string indices relative to the sub-batch; seg_table: word offsets into d_tables
struct WalkPlanInput { uint32_t K, max_live; bool reversed; uint32_t table_words; };
int launch_walk(const WalkPlanInput& p, const uint32_t* d_tables, int n_cus, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                uint8_t* d_results, const uint64_t* d_regions, uint32_t n_seg, const uint32_t* seg_first, const uint32_t* seg_table,
                uint32_t** d_spill, size_t* spill_bytes, unsigned long long* d_counter, void* stream, uint32_t gate = 0, LeanHint* lean = nullptr, void* wait_event = nullptr);
int  walk_mode();          // MFA_WALK: 0 auto (default), 1 table, 2 jit
void set_last_hip_error(int e);

}  // namespace mfa

#endif
