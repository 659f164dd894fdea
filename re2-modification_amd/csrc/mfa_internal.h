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
    std::vector<uint16_t> dfa_trans;   // [dfa_states][n_classes]
    std::vector<uint8_t>  dfa_accept;  // [dfa_states]: finish is in the set after the final pass
};

int parse_blob(const void* blob, size_t n_bytes, HostImage& out);   // MFA_OK / MFA_ERR_*
int check_mfa_invariants(const HostImage& img);                      // MFA_OK / MFA_ERR_UNSUPPORTED
int tabulate_nfa(HostImage& img);                                    // fills the dfa_* members

// ---- per-device state -----------------------------------------------------------------------

struct DeviceState {
    int       device = -1;
    // MFA kind
    uint32_t* d_edge_begin = nullptr;
    uint2*    d_edges      = nullptr;   // mfa_blob_edge reinterpreted as two dwords
    // NFA kind
    uint16_t* d_dfa_trans  = nullptr;
    uint8_t*  d_dfa_accept = nullptr;
    uint8_t*  d_byte_class = nullptr;
    // launch workspace
    unsigned long long* d_counter = nullptr;   // next-string ticket
    uint32_t*           d_scratch = nullptr;   // slot arrays for automata too large for LDS
    size_t              scratch_bytes = 0;
    void*               ev_start = nullptr;    // hipEvent_t
    void*               ev_stop  = nullptr;
    bool                timed = false;
    int                 n_cus = 0;
    // specialised kernel (jit.hip), if one is loaded for this device
    bool                jit_tried = false;
    void*               jit_mod = nullptr;     // hipModule_t
    void*               jit_fn  = nullptr;     // hipFunction_t
    int                 jit_waves_per_cu = 0;
    uint32_t            jit_words = 0;         // words per slot set of the specialised kernel
    uint32_t            jit_lanes = 64;        // string-carrying lanes per wave
};

}  // namespace mfa

struct mfa_image {
    mfa::HostImage                    host;
    std::mutex                        mu;
    std::map<int, mfa::DeviceState>   dev;
    uint32_t                          last_kernel = 0;   // MFA_KERNEL_*
};

namespace mfa {

// launchers (kernels.hip); all asynchronous on `stream`
int launch_mfa_walk(const HostImage& img, DeviceState& ds, const uint8_t* d_bytes, const uint64_t* d_offsets,
                    uint64_t n, uint8_t* d_results, void* stream);
int launch_dfa_walk(const HostImage& img, DeviceState& ds, const uint8_t* d_bytes, const uint64_t* d_offsets,
                    uint64_t n, uint8_t* d_results, void* stream);
int device_prepare(mfa_image* img, int device, DeviceState** out);
// specialised kernels (jit_gen.cpp, jit.hip)
uint32_t    jit_slot_registers(const HostImage& img);
uint32_t    jit_lanes(const HostImage& img);
std::string jit_generate_source(const HostImage& img);
bool        jit_enabled(const HostImage& img);
std::string jit_compile(const HostImage& img, std::string* err);
bool        jit_load(const HostImage& img, DeviceState& ds);
void        jit_unload(DeviceState& ds);
void        jit_print_stats(DeviceState& ds, const char* tag);
int         launch_mfa_jit(DeviceState& ds, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_results,
                           void* stream);
void device_release(DeviceState& ds);
void set_last_hip_error(int e);

}  // namespace mfa

#endif
