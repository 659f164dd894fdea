/* C-ABI of the MI355X match path (libmfa_hip.so).
 *
 * This is the drop-in boundary for the reference's match loop: everything the
 * reference does between "automaton compiled" and "0/1 per string" --
 *     bool MFA::match(string)              reference automata.h:69, mfa.cpp:215-236
 *     bool Automata::match(const string&)  reference automata.h:42, automata.cpp:177-210
 *     the per-string loops that call them  reference matchers/match.cpp:21-31,
 *                                          matchers/match_mfa.cpp:28-36,72-80
 * -- for a whole batch of strings at once, on the GPU.  The reference has no FFI
 * of its own (it is one C++ process); these are the entry points its
 * `Automata`/`MFA` classes bind when the match loop is moved to the device (the
 * binding is shown in INTEGRATION.md, our own host mirror of those classes lives in
 * re2-modification_amd/host/).
 *
 * Plain C types only: pointers and sizes.  No exceptions cross this boundary;
 * every function returns 0 (MFA_OK) or a negative MFA_ERR_* code.  The caller owns
 * every buffer it passes.  Functions are re-entrant per (image, device, stream):
 * every launch takes its own workspace (ticket counter, scratch, region table,
 * events) from a per-image pool, so one image may be used from several host threads
 * and on several streams at the same time; launches on one stream run in stream order.
 *
 * There is NO CPU fallback: without a usable HIP device the match entry points
 * return MFA_ERR_NO_DEVICE.
 */
#ifndef MFA_HIP_H
#define MFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "mfa_image_format.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_OK                 0
#define MFA_ERR_INVALID_ARG   -1  /* NULL pointer, bad size                                            */
#define MFA_ERR_BAD_BLOB      -2  /* blob fails the format checks of mfa_image_format.h                */
#define MFA_ERR_UNSUPPORTED   -3  /* well-formed automaton outside the kernels' limits (see below)     */
#define MFA_ERR_NO_DEVICE     -4  /* no HIP device / device index out of range                         */
#define MFA_ERR_HIP           -5  /* a HIP runtime call failed; mfa_last_hip_error() has the code       */
#define MFA_ERR_NOMEM         -6
#define MFA_ERR_TOO_LONG      -7  /* a string is longer than MFA_MAX_STRING_BYTES                       */
#define MFA_ERR_JIT           -8  /* compiling a specialised kernel failed (the table-driven walk still works) */

/* limits of the device kernels (violations -> MFA_ERR_UNSUPPORTED at image creation) */
#define MFA_MAX_NODES        1024u     /* MFA kind: nodes (any out-degree)                 */
#define MFA_MAX_KERNEL_CELLS 9u        /* MFA kind: distinct memory cells ("1".."9", mfa.cpp:148) */
#define MFA_MAX_DFA_STATES   (1u << 20) /* NFA kind: reachable state sets after tabulation  */
#define MFA_MAX_STRING_BYTES 0x00ffffffu /* 16 MiB - 1 per string                          */

typedef struct mfa_image mfa_image_t;

typedef struct mfa_image_info {
    uint32_t kind;         /* MFA_KIND_NFA / MFA_KIND_MFA                                   */
    uint32_t is_reversed;
    uint32_t n_nodes, n_edges, n_cells;
    uint32_t dfa_states;   /* NFA kind: number of tabulated state sets (incl. the dead set) */
    uint32_t byte_classes; /* NFA kind: number of input byte classes                        */
    uint32_t last_kernel;  /* MFA_KERNEL_*: which kernel the last match call on this image launched */
} mfa_image_info;

#define MFA_KERNEL_NONE        0u
#define MFA_KERNEL_WALK        1u /* walk_kernel: table-driven memory-automaton walk over a list of live states (any automaton, no compiler) */
#define MFA_KERNEL_SPECIALISED 2u /* mfa_jit_kernel: the walk generated for one automaton, one slot per node in VGPRs */
#define MFA_KERNEL_TABLE       3u /* dfa_walk_kernel: tabulated memory-less automaton                    */

/* Build an image from a blob (include/mfa_image_format.h).  Host-only work: parse,
 * check the structural invariants the kernels rely on, and for MFA_KIND_NFA tabulate
 * the reference's step function (automata.cpp:98-128) into a transition table.
 * Needs no GPU.  Replaces: holding an `Automata*` / `MFA*` (automata.h:18-84). */
int  mfa_image_create(const void* blob, size_t n_bytes, mfa_image_t** out);
void mfa_image_destroy(mfa_image_t* img);
int  mfa_image_get_info(const mfa_image_t* img, mfa_image_info* out);

/* Upload the image's tables to `device` and allocate its launch workspace now
 * (otherwise done by the first match call on that device). */
int  mfa_image_prepare(mfa_image_t* img, int device);

/* Every memory automaton is walked by the table-driven kernel (MFA_KERNEL_WALK) as soon as its image
 * exists: nothing is compiled per automaton.  A small automaton (up to 128 nodes) can ALSO be given a
 * kernel specialised to it (straight-line code, its state in registers: faster per input character on
 * text without periodic stretches): generated as HIP source, compiled for gfx950 with hipcc and cached
 * as a code object next to the library (or in $MFA_JIT_CACHE).  This call does that now; it is host-only
 * work and needs no GPU, so caches can be built ahead of time.  A match call uses the specialised kernel
 * when its code object is in the cache and never waits for a compiler (MFA_WALK=table / MFA_WALK=jit force
 * one kernel or the other; MFA_JIT=0 disables specialised kernels; MFA_ACCEL=0 or MFA_REGIONS=0: no region pass,
 * every step is executed -- A/B runs).  MFA_ERR_UNSUPPORTED: the automaton
 * is too large for a specialised kernel; MFA_ERR_JIT: the compiler failed. */
int  mfa_image_specialize(mfa_image_t* img);

/* Match n strings; string k is bytes[offsets[k] .. offsets[k+1]).  ALL pointers are
 * DEVICE pointers on `device` (offsets has n+1 entries; results gets n bytes, 1 =
 * accepted, 0 = rejected -- the value `cout << match` prints, match.cpp:30).
 * A string longer than MFA_MAX_STRING_BYTES is not matched: its result byte is set to 2.
 * The kernels read the batch in whole 16-byte blocks: d_bytes must be readable up to offsets[n]
 * rounded up to the next multiple of 16 (hipMalloc'ed buffers always are; a batch carved out of a
 * larger buffer needs up to 15 bytes of slack behind it).  The bytes there are never interpreted.
 * Asynchronous: work is enqueued on `stream` (a hipStream_t, NULL = default
 * stream) and the call returns.  Replaces: the loop
 *     while (...) { match = automata->match(text); }      match.cpp:21-31 */
int  mfa_match_batch(mfa_image_t* img, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                     uint8_t* d_results, int device, void* stream);

/* ---- region tables -------------------------------------------------------------------------------
 * Before a memory automaton walks a batch, one streaming pass over the batch (region_scan_kernel)
 * finds every string's periodic regions -- stretches with s[j] == s[j+q], q <= 8 -- and leaves them in
 * a table the walk kernels look up instead of measuring the stretches themselves.  mfa_match_batch
 * runs that pass itself; it is exported for callers that match SEVERAL automata against the same
 * batch (mfa_match_batch_regions: the pass then runs once) and for tests.
 * Table layout: MFA_REGION_WORDS uint64 per string; word 0 = count | flags, then `count` entries
 *   lo | hi << 24 | q << 48     (offsets relative to the start of the string, memory order)
 * Every entry is true (s[j] == s[j+q] for lo <= j < hi - q) and entries with q = 1 are maximal runs.
 * MFA_REGION_OVERFLOW in word 0: the string has more regions than fit and the table holds the first three and the longest of the others
 * (or the string is too long to be matched): the walk then executes the other stretches step by step. */
#define MFA_REGION_WORDS    16u
#define MFA_REGION_MAX      15u
#define MFA_REGION_OVERFLOW 0x100ull
#define MFA_REGION_MIN_LEN  64u

/* d_table: device buffer of n * MFA_REGION_WORDS uint64.  Asynchronous on `stream`. */
int  mfa_region_scan(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t* d_table,
                     int device, void* stream);

/* mfa_match_batch with a region table the caller has already filled for this batch on this stream
 * (or on a stream this one waits for).  d_table == NULL: no table, every step is executed. */
int  mfa_match_batch_regions(mfa_image_t* img, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                             uint8_t* d_results, const uint64_t* d_table, int device, void* stream);

/* ---- mixed batches --------------------------------------------------------------------------------
 * ONE batch whose strings belong to several memory automata, segment by segment (the 10-example attack
 * corpus is one byte buffer, one offset array and ten segments).  Replaces: running the loop of
 * match.cpp:21-31 once per automaton.  A mixed object holds the automata's tables back to back, per
 * device; the images must outlive it, scan in the same direction and fit the table-driven walk
 * (MFA_ERR_UNSUPPORTED otherwise).  One call runs the region pass over the batch in a few groups of
 * consecutive segments and walks each group -- all its automata in ONE launch, any lane any automaton --
 * as soon as its regions are known: the region launches go to `stream` itself, the walks to internal streams, so that
 * the walk of a group runs beside the region pass of the next (one region launch per group, the walks wait for the event
 * behind it).  Calls on one object are ordered one behind the other, also when they come on different streams (the object's
 * table and work areas are shared).  (MFA_MIXED_GATE=1, table-driven
 * walk only: the region pass is ONE launch over the whole batch; it counts every finished string for its group, a
 * one-wave kernel in front of a group's walk launches ends when the group is complete, and every word of a table row
 * carries the call's stamp, so that a walk never takes a row that has not arrived yet -- or a stale copy of an earlier
 * call's -- for this call's.  Measured: not faster; kept as an option.)  `stream` sees the call as a single operation: it waits for the internal streams before the call
 * returns, ALSO when the call returns an error (whatever was started is ordered before the caller's next
 * operation on `stream`). */
typedef struct mfa_mixed mfa_mixed_t;
int  mfa_mixed_create(mfa_image_t* const* images, uint32_t n_images, mfa_mixed_t** out);
void mfa_mixed_destroy(mfa_mixed_t* mx);
/* seg_first: HOST array of n_images + 1 string indices, seg_first[0] = 0, seg_first[n_images] = n: strings
 * seg_first[s] .. seg_first[s+1]-1 are matched against images[s].  Device pointers as in mfa_match_batch.
 * Asynchronous on `stream` -- except that the first call with a string count this object has not met (n >= 65536) reads the
 * batch's size in bytes back (offsets[n] - offsets[0]) to choose the number of groups, and waits for `stream` to do so
 * (make such a call outside a stream capture, or use mfa_match_mixed_sized), and that with the generated kernels (MFA_WALK=jit)
 * the first call on a device times the walks and synchronises on its own end.  A later batch with the same string count and
 * other bytes is grouped like the first: a matter of speed only. */
int  mfa_match_mixed(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                     const uint64_t* seg_first, uint8_t* d_results, int device, void* stream);
/* the same for a caller that knows the batch's size in bytes (total_bytes = offsets[n] - offsets[0], > 0): nothing is read back,
 * the call never waits for `stream` */
int  mfa_match_mixed_sized(mfa_mixed_t* mx, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes,
                           const uint64_t* seg_first, uint8_t* d_results, int device, void* stream);
/* the same with HOST pointers (copy in, match, copy out, synchronise): for callers that hold std::strings */
int  mfa_match_mixed_host(mfa_mixed_t* mx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n,
                          const uint64_t* seg_first, uint8_t* results, int device);
/* device time of the last mfa_match_mixed call on `device`: its region launches, and first region launch to
 * last walk (either pointer may be NULL).  Synchronises on the call's last events. */
int  mfa_mixed_last_ms(mfa_mixed_t* mx, int device, float* region_ms, float* span_ms);
/* the same for the call `back` calls ago (0 = the last one; the events of the last 32 calls are kept, so a sequence of calls can be
 * timed without synchronising between them) */
int  mfa_mixed_timing(mfa_mixed_t* mx, int device, uint32_t back, float* region_ms, float* span_ms);
/* what the last call on `device` launched (any pointer may be NULL): region launches (1 when the walks are released by counters),
 * walk launches, groups of strings, and gated = 1 if the walks were released by counters, 0 if by events */
int  mfa_mixed_last_launches(mfa_mixed_t* mx, int device, uint32_t* region_launches, uint32_t* walk_launches, uint32_t* groups, uint32_t* gated);

/* The result vector of a batch as a bitmap: bit k % 8 of byte k / 8 of d_bitmap ((n + 7) / 8 bytes, device memory) = string k was accepted
 * (result code 1).  Asynchronous on `stream`.  What a process sends when the results of a batch sharded over several GPUs are gathered
 * (the reference matches one string at a time and has no such step: matchers/match_mfa.cpp:28-36 prints each answer as it comes). */
int  mfa_pack_result_bitmap(const uint8_t* d_results, uint64_t n, uint8_t* d_bitmap, void* stream);

/* Same with HOST pointers: copies the batch to the device, matches, copies the
 * results back, synchronises.  Convenience for callers that hold std::strings
 * (the CLI); throughput is then bounded by the host link, not by the kernel. */
int  mfa_match_batch_host(mfa_image_t* img, const uint8_t* bytes, const uint64_t* offsets, uint64_t n,
                          uint8_t* results, int device);

/* Device-side time of the last match kernel launched through this image on
 * `device`, in milliseconds, measured with HIP events recorded on the launch stream
 * around the kernel alone.  Synchronises on the stop event. */
int  mfa_last_kernel_ms(mfa_image_t* img, int device, float* ms);
/* Device-side time of the region pass of that launch (0 if it ran none). */
int  mfa_last_region_ms(mfa_image_t* img, int device, float* ms);

int         mfa_device_count(void);        /* >= 0, or MFA_ERR_NO_DEVICE */
int         mfa_last_hip_error(void);      /* hipError_t of the last failed HIP call on this thread */
const char* mfa_strerror(int code);
const char* mfa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MFA_HIP_H */
