// Launch interface of the live-list walk kernel (walk.hip, one object per cell count K).
#ifndef MFA_WALK_H
#define MFA_WALK_H

#include <cstddef>
#include <cstdint>

#ifndef WALK_MIN_WAVES
#define WALK_MIN_WAVES 2      /* __launch_bounds__: waves per SIMD the register allocation leaves room for */
#endif

namespace mfa {

constexpr uint32_t WALK_MAX_SEG = 16;      // automata per launch

struct WalkArgs {                     // kernel parameters; every pointer is a device pointer
    const uint8_t*  bytes;
    const uint64_t* offsets;
    uint64_t        n;
    uint8_t*        results;
    const uint64_t* regions;          // region table of the batch, or nullptr
    const uint32_t* tables;           // table blocks of the launch's automata, back to back (walk_tables.h)
    uint32_t*       spill;            // per wave: list entries and probe images beyond the LDS capacity
    unsigned long long* counter;      // ticket counter, zeroed before the launch
    uint32_t table_words, shared_words;      // LDS words: the tables, and the tables rounded up to a multiple of 64
    uint32_t n_seg, C, CX, accel, refill;
    uint32_t images_global;           // the two probe images of a lane live in global memory (less LDS per wave: more waves per CU)
    uint32_t gate;                    // 0, or 0x1000 | stamp: the region table is being written while the kernel runs, rows carry this stamp (regions.hip: GATE)
    uint32_t nm_words;                // long-list kernel (WALK_NODE_MAP): words of a lane's node map = (nodes of the launch's largest automaton + 3) / 4, else 0
    uint32_t* lean_queue;             // strings without a periodic stretch are handed to walk_lean_kernel through this queue (nullptr: all are walked here);
                                      //   its length is counter[1] (as 32 bits), the lean kernel's ticket counter counter[2]
    uint32_t* lean_seen;              // pinned host word (or nullptr): the lean kernel stores the queue's length + 1 there (mfa_internal.h: LeanHint)
    uint32_t seg_first[WALK_MAX_SEG + 1];    // segment s = strings seg_first[s] .. seg_first[s+1]-1 of this launch ...
    uint32_t seg_table[WALK_MAX_SEG];        // ... walks the automaton whose table block starts at this word of `tables`
};

struct WalkLaunch {
    WalkArgs args;
    unsigned grid;                    // workgroups of 256 threads
    unsigned lean_grid;               // ... of the lean kernel that follows (0: none)
    uint32_t lean_C;                  // its list capacity in LDS
    bool     reversed;
    bool     tables_global;           // the tables do not fit LDS: the kernel reads them from global memory (shared_words = 0)
};

#define MFA_WALK_DECL(K) int launch_walk_k##K(const WalkLaunch& L, void* stream); size_t walk_wave_words_k##K(uint32_t C, bool images_global);
MFA_WALK_DECL(1) MFA_WALK_DECL(2) MFA_WALK_DECL(3) MFA_WALK_DECL(4) MFA_WALK_DECL(5) MFA_WALK_DECL(6) MFA_WALK_DECL(7) MFA_WALK_DECL(8) MFA_WALK_DECL(9)
#undef MFA_WALK_DECL
int launch_walk_long_k1(const WalkLaunch& L, void* stream);   // K = 1, automata of 17-128 nodes whose lists are long (77-node ex. 8 -bnf / -reverse): a node's entry is found through a per-lane map in LDS (WALK_NODE_MAP)
int launch_walk_stats(const WalkLaunch& L, void* stream);      // K = 1 with counters (MFA_WALK_STATS=1; development)
void walk_print_stats(unsigned long long* d_counter, const char* tag);

}  // namespace mfa

#endif
