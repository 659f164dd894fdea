/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's match path.
 * Nothing in the product (re2-modification_amd/, include/) may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline do,
 * and only as the checker.
 *
 * Restates, on a flat automaton image (include/mfa_image_format.h):
 *   MFA::match / evaluateStates / evaluateState / doMemoryWriteActions /
 *   copy_memory / is_siffix_long_enough          (reference mfa.cpp:80-236)
 *   Automata::match / evaluateStates / evaluateState (reference automata.cpp:98-210)
 * under the canonical allocation-order model of SURVEY.md section 0.4: every pointer
 * comparison the reference makes (std::set<MemoryState>, automata.h:12-13) is an
 * allocation-sequence comparison.  Pinned against the reference itself
 * (oracle/_ref/ref_harness, same model) by tests/golden/ -- see tests/golden/README.md.
 */
#ifndef MFA_ORACLE_H
#define MFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfa_oracle_image mfa_oracle_image;

typedef struct mfa_oracle_stats {
    uint64_t steps;          /* evaluateStates calls (incl. the final pass)           */
    uint64_t evaluations;    /* top-level evaluateState calls (winners)                */
    uint64_t cell_reads;     /* cell-read attempts (mfa.cpp:177)                       */
    uint64_t compare_bytes;  /* sum of |value| over reads that reached the compare     */
    uint64_t max_states;     /* largest |states| seen on entry to a step               */
    uint64_t variables;      /* Variables allocated                                    */
} mfa_oracle_stats;

/* returns 0 or a negative error */
int  mfa_oracle_image_load(const void* blob, size_t n_bytes, mfa_oracle_image** out);
void mfa_oracle_image_free(mfa_oracle_image* img);

/* 1 = match, 0 = no match, <0 = error.  `stats` may be NULL (accumulated, not reset). */
int  mfa_oracle_match(const mfa_oracle_image* img, const uint8_t* str, uint64_t len, mfa_oracle_stats* stats);

/* offsets has n+1 entries; results gets n bytes of 0/1.  returns 0 or a negative error */
int  mfa_oracle_match_batch(const mfa_oracle_image* img, const uint8_t* bytes, const uint64_t* offsets,
                            uint64_t n, uint8_t* results, mfa_oracle_stats* stats);

#ifdef __cplusplus
}
#endif
/* 0 (default): the reference's rule, the first state per node in set order wins (mfa.cpp:206-211).  1: among the states that
 * tie on (pos, node) the last one wins -- only used to classify fixtures as tie-sensitive or not. */
void mfa_oracle_set_tie_policy(int last_wins);

#endif
