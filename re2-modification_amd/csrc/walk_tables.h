// Tables of the live-list walk kernel (walk.hip / walk_core.h), built on the host from an automaton image.
//
// The walk keeps, per string, a short list of live states (one per automaton node at most: mfa.cpp:206-211 evaluates the
// first state per node and drops the rest) and interprets the automaton from tables.  What mfa.cpp decides at run time from
// "is cell c in this state's memory" (mfa.cpp:148-160: a digit edge whose cell is absent creates the cell and recurses into
// the target, consuming nothing) is decided HERE, once per automaton: a state's node is refined to a VNODE = (node, set of
// cells present), and for every vnode the recursion through absent-cell edges is flattened, in the reference's evaluation
// order, into
//   * one list of effective edges per input byte class   (states with pos == i: mfa.cpp:161-194)
//   * one list of "waiting" insertions                   (states with pos > i: mfa.cpp:148-160 + 195-197)
//   * flags: an epsilon edge accepts at pos == len, directly or through the recursion (mfa.cpp:138-147)
// so the device code has no recursion, no stack and no presence tests.
#ifndef MFA_WALK_TABLES_H
#define MFA_WALK_TABLES_H

#include <cstdint>
#include <vector>

namespace mfa {

struct HostImage;

// ---- device format (all u32 words; offsets are word offsets from the start of the image's table block) ----------------
// header (WT_HDR words):
//   [0] n_vids = n_nodes << vb   [1] vb (variant bits)   [2] n_classes   [3] K (cells, >= 1)   [4] start vid
//   [5] off_cmap  (64 words: byte -> class, 4 per word)
//   [6] off_vinfo (n_vids words)   [7] off_vc (n_vids words)   [8] off_vb (n_vids * n_classes words)
//   [9] off_ee   [10] n_ee   [11] eew (words per effective edge: 2, or 3 for launches with more than 6 cells)   [12] total words   [13] reversed
//   [14] n_nodes   [15] reserved
// vid = node << vb | variant.  Unused (node, variant) slots have vinfo = 0 and are never reached.
// vinfo: bit 0 valid | bit 1 has_eps (A: accept when pos == len) | bit 2 c_acc (an epsilon edge behind absent-cell edges)
//        | bit 3 qualifies (a waiting state is carried: mfa.cpp:195-197) | bits 4..7 fname (lowest present cell, 0 = none)
//        | bits 8..16 mask (cells present, bit c-1) | bit 17 the vnode has waiting insertions (vc)
// vc / vb: begin << 12 | count   (begin in effective edges, count < 4096)
// effective edge, word 0: bit 0 kind (0 = literal / waiting insertion, 1 = cell read) | bits 1..4 cell (0-based) of a read
//        | bits 5..8 fname of the target vnode | bits 9..31 target vid
//   word 1 (2-word edges, up to 6 cells): actions (2 bits per cell c at bit 2(c-1): 1 open, 2 close) | cm << 12 | com << 18 | rdm << 24
//   words 1, 2 (3-word edges): word 1 = actions | cm << 18, word 2 = com | rdm << 9
//   (the field positions do not depend on the automaton's own cell count: a launch may walk automata with fewer cells than its kernel has)
//   cm  = cells created (empty, at the state's pos) on the way to this edge, com = those of them created open,
//   rdm = cells whose is_read flag earlier read attempts of the same evaluation have set (mfa.cpp:177; the copy of
//         mfa.cpp:167 is taken before the edge's own read)
constexpr uint32_t WT_HDR = 16;
constexpr uint32_t WT_MAX_NODES = 1024;
constexpr uint32_t WT_MAX_VIDS = 1u << 13;

struct WalkTables {
    std::vector<uint32_t> words;      // header + tables
    uint32_t n_vnodes = 0;            // reachable (node, mask) pairs
    uint32_t max_live = 0;            // upper bound of live states per string: nodes other than finish
    uint32_t K = 1;
    bool reversed = false;
};

// MFA_OK, or MFA_ERR_UNSUPPORTED when the automaton exceeds the table format (nodes, variants per node, list lengths)
// wide: 3-word effective edges (the image is walked by a kernel for more than 6 cells)
int build_walk_tables(const HostImage& img, WalkTables& out, bool wide = false);

}  // namespace mfa

#endif
