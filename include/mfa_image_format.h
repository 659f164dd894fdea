/* Automaton image: the flat, pointer-free form of one compiled automaton.
 *
 * It freezes exactly what the reference's match loop reads from its heap graph
 * (reference: automata.h:18-84 `Automata`/`MFA` public fields, node.h:11-38,
 * edge.h:13-49): the node list, per node the ordered out-edge list, per edge the
 * label, the target and the per-cell memory actions, plus start/finish and
 * is_reversed.  Nodes are numbered by their ORDER RANK: the reference orders its
 * state sets by raw node pointers (automata.h:12-13, automata.cpp:121-123), which
 * under the canonical allocation-order model is allocation
 * order; node k of an image is the k-th allocated node.  Edges of a node keep the
 * reference's std::list order.
 *
 * All fields little-endian.  Layout of a blob:
 *     mfa_blob_header                      40 bytes
 *     uint32_t edge_begin[n_nodes + 1]     CSR offsets into the edge array
 *     mfa_blob_edge   edges[n_edges]       8 bytes each
 */
#ifndef MFA_IMAGE_FORMAT_H
#define MFA_IMAGE_FORMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_BLOB_MAGIC   0x4941464Du /* "MFAI" */
#define MFA_BLOB_VERSION 1u

#define MFA_KIND_NFA 0u /* memory-less automaton, walked by Automata::match (automata.cpp:177) */
#define MFA_KIND_MFA 1u /* memory automaton, walked by MFA::match (mfa.cpp:215)               */

#define MFA_MAX_CELLS 9u /* cells are named "1".."9" (README.md:21, mfa.cpp:148) */

#define MFA_EDGE_EPS 1u /* flags bit 0: epsilon edge (`by` empty, or the UTF-8 epsilon mfa.cpp:40-42 writes) */

/* per-cell action codes inside mfa_blob_edge.actions, 2 bits per cell c at bit 2*c (c = 1..9) */
#define MFA_ACT_NONE  0u
#define MFA_ACT_OPEN  1u /* MemoryAction::open  (edge.h:29-32) */
#define MFA_ACT_CLOSE 2u /* MemoryAction::close */

typedef struct mfa_blob_header {
    uint32_t magic;
    uint32_t version;
    uint32_t kind;        /* MFA_KIND_* */
    uint32_t is_reversed; /* Automata::is_reversed / MFA::is_reversed (automata.h:24,55) */
    uint32_t n_nodes;
    uint32_t n_edges;
    uint32_t start;
    uint32_t finish;
    uint32_t n_cells;     /* highest cell number that appears as a label or in an action (0..9) */
    uint32_t reserved;
} mfa_blob_header;

typedef struct mfa_blob_edge {
    uint8_t  label;   /* the single byte of Edge::by (edge.h:15); '.' = any byte; '1'..'9' = read cell */
    uint8_t  flags;   /* MFA_EDGE_EPS */
    uint16_t target;  /* node number of Edge::to */
    uint32_t actions; /* MemoryEdge::memoryActions (edge.h:36), 2 bits per cell */
} mfa_blob_edge;

#define MFA_EDGE_ACTION(e, cell) (((e).actions >> (2u * (cell))) & 3u)

#ifdef __cplusplus
}
#endif
#endif /* MFA_IMAGE_FORMAT_H */
