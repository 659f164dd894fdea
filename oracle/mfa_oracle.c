/* TEST INFRASTRUCTURE ONLY -- see mfa_oracle.h.
 *
 * A deliberately literal restatement: full state sets, the reference's own
 * ordering rules, Variables as shared heap objects -- no winner pruning, no
 * laziness.  The product kernels use a different (slot-per-node) formulation;
 * the tests prove the two agree.
 *
 * Model of the reference's containers:
 *   Variable*            -> index into a per-match pool; the index IS the allocation
 *                           sequence number, so "pointer <" == "index <".
 *   Memory (map<string,Variable*>, automata.h:12) -> cell[1..9], -1 = absent
 *   MemoryState (pair<int,pair<MemoryNode*,Memory>>, automata.h:13) -> mstate_t
 *   set<MemoryState>     -> array kept sorted by mstate_cmp (the pair/map operator<)
 *   MemoryNode*          -> node number (images number nodes by pointer rank)
 *   value (std::string)  -> span [start,start+len) of the SCAN-ORDER input: every
 *                           write appends exactly the text just consumed
 *                           (mfa.cpp:89-104), so values are contiguous.
 */
#include "mfa_oracle.h"
#include "../include/mfa_image_format.h"

#include <stdlib.h>
#include <string.h>

struct mfa_oracle_image {
    mfa_blob_header h;
    uint32_t*       edge_begin;
    mfa_blob_edge*  edges;
};

int mfa_oracle_image_load(const void* blob, size_t n_bytes, mfa_oracle_image** out) {
    if (!blob || !out || n_bytes < sizeof(mfa_blob_header)) return -1;
    mfa_blob_header h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != MFA_BLOB_MAGIC || h.version != MFA_BLOB_VERSION) return -2;
    if (h.kind > 1 || h.n_nodes == 0 || h.start >= h.n_nodes || h.finish >= h.n_nodes || h.n_cells > 9) return -3;
    size_t need = sizeof h + (size_t)(h.n_nodes + 1) * 4 + (size_t)h.n_edges * sizeof(mfa_blob_edge);
    if (n_bytes < need) return -4;
    mfa_oracle_image* img = (mfa_oracle_image*)calloc(1, sizeof *img);
    if (!img) return -5;
    img->h = h;
    img->edge_begin = (uint32_t*)malloc((size_t)(h.n_nodes + 1) * 4);
    img->edges = (mfa_blob_edge*)malloc((size_t)(h.n_edges ? h.n_edges : 1) * sizeof(mfa_blob_edge));
    memcpy(img->edge_begin, (const char*)blob + sizeof h, (size_t)(h.n_nodes + 1) * 4);
    memcpy(img->edges, (const char*)blob + sizeof h + (size_t)(h.n_nodes + 1) * 4,
           (size_t)h.n_edges * sizeof(mfa_blob_edge));
    for (uint32_t k = 0; k < h.n_nodes; k++)
        if (img->edge_begin[k] > img->edge_begin[k + 1] || img->edge_begin[k + 1] > h.n_edges) {
            mfa_oracle_image_free(img); return -6;
        }
    for (uint32_t e = 0; e < h.n_edges; e++)
        if (img->edges[e].target >= h.n_nodes) { mfa_oracle_image_free(img); return -7; }
    *out = img;
    return 0;
}

void mfa_oracle_image_free(mfa_oracle_image* img) {
    if (!img) return;
    free(img->edge_begin); free(img->edges); free(img);
}

/* ------------------------------------------------------------------ MFA --- */

typedef struct { uint8_t is_open, is_read; uint64_t start, len; } var_t;   /* variable.h:8-41 */
typedef struct { int64_t pos; int32_t node; int32_t cell[10]; } mstate_t;
typedef struct { mstate_t* v; size_t n, cap; } sset_t;

typedef struct {
    const mfa_oracle_image* img;
    const uint8_t* str; int64_t len; int reversed;
    var_t* pool; size_t npool, cappool;
    mfa_oracle_stats* st;
} mctx_t;

static inline uint8_t scan_at(const mctx_t* c, int64_t j) {       /* mfa.cpp:163-166 */
    return c->reversed ? c->str[c->len - 1 - j] : c->str[j];
}

static int32_t var_new(mctx_t* c, int is_open, int is_read, uint64_t start, uint64_t len) {
    if (c->npool == c->cappool) {
        c->cappool = c->cappool ? c->cappool * 2 : 1024;
        c->pool = (var_t*)realloc(c->pool, c->cappool * sizeof(var_t));
        if (!c->pool) abort();
    }
    var_t* v = &c->pool[c->npool];
    v->is_open = (uint8_t)is_open; v->is_read = (uint8_t)is_read; v->start = start; v->len = len;
    if (c->st) c->st->variables++;
    return (int32_t)c->npool++;
}

/* operator< of map<string,Variable*>: lexicographic over (name, pointer) pairs */
static int mem_cmp(const int32_t* a, const int32_t* b) {
    int ca = 1, cb = 1;
    for (;;) {
        while (ca <= 9 && a[ca] < 0) ca++;
        while (cb <= 9 && b[cb] < 0) cb++;
        if (ca > 9 && cb > 9) return 0;
        if (ca > 9) return -1;
        if (cb > 9) return 1;
        if (ca != cb) return ca < cb ? -1 : 1;
        if (a[ca] != b[cb]) return a[ca] < b[cb] ? -1 : 1;
        ca++; cb++;
    }
}

static int mstate_cmp(const mstate_t* a, const mstate_t* b) {    /* automata.h:13 */
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    if (a->node != b->node) return a->node < b->node ? -1 : 1;
    return mem_cmp(a->cell, b->cell);
}

static void sset_insert(sset_t* s, const mstate_t* x) {
    size_t k = 0;
    while (k < s->n) {
        int c = mstate_cmp(&s->v[k], x);
        if (c == 0) return;
        if (c > 0) break;
        k++;
    }
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 16;
        s->v = (mstate_t*)realloc(s->v, s->cap * sizeof(mstate_t));
        if (!s->v) abort();
    }
    memmove(&s->v[k + 1], &s->v[k], (s->n - k) * sizeof(mstate_t));
    s->v[k] = *x;
    s->n++;
}

/* copy_memory, mfa.cpp:107-114: a fresh Variable per cell, in name order */
static void copy_memory(mctx_t* c, mstate_t* s) {
    for (int k = 1; k <= 9; k++)
        if (s->cell[k] >= 0) {
            var_t v = c->pool[s->cell[k]];
            s->cell[k] = var_new(c, v.is_open, v.is_read, v.start, v.len);
        }
}

/* MFA::doMemoryWriteActions, mfa.cpp:80-105; the text is scan[tstart, tstart+tlen) */
static void do_actions(mctx_t* c, const mfa_blob_edge* e, mstate_t* s, uint64_t tstart, uint64_t tlen) {
    for (unsigned k = 1; k <= 9; k++)                                /* mfa.cpp:82-86 */
        if (MFA_EDGE_ACTION(*e, k) == MFA_ACT_OPEN && s->cell[k] < 0)
            s->cell[k] = var_new(c, 0, 0, tstart, 0);
    for (unsigned k = 1; k <= 9; k++) {                              /* mfa.cpp:89-104 */
        if (s->cell[k] < 0) continue;
        var_t* v = &c->pool[s->cell[k]];
        unsigned act = MFA_EDGE_ACTION(*e, k);
        if (act == MFA_ACT_OPEN) {
            v->is_open = 1; v->is_read = 0; v->start = tstart; v->len = tlen;   /* open(); write(t) */
        } else if (act == MFA_ACT_CLOSE) {
            v->is_open = 0;                                                      /* close() */
        } else if (v->is_open) {
            if (v->len == 0) v->start = tstart;
            if (v->start + v->len != tstart) abort();    /* contiguity invariant of the span model */
            v->len += tlen;                                                      /* write(t) */
        }
    }
}

/* MFA::is_siffix_long_enough, mfa.cpp:116-133 */
static int suffix_ok(const mctx_t* c, const mstate_t* s, int64_t i) {
    if (!c->reversed) return 1;
    int64_t suffix = c->len - i, needed = 0;
    for (int k = 1; k <= 9; k++)
        if (s->cell[k] >= 0) {
            const var_t* v = &c->pool[s->cell[k]];
            if (v->is_open || !v->is_read) needed += (int64_t)v->len;
        }
    return needed <= suffix;
}

/* MFA::evaluateState, mfa.cpp:136-200 */
static void m_eval_state(mctx_t* c, mstate_t st, int64_t i, sset_t* out) {
    const mfa_oracle_image* g = c->img;
    if ((uint32_t)st.node == g->h.finish && st.pos == c->len) { sset_insert(out, &st); return; }
    if (!suffix_ok(c, &st, i)) return;
    for (uint32_t ei = g->edge_begin[st.node]; ei < g->edge_begin[st.node + 1]; ei++) {
        const mfa_blob_edge* e = &g->edges[ei];
        int digit = (!(e->flags & MFA_EDGE_EPS) && e->label >= '1' && e->label <= '9') ? e->label - '0' : 0;
        if (e->flags & MFA_EDGE_EPS) {                                           /* mfa.cpp:143-147 */
            mstate_t ns = st; ns.node = e->target; copy_memory(c, &ns);
            m_eval_state(c, ns, i, out);
        } else if (digit && st.cell[digit] < 0) {                                /* mfa.cpp:148-160 */
            mstate_t ns = st; ns.node = e->target; copy_memory(c, &ns);
            unsigned act = MFA_EDGE_ACTION(*e, (unsigned)digit);
            ns.cell[digit] = var_new(c, act == MFA_ACT_OPEN, 0, (uint64_t)st.pos, 0);
            m_eval_state(c, ns, i, out);
        } else if (i != c->len && i == st.pos) {                                 /* mfa.cpp:161-194 */
            uint8_t ch = scan_at(c, i);
            mstate_t ns = st; ns.node = e->target; copy_memory(c, &ns);          /* copy BEFORE any read() */
            if (e->label == '.' || e->label == ch) {
                do_actions(c, e, &ns, (uint64_t)i, 1);
                ns.pos += 1;
                sset_insert(out, &ns);
            } else if (digit && st.cell[digit] >= 0) {
                var_t* v = &c->pool[st.cell[digit]];
                v->is_read = 1;                                  /* read() marks the SOURCE's Variable, mfa.cpp:177 */
                uint64_t l = v->len;
                if (c->st) { c->st->cell_reads++; }
                if ((uint64_t)(c->len - i) >= l) {
                    if (c->st) c->st->compare_bytes += l;
                    int eq = 1;
                    for (uint64_t k = 0; k < l; k++)
                        if (scan_at(c, (int64_t)(v->start + k)) != scan_at(c, i + (int64_t)k)) { eq = 0; break; }
                    if (eq) {
                        ns.pos += (int64_t)l;
                        do_actions(c, e, &ns, (uint64_t)i, l);
                        sset_insert(out, &ns);
                    }
                }
            }
        } else if (i != c->len && i < st.pos) {                                  /* mfa.cpp:195-197 */
            sset_insert(out, &st);
        }
    }
}

static int g_tie_last = 0;
void mfa_oracle_set_tie_policy(int last_wins) { g_tie_last = last_wins; }

static int match_mfa(const mfa_oracle_image* img, const uint8_t* str, int64_t len, mfa_oracle_stats* stats) {
    mctx_t c; memset(&c, 0, sizeof c);
    c.img = img; c.str = str; c.len = len; c.reversed = (int)img->h.is_reversed; c.st = stats;
    sset_t cur = {0, 0, 0}, nxt = {0, 0, 0};
    uint8_t* visited = (uint8_t*)malloc(img->h.n_nodes);
    mstate_t s0; s0.pos = 0; s0.node = (int32_t)img->h.start;
    for (int k = 0; k < 10; k++) s0.cell[k] = -1;
    sset_insert(&cur, &s0);                                                      /* mfa.cpp:217-219 */
    for (int64_t i = 0; i <= len; i++) {            /* i == len is the final pass, mfa.cpp:227-228 */
        memset(visited, 0, img->h.n_nodes);
        nxt.n = 0;
        if (stats) { stats->steps++; if (cur.n > stats->max_states) stats->max_states = cur.n; }
        for (size_t k = 0; k < cur.n; k++) {                                     /* mfa.cpp:203-213 */
            mstate_t s = cur.v[k];
            if (!visited[s.node]) {
                /* fixture classification only (tests/golden/classify.py): let the LAST of the states that tie on
                 * (pos, node) win instead of the first, to see whether the answer depends on the tie-break at all */
                if (g_tie_last)
                    while (k + 1 < cur.n && cur.v[k + 1].node == s.node && cur.v[k + 1].pos == s.pos) s = cur.v[++k];
                visited[s.node] = 1;
                if (stats) stats->evaluations++;
                m_eval_state(&c, s, i, &nxt);
            }
        }
        sset_t t = cur; cur = nxt; nxt = t;
        if (i < len && cur.n == 0) i = len - 1;                      /* mfa.cpp:224-225 */
    }
    int res = 0;
    for (size_t k = 0; k < cur.n; k++) if ((uint32_t)cur.v[k].node == img->h.finish) res = 1;
    free(cur.v); free(nxt.v); free(visited); free(c.pool);
    return res;
}

/* ------------------------------------------------------------------ NFA --- */

typedef struct {
    const mfa_oracle_image* img;
    uint8_t* nxt;       /* new_states membership */
    uint8_t* visited;
} nctx_t;

/* Automata::evaluateState, automata.cpp:98-117.  letter < 0 is the empty string of the final pass */
static void n_eval_state(nctx_t* c, uint32_t node, int letter) {
    const mfa_oracle_image* g = c->img;
    if (letter < 0 && node == g->h.finish) {
        c->nxt[node] = 1;
    } else {
        for (uint32_t ei = g->edge_begin[node]; ei < g->edge_begin[node + 1]; ei++) {
            const mfa_blob_edge* e = &g->edges[ei];
            if (c->visited[e->target]) continue;                                 /* automata.cpp:105-107 */
            if (e->flags & MFA_EDGE_EPS) n_eval_state(c, e->target, letter);
            else if (letter >= 0 && (e->label == '.' || e->label == (uint8_t)letter)) c->nxt[e->target] = 1;
        }
    }
    c->visited[node] = 1;                                                        /* automata.cpp:116 */
}

static int match_nfa(const mfa_oracle_image* img, const uint8_t* str, int64_t len, mfa_oracle_stats* stats) {
    uint32_t n = img->h.n_nodes;
    uint8_t* cur = (uint8_t*)calloc(n, 1); uint8_t* nxt = (uint8_t*)calloc(n, 1); uint8_t* vis = (uint8_t*)calloc(n, 1);
    nctx_t c; c.img = img; c.nxt = nxt; c.visited = vis;
    cur[img->h.start] = 1;                                                       /* automata.cpp:178-179 */
    for (int64_t k = 0; k <= len; k++) {
        int letter = -1;
        if (k < len) letter = img->h.is_reversed ? str[len - 1 - k] : str[k];   /* automata.cpp:181-200 */
        memset(vis, 0, n); memset(nxt, 0, n);
        c.nxt = nxt;
        if (stats) stats->steps++;
        for (uint32_t v = 0; v < n; v++)                                         /* automata.cpp:119-128, pointer order */
            if (cur[v] && !vis[v]) { if (stats) stats->evaluations++; n_eval_state(&c, v, letter); }
        uint8_t* t = cur; cur = nxt; nxt = t;
        if (k < len) {
            int any = 0; for (uint32_t v = 0; v < n; v++) any |= cur[v];
            if (!any) k = len - 1;                                               /* break; the final pass still runs */
        }
    }
    int res = cur[img->h.finish] != 0;                                           /* automata.cpp:204-208 */
    free(cur); free(nxt); free(vis);
    return res;
}

int mfa_oracle_match(const mfa_oracle_image* img, const uint8_t* str, uint64_t len, mfa_oracle_stats* stats) {
    if (!img || (!str && len)) return -1;
    if (len > 0x7fffffffu) return -2;          /* the reference indexes with int (mfa.cpp:220) */
    return img->h.kind == MFA_KIND_MFA ? match_mfa(img, str, (int64_t)len, stats)
                                       : match_nfa(img, str, (int64_t)len, stats);
}

int mfa_oracle_match_batch(const mfa_oracle_image* img, const uint8_t* bytes, const uint64_t* offsets,
                           uint64_t n, uint8_t* results, mfa_oracle_stats* stats) {
    if (!img || !offsets || !results) return -1;
    for (uint64_t k = 0; k < n; k++) {
        if (offsets[k + 1] < offsets[k]) return -3;
        int r = mfa_oracle_match(img, bytes + offsets[k], offsets[k + 1] - offsets[k], stats);
        if (r < 0) return r;
        results[k] = (uint8_t)r;
    }
    return 0;
}
