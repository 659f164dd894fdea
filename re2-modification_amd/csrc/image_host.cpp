// Host side of an automaton image: blob parsing, the structural checks the MFA kernel
// relies on, and tabulation of the memory-less step function for the table-walk kernel.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>

#include "mfa_internal.h"

namespace mfa {

int parse_blob(const void* blob, size_t n_bytes, HostImage& out) {
    if (!blob || n_bytes < sizeof(mfa_blob_header)) return MFA_ERR_BAD_BLOB;
    mfa_blob_header h;
    std::memcpy(&h, blob, sizeof h);
    if (h.magic != MFA_BLOB_MAGIC || h.version != MFA_BLOB_VERSION) return MFA_ERR_BAD_BLOB;
    if (h.kind > MFA_KIND_MFA || h.n_nodes == 0 || h.n_nodes > 0xffffu) return MFA_ERR_BAD_BLOB;
    if (h.start >= h.n_nodes || h.finish >= h.n_nodes || h.n_cells > MFA_MAX_CELLS) return MFA_ERR_BAD_BLOB;
    size_t need = sizeof h + (size_t)(h.n_nodes + 1) * 4 + (size_t)h.n_edges * sizeof(mfa_blob_edge);
    if (n_bytes < need) return MFA_ERR_BAD_BLOB;
    out.h = h;
    out.edge_begin.resize(h.n_nodes + 1);
    out.edges.resize(h.n_edges);
    const char* p = (const char*)blob + sizeof h;
    std::memcpy(out.edge_begin.data(), p, (size_t)(h.n_nodes + 1) * 4);
    p += (size_t)(h.n_nodes + 1) * 4;
    if (h.n_edges) std::memcpy(out.edges.data(), p, (size_t)h.n_edges * sizeof(mfa_blob_edge));
    if (out.edge_begin[0] != 0 || out.edge_begin[h.n_nodes] != h.n_edges) return MFA_ERR_BAD_BLOB;
    for (uint32_t k = 0; k < h.n_nodes; k++)
        if (out.edge_begin[k] > out.edge_begin[k + 1]) return MFA_ERR_BAD_BLOB;
    for (const auto& e : out.edges) {
        if (e.target >= h.n_nodes) return MFA_ERR_BAD_BLOB;
        for (unsigned c = h.n_cells + 1; c <= MFA_MAX_CELLS; c++)
            if (MFA_EDGE_ACTION(e, c)) return MFA_ERR_BAD_BLOB;
        if (MFA_EDGE_ACTION(e, 0u)) return MFA_ERR_BAD_BLOB;
        for (unsigned c = 1; c <= MFA_MAX_CELLS; c++)
            if (MFA_EDGE_ACTION(e, c) == 3u) return MFA_ERR_BAD_BLOB;
        if (!(e.flags & MFA_EDGE_EPS) && e.label >= '1' && e.label <= '9' && (uint32_t)(e.label - '0') > h.n_cells &&
            h.kind == MFA_KIND_MFA)
            return MFA_ERR_BAD_BLOB;
    }
    return MFA_OK;
}

// What bt/bt_mfa.cpp guarantees for every automaton it builds (SURVEY.md section 8a,
// "Structural invariants") and what the MFA kernel is written against:
//   * every epsilon edge targets `finish`, and nothing but epsilon edges does
//     (so states at `finish` only ever exist with pos == len: an accept flag);
//   * `finish` has no out-edges.
// An image that breaks them is refused (there is no CPU fallback behind this library).
int check_mfa_invariants(const HostImage& img) {
    const auto& h = img.h;
    if (h.n_nodes > MFA_MAX_NODES || h.n_cells > MFA_MAX_KERNEL_CELLS) return MFA_ERR_UNSUPPORTED;
    if (img.edge_begin[h.finish + 1] != img.edge_begin[h.finish]) return MFA_ERR_UNSUPPORTED;
    for (uint32_t n = 0; n < h.n_nodes; n++) {
        for (uint32_t e = img.edge_begin[n]; e < img.edge_begin[n + 1]; e++) {
            bool eps = img.edges[e].flags & MFA_EDGE_EPS;
            bool to_finish = img.edges[e].target == h.finish;
            if (eps != to_finish) return MFA_ERR_UNSUPPORTED;
        }
    }
    return MFA_OK;
}

// ---- memory-less automata: tabulate Automata::evaluateStates ---------------------------------
//
// automata.cpp:119-128 maps (state set, letter) -> state set and depends on nothing else
// (letter_index is unused), so it can be tabulated once per automaton.  The tabulation runs
// the reference's own step -- including its quirks: nodes are visited in pointer (= node
// number) order, an edge whose target is already in `visited` is skipped even when it is a
// letter edge (automata.cpp:105-107), and a node is marked visited only after its edges were
// walked (automata.cpp:116).
namespace {

struct Stepper {
    const HostImage& g;
    std::vector<uint8_t> nxt, vis;
    explicit Stepper(const HostImage& img) : g(img), nxt(img.h.n_nodes), vis(img.h.n_nodes) {}

    void eval_state(uint32_t node, int letter) {           // automata.cpp:98-117; letter < 0: the final pass's ""
        if (letter < 0 && node == g.h.finish) {
            nxt[node] = 1;
        } else {
            for (uint32_t e = g.edge_begin[node]; e < g.edge_begin[node + 1]; e++) {
                const mfa_blob_edge& ed = g.edges[e];
                if (vis[ed.target]) continue;
                if (ed.flags & MFA_EDGE_EPS) eval_state(ed.target, letter);
                else if (letter >= 0 && (ed.label == '.' || ed.label == (uint8_t)letter)) nxt[ed.target] = 1;
            }
        }
        vis[node] = 1;
    }

    std::vector<uint8_t> step(const std::vector<uint8_t>& cur, int letter) {   // automata.cpp:119-128
        std::fill(nxt.begin(), nxt.end(), 0);
        std::fill(vis.begin(), vis.end(), 0);
        for (uint32_t v = 0; v < g.h.n_nodes; v++)
            if (cur[v] && !vis[v]) eval_state(v, letter);
        return nxt;
    }
};

}  // namespace

int tabulate_nfa(HostImage& img) {
    const uint32_t n = img.h.n_nodes;
    // byte classes: one per distinct literal label, one for every other byte ('.' edges match all)
    int cls_of[256];
    for (int b = 0; b < 256; b++) cls_of[b] = -1;
    std::vector<int> rep;           // representative byte per class
    int other_rep = -1;
    for (const auto& e : img.edges)
        if (!(e.flags & MFA_EDGE_EPS) && e.label != '.' && cls_of[e.label] < 0) {
            cls_of[e.label] = (int)rep.size();
            rep.push_back(e.label);
        }
    for (int b = 0; b < 256; b++)
        if (cls_of[b] < 0 && b != '.') { other_rep = b; break; }
    int other_cls = (int)rep.size();
    rep.push_back(other_rep < 0 ? 0 : other_rep);
    for (int b = 0; b < 256; b++) img.byte_class[b] = (uint8_t)(cls_of[b] >= 0 ? cls_of[b] : other_cls);
    // a '.' byte in the input only matches '.' labels, like any other unlabelled byte -- unless some
    // edge carries the literal label '.', which the reference treats as the wildcard (automata.cpp:111)
    img.n_classes = (uint32_t)rep.size();
    if (img.n_classes > 255) return MFA_ERR_UNSUPPORTED;

    // state sets as packed bit strings (up to MFA_MAX_DFA_STATES of them: determinisation can be exponential, e.g.
    // (a|b)*a(a|b)^k has 2^(k+1) sets)
    Stepper st(img);
    const size_t key_bytes = (n + 7) / 8;
    auto pack = [&](const std::vector<uint8_t>& set) {
        std::string k(key_bytes, '\0');
        for (uint32_t v = 0; v < n; v++)
            if (set[v]) k[v >> 3] = (char)(k[v >> 3] | (1 << (v & 7)));
        return k;
    };
    auto unpack = [&](const std::string& k, std::vector<uint8_t>& set) {
        for (uint32_t v = 0; v < n; v++) set[v] = (uint8_t)((k[v >> 3] >> (v & 7)) & 1);
    };
    std::unordered_map<std::string, uint32_t> ids;
    std::vector<std::string> sets;
    std::vector<uint8_t> empty(n, 0), start(n, 0), cur(n, 0);
    start[img.h.start] = 1;
    ids[pack(empty)] = 0; sets.push_back(pack(empty));
    if (start != empty) { ids[pack(start)] = 1; sets.push_back(pack(start)); }
    img.dfa_trans.clear();
    for (size_t s = 0; s < sets.size(); s++) {
        unpack(sets[s], cur);
        for (uint32_t c = 0; c < img.n_classes; c++) {
            const std::string t = pack(s == 0 ? empty : st.step(cur, rep[c]));
            auto it = ids.find(t);
            uint32_t id;
            if (it == ids.end()) {
                id = (uint32_t)sets.size();
                if (id >= MFA_MAX_DFA_STATES) return MFA_ERR_UNSUPPORTED;
                ids.emplace(t, id); sets.push_back(t);
            } else id = it->second;
            img.dfa_trans.push_back(id);
        }
    }
    img.dfa_states = (uint32_t)sets.size();
    img.dfa_accept.assign(img.dfa_states, 0);
    for (uint32_t s = 1; s < img.dfa_states; s++) {
        unpack(sets[s], cur);
        std::vector<uint8_t> f = st.step(cur, -1);            // automata.cpp:201-202
        img.dfa_accept[s] = f[img.h.finish];                  // automata.cpp:204-208
    }
    return MFA_OK;
}

}  // namespace mfa
