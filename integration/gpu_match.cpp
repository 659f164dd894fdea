// Binding B of INTEGRATION.md: the reference's own front-end (parser, BNF rewriter, toMFA / toGlushkov / toThomson) with only
// the match loop on the MI355X.  This is the translation unit a maintainer adds to the reference tree (CMakeLists.txt:8) and
// links with -lmfa_hip; it includes the reference's headers and this repository's include/mfa_hip.h, nothing else.
//
//   reference                                          here
//   bool MFA::match(string)            mfa.cpp:215     diploma_gpu::match_batch(MFA*, ...)      one launch for a batch
//   bool Automata::match(const string&) automata.cpp:177  diploma_gpu::match_batch(Automata*, ...)
//   the heap graph behind them          automata.h:18-84  diploma_gpu::freeze(...): the flat image of include/mfa_image_format.h
//
// Free functions, so that no reference header has to change; `MFA::match_batch` / `Automata::match_batch` members that
// forward to them are a two-line edit of automata.h.
// tests/test_integration_binding.py compiles this file against /root/reference (build container only) and checks that
// freeze() of the reference's graphs equals the committed golden images.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "automata.h"
#include "mfa_hip.h"

namespace diploma_gpu {

// Freeze a graph into the image format.  Nodes are numbered in pointer order -- the order the reference's own
// std::set<MemoryState> uses (automata.h:12-13).
template <class NodeT, class Fn>
static std::vector<uint8_t> freeze_graph(uint32_t kind, bool reversed, std::vector<NodeT*> nodes, NodeT* start, NodeT* finish, Fn actions_of) {
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    auto rank = [&](NodeT* n) { return (uint32_t)(std::lower_bound(nodes.begin(), nodes.end(), n) - nodes.begin()); };
    std::vector<uint32_t> begin{0};
    std::vector<mfa_blob_edge> edges;
    uint32_t n_cells = 0;
    for (NodeT* n : nodes) {
        for (auto* e : n->edges) {
            mfa_blob_edge be{};
            if (e->by.empty() || e->by == "\xce\xb5") be.flags = MFA_EDGE_EPS;      // "" or the epsilon sign MFA::makeDOTFile writes (mfa.cpp:40-42)
            else be.label = (uint8_t)e->by[0];
            if (kind == MFA_KIND_MFA && !be.flags && be.label >= '1' && be.label <= '9') n_cells = std::max<uint32_t>(n_cells, be.label - '0');
            be.target = (uint16_t)rank(e->to);
            be.actions = actions_of(e, n_cells);
            edges.push_back(be);
        }
        begin.push_back((uint32_t)edges.size());
    }
    mfa_blob_header h{MFA_BLOB_MAGIC, MFA_BLOB_VERSION, kind, reversed ? 1u : 0u, (uint32_t)nodes.size(), (uint32_t)edges.size(),
                      rank(start), rank(finish), n_cells, 0};
    std::vector<uint8_t> blob(sizeof h + begin.size() * 4 + edges.size() * sizeof(mfa_blob_edge));
    std::memcpy(blob.data(), &h, sizeof h);
    std::memcpy(blob.data() + sizeof h, begin.data(), begin.size() * 4);
    if (!edges.empty()) std::memcpy(blob.data() + sizeof h + begin.size() * 4, edges.data(), edges.size() * sizeof(mfa_blob_edge));
    return blob;
}

std::vector<uint8_t> freeze(MFA* m) {
    std::vector<MemoryNode*> all(m->nodes.begin(), m->nodes.end());
    return freeze_graph(MFA_KIND_MFA, m->is_reversed, all, m->start, m->finish, [](MemoryEdge* e, uint32_t& n_cells) {
        uint32_t a = 0;
        for (auto& kv : e->memoryActions) {
            const uint32_t c = (uint32_t)(kv.first[0] - '0');
            a |= (uint32_t)(kv.second == open ? MFA_ACT_OPEN : MFA_ACT_CLOSE) << (2 * c);
            n_cells = std::max(n_cells, c);
        }
        return a;
    });
}

std::vector<uint8_t> freeze(Automata* a) {
    std::vector<Node*> all(a->nodes.begin(), a->nodes.end());
    return freeze_graph(MFA_KIND_NFA, a->is_reversed, all, a->start, a->finish, [](Edge*, uint32_t&) { return 0u; });
}

static void run(const std::vector<uint8_t>& blob, const uint8_t* bytes, const uint64_t* off, uint64_t n, uint8_t* res) {
    mfa_image_t* img = nullptr;
    int rc = mfa_image_create(blob.data(), blob.size(), &img);
    if (rc == MFA_OK) rc = mfa_match_batch_host(img, bytes, off, n, res, /*device=*/0);
    mfa_image_destroy(img);
    if (rc != MFA_OK) throw std::runtime_error(mfa_strerror(rc));
}

// results[k] = what m->match(string k) returns; string k is bytes[off[k], off[k+1])
void match_batch(MFA* m, const uint8_t* bytes, const uint64_t* off, uint64_t n, uint8_t* res) { run(freeze(m), bytes, off, n, res); }
void match_batch(Automata* a, const uint8_t* bytes, const uint64_t* off, uint64_t n, uint8_t* res) { run(freeze(a), bytes, off, n, res); }

}  // namespace diploma_gpu
