// Automata / MFA objects of the host mirror: graph bookkeeping the reference does in
// automata.cpp:16-96 and mfa.cpp:11-77, freezing a graph into an automaton image, and match()
// through the C-ABI of libmfa_hip.so.
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <thread>

#include "../../include/mfa_hip.h"
#include "diploma_api.h"

namespace {

[[noreturn]] void fail(const char* where, int code) {
    std::string msg = std::string(where) + ": " + mfa_strerror(code);
    if (code == MFA_ERR_HIP) msg += " (hipError " + std::to_string(mfa_last_hip_error()) + ")";
    throw std::runtime_error(msg);
}

void put32(vector<uint8_t>& out, uint32_t v) {
    for (int k = 0; k < 4; k++) out.push_back((uint8_t)(v >> (8 * k)));
}

struct FlatEdge { std::string by; size_t to; const map<string, MemoryAction>* actions; };
struct FlatNode { uint64_t seq; vector<FlatEdge> edges; };

// Serialise per include/mfa_image_format.h.  Nodes are numbered by allocation order (`seq`), the
// stand-in for the pointer order of the reference's state sets.
vector<uint8_t> serialise(uint32_t kind, bool reversed, vector<FlatNode>& nodes, size_t start, size_t finish) {
    vector<size_t> order(nodes.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return nodes[a].seq < nodes[b].seq; });
    vector<uint32_t> rank(nodes.size());
    for (size_t r = 0; r < order.size(); r++) rank[order[r]] = (uint32_t)r;
    vector<uint8_t> edges;
    vector<uint32_t> begin{0};
    uint32_t n_edges = 0, n_cells = 0;
    for (size_t r = 0; r < order.size(); r++) {
        for (const FlatEdge& e : nodes[order[r]].edges) {
            uint8_t label = 0, flags = 0;
            if (e.by.empty() || e.by == "\xce\xb5") flags |= MFA_EDGE_EPS;
            else if (e.by.size() == 1) label = (uint8_t)e.by[0];
            else throw std::runtime_error("automaton has a multi-byte edge label '" + e.by + "': not representable in an image");
            if (!flags && label >= '1' && label <= '9' && kind == MFA_KIND_MFA) n_cells = std::max<uint32_t>(n_cells, label - '0');
            uint32_t actions = 0;
            if (e.actions)
                for (const auto& kv : *e.actions) {
                    if (kv.first.size() != 1 || kv.first[0] < '1' || kv.first[0] > '9')
                        throw std::runtime_error("memory cell name '" + kv.first + "' is not a digit 1..9");
                    uint32_t c = (uint32_t)(kv.first[0] - '0');
                    actions |= (kv.second == open ? MFA_ACT_OPEN : MFA_ACT_CLOSE) << (2 * c);
                    n_cells = std::max(n_cells, c);
                }
            edges.push_back(label); edges.push_back(flags);
            edges.push_back((uint8_t)(rank[e.to] & 0xff)); edges.push_back((uint8_t)(rank[e.to] >> 8));
            put32(edges, actions);
            n_edges++;
        }
        begin.push_back(n_edges);
    }
    vector<uint8_t> out;
    for (uint32_t v : {MFA_BLOB_MAGIC, MFA_BLOB_VERSION, kind, (uint32_t)reversed, (uint32_t)nodes.size(), n_edges,
                       rank[start], rank[finish], n_cells, 0u})
        put32(out, v);
    for (uint32_t v : begin) put32(out, v);
    out.insert(out.end(), edges.begin(), edges.end());
    return out;
}

template <class NodeT, class Index>
size_t index_of(Index& idx, vector<NodeT*>& all, NodeT* n) {
    auto it = idx.find(n);
    if (it != idx.end()) return it->second;
    idx[n] = all.size();
    all.push_back(n);
    return all.size() - 1;
}

}  // namespace

// ---- Automata ---------------------------------------------------------------------------------------

Automata::Automata() {
    start = new Node();
    finish = new Node();
    nodes.push_back(start);
    nodes.push_back(finish);
    last_idx = 0;
}

Automata::~Automata() {
    if (cached_image_) mfa_image_destroy(cached_image_);
}

void Automata::changeFinalState(Node* new_final) {                               // automata.cpp:16-32
    for (Node* n : nodes)
        for (Edge* e : n->edges)
            if (e->to == finish) e->to = new_final;
    auto it = std::find(nodes.begin(), nodes.end(), finish);
    if (it != nodes.end()) nodes.erase(it);
    start->start_for = new_final;
}

bool Automata::isDeterministic() {                                               // automata.cpp:84-96
    set<pair<uint64_t, string>> seen;
    for (Node* n : nodes)
        for (Edge* e : n->edges)
            if (!seen.insert({n->seq, e->by}).second) return false;
    return true;
}

void Automata::makeDOTFile(const string& filename) {                             // automata.cpp:34-66
    string text = "digraph g {\n";
    auto name_of = [&](Node* n) -> const string& {
        if (n->name.empty()) n->name = std::to_string(last_idx++);
        return n->name;
    };
    for (Node* n : nodes) {
        name_of(n);
        for (Edge* e : n->edges) {
            name_of(e->to);
            string by = e->by.empty() ? "\xce\xb5" : (e->by == "." ? "dot" : e->by);
            if (!e->drawn) text += "\t" + n->name + " -> " + e->to->name + " [label=" + by + "]\n";
        }
    }
    text += "}";
    std::ofstream out(filename + ".dot");
    if (out.is_open()) out << text;
}

bool Automata::draw(const string& filename) {
    makeDOTFile(filename);
    return true;
}

vector<uint8_t> Automata::image_blob() const {
    map<Node*, size_t> idx;
    vector<Node*> all;
    for (Node* n : nodes) index_of(idx, all, n);
    index_of(idx, all, start);
    index_of(idx, all, finish);
    for (size_t k = 0; k < all.size(); k++)
        for (Edge* e : all[k]->edges) index_of(idx, all, e->to);
    vector<FlatNode> flat(all.size());
    for (size_t k = 0; k < all.size(); k++) {
        flat[k].seq = all[k]->seq;
        for (Edge* e : all[k]->edges) flat[k].edges.push_back({e->by, idx[e->to], nullptr});
    }
    return serialise(MFA_KIND_NFA, is_reversed, flat, idx[start], idx[finish]);
}

mfa_image* Automata::image_for_match() {
    vector<uint8_t> blob = image_blob();          // the graph's fields are public: re-freeze and compare
    if (!cached_image_ || blob != cached_blob_) {
        if (cached_image_) { mfa_image_destroy(cached_image_); cached_image_ = nullptr; }
        int rc = mfa_image_create(blob.data(), blob.size(), &cached_image_);
        if (rc != MFA_OK) fail("mfa_image_create", rc);
        cached_blob_ = std::move(blob);
    }
    return cached_image_;
}

extern "C" int diploma_partition_by_bytes(const uint64_t* offsets, uint64_t n, uint32_t parts, uint64_t* cuts) {
    if (!offsets || !cuts || parts == 0) return -1;
    const uint64_t total = offsets[n] - offsets[0];
    cuts[0] = 0;
    for (uint32_t r = 1; r < parts; r++) {
        // (128-bit product: a batch may hold more than 2^64 / parts bytes only in theory, but the rule should not depend on that)
        const uint64_t target = offsets[0] + (uint64_t)(((unsigned __int128)total * r) / parts);
        uint64_t k = (uint64_t)(std::lower_bound(offsets, offsets + n + 1, target) - offsets);
        if (k < cuts[r - 1]) k = cuts[r - 1];
        if (k > n) k = n;
        cuts[r] = k;
    }
    cuts[parts] = n;
    return 0;
}

void Automata::match_packed(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* results) {
    mfa_image* img = image_for_match();
    // the devices of the node: all of them by default (north_star: "the string batch shards trivially across the 8 GPUs of one node")
    const int count = mfa_device_count();
    int want = devices;
    if (const char* e = getenv("DIPLOMA_DEVICES")) want = atoi(e);
    unsigned use = count <= 0 ? 1u : (want <= 0 ? (unsigned)count : (unsigned)std::min(want, count));
    unsigned shards = use;
    if (const char* e = getenv("DIPLOMA_FORCE_SHARDS")) shards = (unsigned)std::max(1, atoi(e));      // (tests: several shards on the devices there are)
    const uint64_t total = n ? offsets[n] - offsets[0] : 0;
    if (shards <= 1 || (!getenv("DIPLOMA_FORCE_SHARDS") && (n < 2ull * shards || total < (4ull << 20)))) {
        int rc = mfa_match_batch_host(img, bytes, offsets, n, results, device);
        if (rc != MFA_OK) fail("mfa_match_batch_host", rc);
        return;
    }
    vector<uint64_t> cuts(shards + 1);
    diploma_partition_by_bytes(offsets, n, shards, cuts.data());
    vector<int> rcs(shards, MFA_OK);
    vector<std::thread> workers;
    for (unsigned r = 0; r < shards; r++) {
        if (cuts[r + 1] == cuts[r]) continue;
        const int dev = (device + (int)(r % use)) % std::max(count, 1);
        workers.emplace_back([&, r, dev]() {                      // (HIP's current device is per host thread; the C-ABI is re-entrant per device)
            rcs[r] = mfa_match_batch_host(img, bytes, offsets + cuts[r], cuts[r + 1] - cuts[r], results + cuts[r], dev);
        });
    }
    for (std::thread& t : workers) t.join();
    for (unsigned r = 0; r < shards; r++)
        if (rcs[r] != MFA_OK) fail("mfa_match_batch_host (one of the devices)", rcs[r]);
}

vector<bool> Automata::match_batch(const vector<string>& strs) {
    vector<uint64_t> off(strs.size() + 1, 0);
    for (size_t k = 0; k < strs.size(); k++) off[k + 1] = off[k] + strs[k].size();
    vector<uint8_t> bytes(off.back() + 16);
    for (size_t k = 0; k < strs.size(); k++) std::copy(strs[k].begin(), strs[k].end(), bytes.begin() + off[k]);
    vector<uint8_t> res(strs.size() + 1);
    match_packed(bytes.data(), off.data(), strs.size(), res.data());
    return vector<bool>(res.begin(), res.begin() + strs.size());
}

bool Automata::match(const string& str) {
    uint64_t off[2] = {0, str.size()};
    uint8_t res = 0;
    match_packed(reinterpret_cast<const uint8_t*>(str.data()), off, 1, &res);
    return res != 0;
}

// ---- MFA ----------------------------------------------------------------------------------------------

MFA::MFA() {
    start = new MemoryNode();
    finish = new MemoryNode();
    nodes.push_back(start);
    nodes.push_back(finish);
    last_idx = 0;
}

void MFA::changeFinalState(MemoryNode* new_final) {                              // mfa.cpp:11-26
    for (MemoryNode* n : nodes)
        for (MemoryEdge* e : n->edges)
            if (e->to == finish) e->to = new_final;
    auto it = std::find(nodes.begin(), nodes.end(), finish);
    if (it != nodes.end()) nodes.erase(it);
}

void MFA::makeDOTFile(const string& filename) {                                  // mfa.cpp:28-61
    string text = "digraph g {\n";
    for (MemoryNode* n : nodes) {
        if (n->name.empty()) n->name = std::to_string(last_idx++);
        for (MemoryEdge* e : n->edges) {
            if (e->to->name.empty()) e->to->name = std::to_string(last_idx++);
            if (e->by.empty()) e->by = "\xce\xb5";          // the reference rewrites the label itself (mfa.cpp:40-42)
            if (e->drawn) continue;
            string acts;
            for (const auto& kv : e->memoryActions) acts += string(kv.second == open ? "o" : "c") + kv.first + "/";
            text += "\t" + n->name + " -> " + e->to->name + " [label=\"" + e->by + "/" + acts + "\"]\n";
        }
    }
    text += "}";
    std::ofstream out(filename + ".dot");
    if (out.is_open()) out << text;
}

bool MFA::draw(const string& filename) {
    makeDOTFile(filename);
    return true;
}

vector<uint8_t> MFA::image_blob() const {
    map<MemoryNode*, size_t> idx;
    vector<MemoryNode*> all;
    for (MemoryNode* n : nodes) index_of(idx, all, n);
    index_of(idx, all, start);
    index_of(idx, all, finish);
    for (size_t k = 0; k < all.size(); k++)
        for (MemoryEdge* e : all[k]->edges) index_of(idx, all, e->to);
    vector<FlatNode> flat(all.size());
    for (size_t k = 0; k < all.size(); k++) {
        flat[k].seq = all[k]->seq;
        for (MemoryEdge* e : all[k]->edges) flat[k].edges.push_back({e->by, idx[e->to], &e->memoryActions});
    }
    return serialise(MFA_KIND_MFA, is_reversed, flat, idx[start], idx[finish]);
}

void MFA::match_packed(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* results) {
    Automata::match_packed(bytes, offsets, n, results);        // image_blob() is virtual: freezes the MFA graph
}

vector<bool> MFA::match_batch(const vector<string>& strs) { return Automata::match_batch(strs); }

vector<vector<bool>> match_mixed(const vector<MFA*>& automata, const vector<vector<string>>& strs) {
    if (automata.empty() || automata.size() != strs.size()) throw std::runtime_error("match_mixed: one list of strings per automaton");
    vector<mfa_image_t*> images;
    for (MFA* m : automata) images.push_back(m->image_for_match());
    mfa_mixed_t* mx = nullptr;
    int rc = mfa_mixed_create(images.data(), (uint32_t)images.size(), &mx);
    if (rc != MFA_OK) fail("mfa_mixed_create", rc);
    vector<uint64_t> off{0}, seg{0};
    vector<uint8_t> bytes;
    for (const auto& list : strs) {
        for (const string& s : list) { bytes.insert(bytes.end(), s.begin(), s.end()); off.push_back(bytes.size()); }
        seg.push_back(off.size() - 1);
    }
    bytes.resize(bytes.size() + 16);
    const uint64_t n = off.size() - 1;
    vector<uint8_t> res(n + 1);
    rc = mfa_match_mixed_host(mx, bytes.data(), off.data(), n, seg.data(), res.data(), automata[0]->device);
    mfa_mixed_destroy(mx);
    if (rc != MFA_OK) fail("mfa_match_mixed_host", rc);
    vector<vector<bool>> out;
    for (size_t k = 0; k < strs.size(); k++) out.emplace_back(res.begin() + seg[k], res.begin() + seg[k + 1]);
    return out;
}

bool MFA::match(string str) { return Automata::match(str); }
