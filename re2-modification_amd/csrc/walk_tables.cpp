// Host side of the live-list walk: automaton image -> tables (format: walk_tables.h).
#include "walk_tables.h"

#include <algorithm>
#include <map>

#include "mfa_internal.h"

namespace mfa {

namespace {

struct SymEdge {                 // an effective edge before vnodes have numbers
    uint32_t kind;               // 0 literal / waiting insertion, 1 read
    uint32_t cell;               // 0-based cell of a read
    uint32_t tnode, tmask;       // target vnode
    uint32_t actions;            // 2 bits per cell at bit 2(c-1)
    uint32_t cm, com, rdm;
};

struct VNode {
    uint32_t node, mask;
    bool has_eps = false, c_acc = false, qualifies = false;
    std::vector<std::vector<SymEdge>> b;      // per class
    std::vector<SymEdge> c;
};

struct Builder {
    const HostImage& g;
    uint32_t K, n_classes;
    uint8_t cmap[256];
    std::vector<int> class_byte;              // class -> its byte, -1 for "every other byte"

    explicit Builder(const HostImage& img) : g(img), K(img.h.n_cells ? img.h.n_cells : 1) {}

    uint32_t deg(uint32_t n) const { return g.edge_begin[n + 1] - g.edge_begin[n]; }
    const mfa_blob_edge& edge(uint32_t n, uint32_t k) const { return g.edges[g.edge_begin[n] + k]; }
    static bool eps(const mfa_blob_edge& e) { return e.flags & MFA_EDGE_EPS; }
    static int digit(const mfa_blob_edge& e) { return (!eps(e) && e.label >= '1' && e.label <= '9') ? e.label - '1' : -1; }
    // actions of the image (2 bits per cell c at bit 2c, c = 1..9) -> 2 bits per cell at bit 2(c-1)
    static uint32_t actions_of(const mfa_blob_edge& e) { return e.actions >> 2; }
    static uint32_t opens_of(const mfa_blob_edge& e) {
        uint32_t m = 0;
        for (uint32_t c = 0; c < 9; c++)
            if (((e.actions >> (2 * (c + 1))) & 3u) == MFA_ACT_OPEN) m |= 1u << c;
        return m;
    }

    void classes() {
        bool lit[256] = {false};
        for (const auto& e : g.edges)
            if (!eps(e) && e.label != '.') lit[e.label] = true;
        n_classes = 0;
        for (int b = 0; b < 256; b++)
            if (lit[b]) { cmap[b] = (uint8_t)n_classes++; class_byte.push_back(b); }
        for (int b = 0; b < 256; b++)
            if (!lit[b]) cmap[b] = (uint8_t)n_classes;
        class_byte.push_back(-1);
        n_classes++;
    }

    // mfa.cpp:195-197 is reached by letter, dot and present-cell edges only
    bool qualifies(uint32_t n, uint32_t mask) const {
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e)) continue;
            const int d = digit(e);
            if (d < 0 || (mask >> d) & 1u) return true;
        }
        return false;
    }

    // states with pos == i (mfa.cpp:161-194), the recursion of mfa.cpp:148-160 flattened; `rd` = is_read marks of this frame
    void flat_b(uint32_t n, uint32_t fm, uint32_t cm, uint32_t com, uint32_t rd, int cls, std::vector<SymEdge>& out) const {
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e)) continue;
            const int d = digit(e);
            if (d >= 0 && !((fm >> d) & 1u)) {
                if (e.target != g.h.finish) {
                    const bool open = ((e.actions >> (2 * (d + 1))) & 3u) == MFA_ACT_OPEN;
                    flat_b(e.target, fm | (1u << d), cm | (1u << d), com | (open ? 1u << d : 0u), rd, cls, out);
                }
                continue;
            }
            const bool lit = e.label == '.' || (class_byte[cls] >= 0 && class_byte[cls] == (int)e.label);
            if (lit) out.push_back(SymEdge{0u, 0u, e.target, fm | opens_of(e), actions_of(e), cm, com, rd});
            else if (d >= 0) {
                out.push_back(SymEdge{1u, (uint32_t)d, e.target, fm | opens_of(e), actions_of(e), cm, com, rd});
                rd |= 1u << d;
            }
        }
    }

    // waiting states (pos > i) and the final pass: only the recursion through absent-cell edges does anything
    void flat_c(uint32_t n, uint32_t fm, uint32_t cm, uint32_t com, bool root, VNode& v) const {
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e)) { if (!root) v.c_acc = true; continue; }
            const int d = digit(e);
            if (d < 0 || ((fm >> d) & 1u) || e.target == g.h.finish) continue;
            const bool open = ((e.actions >> (2 * (d + 1))) & 3u) == MFA_ACT_OPEN;
            const uint32_t fm2 = fm | (1u << d), cm2 = cm | (1u << d), com2 = com | (open ? 1u << d : 0u);
            flat_c(e.target, fm2, cm2, com2, false, v);
            if (qualifies(e.target, fm2)) v.c.push_back(SymEdge{0u, 0u, e.target, fm2, 0u, cm2, com2, 0u});
        }
    }
};

uint32_t fname_of(uint32_t mask) { return mask ? (uint32_t)__builtin_ctz(mask) + 1u : 0u; }

}  // namespace

int build_walk_tables(const HostImage& img, WalkTables& out, bool wide) {
    if (img.h.kind != MFA_KIND_MFA) return MFA_ERR_UNSUPPORTED;
    const uint32_t N = img.h.n_nodes;
    if (N > WT_MAX_NODES) return MFA_ERR_UNSUPPORTED;
    Builder b(img);
    b.classes();
    const uint32_t K = b.K, NC = b.n_classes;

    // reachable vnodes, breadth first from (start, no cells)
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> index;
    std::vector<VNode> vn;
    auto reach = [&](uint32_t node, uint32_t mask) {
        auto key = std::make_pair(node, mask);
        if (index.count(key)) return;
        index[key] = (uint32_t)vn.size();
        VNode v;
        v.node = node; v.mask = mask;
        vn.push_back(v);
    };
    reach(img.h.start, 0u);
    for (size_t at = 0; at < vn.size(); at++) {
        if (vn.size() > WT_MAX_VIDS) return MFA_ERR_UNSUPPORTED;
        VNode v = vn[at];
        for (uint32_t k = 0; k < b.deg(v.node); k++)
            if (Builder::eps(b.edge(v.node, k))) v.has_eps = true;
        v.qualifies = b.qualifies(v.node, v.mask);
        v.b.resize(NC);
        for (uint32_t c = 0; c < NC; c++) b.flat_b(v.node, v.mask, 0u, 0u, 0u, (int)c, v.b[c]);
        b.flat_c(v.node, v.mask, 0u, 0u, true, v);
        for (uint32_t c = 0; c < NC; c++)
            for (const auto& e : v.b[c]) reach(e.tnode, e.tmask);
        for (const auto& e : v.c) reach(e.tnode, e.tmask);
        vn[at] = v;
    }

    // vid = node << vb | variant
    std::vector<uint32_t> variants(N, 0);
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> vid;
    uint32_t most = 1;
    for (const auto& v : vn) {
        vid[std::make_pair(v.node, v.mask)] = variants[v.node]++;
        most = std::max(most, variants[v.node]);
    }
    uint32_t vb = 0;
    while ((1u << vb) < most) vb++;
    const uint32_t n_vids = N << vb;
    if (n_vids > WT_MAX_VIDS) return MFA_ERR_UNSUPPORTED;
    for (auto& kv : vid) kv.second |= kv.first.first << vb;

    const uint32_t eew = (K <= 6 && !wide) ? 2u : 3u;
    std::vector<uint32_t>& w = out.words;
    w.assign(WT_HDR, 0u);
    const uint32_t off_cmap = (uint32_t)w.size();
    w.resize(w.size() + 64, 0u);
    for (int bte = 0; bte < 256; bte++) w[off_cmap + bte / 4] |= (uint32_t)b.cmap[bte] << (8 * (bte % 4));
    const uint32_t off_vinfo = (uint32_t)w.size();
    w.resize(w.size() + n_vids, 0u);
    const uint32_t off_vc = (uint32_t)w.size();
    w.resize(w.size() + n_vids, 0u);
    const uint32_t off_vb = (uint32_t)w.size();
    w.resize(w.size() + (size_t)n_vids * NC, 0u);
    const uint32_t off_ee = (uint32_t)w.size();
    uint32_t n_ee = 0;
    auto put = [&](const SymEdge& e) {
        const uint32_t t = vid.at(std::make_pair(e.tnode, e.tmask));
        w.push_back(e.kind | (e.cell << 1) | (fname_of(e.tmask) << 5) | (t << 9));
        if (eew == 2) w.push_back(e.actions | (e.cm << 12) | (e.com << 18) | (e.rdm << 24));
        else { w.push_back(e.actions | (e.cm << 18)); w.push_back(e.com | (e.rdm << 9)); }
        n_ee++;
    };
    for (const auto& v : vn) {
        const uint32_t id = vid.at(std::make_pair(v.node, v.mask));
        w[off_vinfo + id] = 1u | (v.has_eps ? 2u : 0u) | (v.c_acc ? 4u : 0u) | (v.qualifies ? 8u : 0u) | (fname_of(v.mask) << 4) | (v.mask << 8) | (v.c.empty() ? 0u : 1u << 17);
        if (v.c.size() >= 4096 || n_ee >= (1u << 20)) return MFA_ERR_UNSUPPORTED;
        w[off_vc + id] = (n_ee << 12) | (uint32_t)v.c.size();
        for (const auto& e : v.c) put(e);
        for (uint32_t c = 0; c < NC; c++) {
            if (v.b[c].size() >= 4096 || n_ee >= (1u << 20)) return MFA_ERR_UNSUPPORTED;
            w[off_vb + id * NC + c] = (n_ee << 12) | (uint32_t)v.b[c].size();
            for (const auto& e : v.b[c]) put(e);
        }
    }
    w[0] = n_vids; w[1] = vb; w[2] = NC; w[3] = K; w[4] = vid.at(std::make_pair(img.h.start, 0u));
    w[5] = off_cmap; w[6] = off_vinfo; w[7] = off_vc; w[8] = off_vb; w[9] = off_ee; w[10] = n_ee; w[11] = eew;
    w[12] = (uint32_t)w.size(); w[13] = img.h.is_reversed; w[14] = N; w[15] = 0;
    out.n_vnodes = (uint32_t)vn.size();
    out.max_live = N > 1 ? N - 1 : 1;
    out.K = K;
    out.reversed = img.h.is_reversed != 0;
    return MFA_OK;
}

}  // namespace mfa
