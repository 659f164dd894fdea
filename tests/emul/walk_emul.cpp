// TEST INFRASTRUCTURE ONLY.  The walk kernel's own source (csrc/walk_core.h, csrc/walk_tables.cpp) compiled for the host as a
// wave of one lane, so that the step function, the table builder and the jump logic can be checked against the oracle on a
// machine without a GPU.  Not a fallback: nothing in re2-modification_amd/ or include/ builds, links or calls this.
//
//   walk_emul <image.blob> <C> <accel 0|1> [<image2.blob> <first string of segment 2> ...]  < strings (one per line)  > 0/1 per line
// With accel = 1 every string gets a region table computed here from the definition in include/mfa_hip.h (brute force).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>

#define MFA_HOST_EMUL 1
#define WALK_WV 1u
#define WALK_DEV inline
#include "walk_core.h"
#include "walk_tables.h"
#include "mfa_internal.h"

using namespace mfa;
using namespace mfa_walk;

static std::vector<uint8_t> slurp(const char* path) {
    std::vector<uint8_t> v;
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

// region table of one string, from the definition: maximal q-periodic regions (q <= 8) of at least MFA_REGION_MIN_LEN bytes, those
// covered by a region of a proper divisor period dropped, the 15 longest kept, sorted by start
static void region_row(const uint8_t* s, uint32_t len, uint64_t* row) {
    struct R { uint32_t lo, hi, q; };
    std::vector<R> all;
    for (uint32_t q = 1; q <= 8; q++) {
        if (len <= q) continue;
        uint32_t a = 0;
        while (a + q < len) {
            if (s[a] != s[a + q]) { a++; continue; }
            uint32_t b = a;
            while (b + q < len && s[b] == s[b + q]) b++;
            if (b + q - a >= MFA_REGION_MIN_LEN) all.push_back(R{a, b + q, q});
            a = b + 1;
        }
    }
    std::vector<R> keep;
    for (const R& r : all) {
        bool covered = false;
        for (const R& d : all)
            if (d.q < r.q && r.q % d.q == 0 && d.lo <= r.lo && d.hi >= r.hi) covered = true;
        if (!covered) keep.push_back(r);
    }
    bool overflow = false;
    if (keep.size() > MFA_REGION_MAX) {
        std::sort(keep.begin(), keep.end(), [](const R& x, const R& y) { return x.hi - x.lo > y.hi - y.lo; });
        keep.resize(MFA_REGION_MAX);
        overflow = true;
    }
    std::sort(keep.begin(), keep.end(), [](const R& x, const R& y) { return x.lo < y.lo || (x.lo == y.lo && x.q < y.q); });
    for (uint32_t k = 0; k < MFA_REGION_WORDS; k++) row[k] = 0;
    row[0] = keep.size() | (overflow ? MFA_REGION_OVERFLOW : 0ull);
    for (size_t k = 0; k < keep.size(); k++) row[1 + k] = (uint64_t)keep[k].lo | ((uint64_t)keep[k].hi << 24) | ((uint64_t)keep[k].q << 48);
}

struct QueueSeqFeeder {                       // the lean walk's strings: the queue the first pass filled
    const uint32_t* queue; uint32_t count, next = 0;
    bool take(bool want, uint64_t& sid) {
        if (!want || next >= count) return false;
        sid = queue[next++];
        return true;
    }
};

struct SeqFeeder {
    uint64_t next = 0, n = 0;
    bool take(bool want, uint64_t& sid) {
        if (!want || next >= n) return false;
        sid = next++;
        return true;
    }
};

template <int K, bool REV>
static void run(const Batch& b, const std::vector<uint32_t>& T, uint32_t C, uint32_t CM, WaveStats* ws) {
    const uint32_t CX = CM > C ? CM - C : 1u;
    std::vector<uint32_t> lv(2 * C * Lay<K>::W, 0xdeadbeefu), ld(2 * C * Lay<K>::DW, 0xdeadbeefu), sb(C * Lay<K>::W, 0xdeadbeefu), sa(C * Lay<K>::DW, 0xdeadbeefu),
        gv(2 * CX * Lay<K>::W, 0xdeadbeefu), gd(2 * CX * Lay<K>::DW, 0xdeadbeefu), gsb(CX * Lay<K>::W, 0xdeadbeefu), gsa(CX * Lay<K>::DW, 0xdeadbeefu), gq(CMP_CACHE * 4, 0xdeadbeefu);
    Store st;
    std::vector<uint32_t> nm((CM + 1u + 3u) / 4u + 1u, 0xdeadbeefu);      // WALK_NODE_MAP builds: the lane's node map
    st.nm = reinterpret_cast<uint8_t*>(nm.data()); st.nm_words = (CM + 1u + 3u) / 4u;
    st.lv = lv.data(); st.ld = ld.data(); st.sb = sb.data(); st.sa = sa.data(); st.gv = gv.data(); st.gd = gd.data(); st.gsb = gsb.data(); st.gsa = gsa.data(); st.gq = gq.data(); st.CI = C; st.C = C; st.CX = CX;
    uint64_t rtc[MFA_RT_CACHED] = {0};
    SeqFeeder feed;
    feed.n = b.n;
    walk_wave<K, REV, SeqFeeder>(b, T.data(), st, rtc, feed, ws);
    if (b.lean_queue != nullptr && *b.lean_count != 0u) {      // the strings handed on: the plain step only (walk_core.h: walk_wave_lean)
        Store sl = st;
        sl.ld = nullptr; sl.sb = nullptr; sl.sa = nullptr; sl.gd = nullptr; sl.gsb = nullptr; sl.gsa = nullptr; sl.CI = 0;
        QueueSeqFeeder qf{b.lean_queue, *b.lean_count};
        Batch bl = b;
        bl.regions = nullptr; bl.accel = 0; bl.lean_queue = nullptr; bl.lean_count = nullptr;
        walk_wave_lean<K, REV, QueueSeqFeeder>(bl, T.data(), sl, qf);
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: walk_emul image.blob C accel [image.blob first ...]\n"); return 2; }
    const uint32_t C = (uint32_t)atoi(argv[2]);
    const int accel = atoi(argv[3]);
    std::vector<uint32_t> T, seg_first, seg_table;
    uint32_t K = 1, CM = 1;
    bool rev = false;
    for (int a = 1; a < argc; a += (a == 1 ? 3 : 2)) {          // the launch's cell count: the largest of its automata's
        std::vector<uint8_t> blob = slurp(argv[a]);
        HostImage img;
        if (parse_blob(blob.data(), blob.size(), img) != MFA_OK) { fprintf(stderr, "bad image %s\n", argv[a]); return 2; }
        K = std::max(K, (uint32_t)(img.h.n_cells ? img.h.n_cells : 1));
    }
    for (int a = 1; a < argc; a += (a == 1 ? 3 : 2)) {
        std::vector<uint8_t> blob = slurp(argv[a]);
        HostImage img;
        if (parse_blob(blob.data(), blob.size(), img) != MFA_OK || check_mfa_invariants(img) != MFA_OK) { fprintf(stderr, "bad image %s\n", argv[a]); return 2; }
        WalkTables wt;
        if (build_walk_tables(img, wt, K > 6) != MFA_OK) { fprintf(stderr, "tables: unsupported %s\n", argv[a]); return 3; }
        seg_first.push_back(a == 1 ? 0u : (uint32_t)atoi(argv[a + 1]));
        seg_table.push_back((uint32_t)T.size());
        T.insert(T.end(), wt.words.begin(), wt.words.end());
        K = std::max(K, wt.K); CM = std::max(CM, wt.max_live);
        if (a == 1) rev = wt.reversed;
        else if (rev != wt.reversed) { fprintf(stderr, "segments of one launch must scan in the same direction\n"); return 2; }
    }
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off{0};
    {
        std::string line;
        int c;
        while ((c = getchar()) != EOF) {
            if (c == '\n') { bytes.insert(bytes.end(), line.begin(), line.end()); off.push_back(bytes.size()); line.clear(); }
            else line.push_back((char)c);
        }
    }
    const uint64_t n = off.size() - 1;
    seg_first.push_back((uint32_t)n);
    bytes.resize(bytes.size() + 64, 0);
    std::vector<uint8_t> res(n ? n : 1, 9);
    std::vector<uint64_t> table;
    if (accel) {
        table.resize(n * MFA_REGION_WORDS);
        for (uint64_t k = 0; k < n; k++) region_row(bytes.data() + off[k], (uint32_t)(off[k + 1] - off[k]), &table[k * MFA_REGION_WORDS]);
    }
    Batch b{bytes.data(), off.data(), n, res.data(), accel ? table.data() : nullptr, (uint32_t)(accel != 0), 1u, (uint32_t)seg_table.size(), seg_first.data(), seg_table.data(), 0u, nullptr, nullptr};
    std::vector<uint32_t> lean_queue(n ? n : 1);
    uint32_t lean_count = 0;
    if (accel && !getenv("EMUL_NO_LEAN")) { b.lean_queue = lean_queue.data(); b.lean_count = &lean_count; }
    WaveStats ws;
    if (n) {
#define GO(KK) do { if (rev) run<KK, true>(b, T, C, CM, &ws); else run<KK, false>(b, T, C, CM, &ws); } while (0)
        switch (K) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; case 5: GO(5); break;
                     case 6: GO(6); break; case 7: GO(7); break; case 8: GO(8); break; default: GO(9); break; }
    }
    for (uint64_t k = 0; k < n; k++) { putchar('0' + res[k]); putchar('\n'); }
    fprintf(stderr, "emul: %u of the strings walked by the lean pass\n", lean_count);
    fprintf(stderr, "emul: %llu strings, steps %llu, dual %llu, probes %llu, hits %llu, skipped %llu, spill-steps %llu\n", (unsigned long long)n, ws.steps, ws.dual,
            ws.probes, ws.hits, ws.skipped, ws.spills);
    if (getenv("EMUL_HIST")) { fprintf(stderr, "events: entries %llu edge-evals %llu inserts %llu search-iters %llu c-items %llu\n", g_ev[0], g_ev[1], g_ev[2], g_ev[3], g_ev[6]); for (int k = 0; k < 80; k++) if (ws.hist[k]) fprintf(stderr, " n=%d:%llu", k, ws.hist[k]); fprintf(stderr, "\n"); }
    return 0;
}
