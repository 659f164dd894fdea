// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Harness around the *unmodified* reference sources (compiled where they lie
// under /root/reference by oracle/Makefile, objects only in oracle/_ref/).
// It never touches main.cpp / matchers/*.cpp (those do not compile at this
// commit: they call a 5-argument compile(), regex/regex.h:226 declares 4), it
// drives the same API they would:
//     Regexp::parse_regexp  (regex/parser.cpp:8)
//     Regexp::compile       (regex/regex.cpp:266)
//     MFA::match            (mfa.cpp:215)   /  Automata::match (automata.cpp:177)
//     BinaryTree::toThomson / toGlushkov / toMFA (bt/*.cpp)
//
// Canonical oracle mode: the reference orders states by raw
// heap pointers (automata.h:12-13), so every allocation the reference makes is
// served from a monotonic, zero-filled arena => pointer order == allocation
// order, and results no longer depend on glibc heap history.  The arena is
// rewound to the post-compile mark before every match() so memory stays bounded
// (the reference leaks every Variable, mfa.cpp:107-114).
//
// Commands (strings are read from stdin, one per line, an empty line is the
// empty string):
//   ref_harness dump  <mode> <regex>     -> automaton image (text) on stdout
//   ref_harness match <mode> <regex>     -> one 0/1 line per input string
//   ref_harness time  <mode> <regex>     -> "<n_strings> <bytes> <seconds>" (match loop only)
//   ref_harness front <regex>            -> "BNF: ..", "Reverse: .." lines (main.cpp:50-85 REPL body)
// modes: plain | bnf | reverse | thompson | glushkov | mfa (toMFA through the API)

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <string>
#include <vector>
#include <map>
#include <iostream>
#include <sstream>
#include <algorithm>
#include <sys/mman.h>

#include "regex/regex.h"
#include "bt/binary_tree.h"
#include "automata.h"

// ---------------------------------------------------------------- arena ----
static char*  g_arena = nullptr;
static size_t g_arena_cap = 0, g_arena_top = 0, g_arena_mark = 0, g_arena_hi = 0;
static bool   g_in_ref = false;          // route operator new to the arena?
static const size_t kSmall = 4096;       // everything <= kSmall is bump-allocated (all reference objects;
                                         // only long std::string buffers are larger)

struct BigHdr { BigHdr* prev; BigHdr* next; size_t tracked; size_t pad; };
static BigHdr g_big_head = { &g_big_head, &g_big_head, 0, 0 };

static void arena_init() {
    g_arena_cap = (size_t)48 << 30;
    g_arena = (char*)mmap(nullptr, g_arena_cap, PROT_READ | PROT_WRITE,
                          MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_arena == MAP_FAILED) { perror("mmap"); abort(); }
}

static void* big_alloc(size_t n, bool tracked) {
    BigHdr* h = (BigHdr*)malloc(sizeof(BigHdr) + n);
    if (!h) abort();
    h->tracked = tracked;
    if (tracked) {
        h->next = g_big_head.next; h->prev = &g_big_head;
        g_big_head.next->prev = h; g_big_head.next = h;
    } else { h->prev = h->next = nullptr; }
    return (void*)(h + 1);
}

static void big_free(void* p) {
    BigHdr* h = ((BigHdr*)p) - 1;
    if (h->tracked) { h->prev->next = h->next; h->next->prev = h->prev; }
    free(h);
}

// REF_HARNESS_ALLOC=glibc: no arena -- the reference on glibc's heap, as a user runs it (fixture class C: where that differs)
static const bool g_glibc = getenv("REF_HARNESS_ALLOC") && !strcmp(getenv("REF_HARNESS_ALLOC"), "glibc");

void* operator new(size_t n) {
    if (g_glibc) { void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
    if (g_in_ref && n <= kSmall) {
        if (!g_arena) arena_init();
        size_t a = (n + 15) & ~(size_t)15;
        if (a == 0) a = 16;
        if (g_arena_top + a > g_arena_cap) { fprintf(stderr, "arena exhausted\n"); abort(); }
        void* p = g_arena + g_arena_top;
        g_arena_top += a;
        if (g_arena_top > g_arena_hi) g_arena_hi = g_arena_top;
        return p;
    }
    return big_alloc(n, g_in_ref);
}
void* operator new[](size_t n) { return operator new(n); }
// the reference releases one new-ed object with ::free (regex/bnf.cpp:222): ignore arena pointers
extern "C" void __libc_free(void*);
extern "C" void free(void* p) {
    if (g_arena && (char*)p >= g_arena && (char*)p < g_arena + g_arena_cap) return;
    __libc_free(p);
}
void operator delete(void* p) noexcept {
    if (!p) return;
    if (g_glibc) { __libc_free(p); return; }
    if (g_arena && (char*)p >= g_arena && (char*)p < g_arena + g_arena_cap) return;
    big_free(p);
}
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }

static void arena_set_mark() {
    g_arena_mark = g_arena_top;
    // blocks the reference allocated while compiling belong to the automaton: keep them for good
    while (g_big_head.next != &g_big_head) {
        BigHdr* h = g_big_head.next;
        h->prev->next = h->next; h->next->prev = h->prev;
        h->tracked = 0; h->prev = h->next = nullptr;
    }
}
static void arena_rewind() {
    // zero what the last match dirtied so reused memory looks like fresh memory
    if (g_arena_top > g_arena_mark) memset(g_arena + g_arena_mark, 0, g_arena_top - g_arena_mark);
    g_arena_top = g_arena_mark;
    while (g_big_head.next != &g_big_head) {   // blocks the reference leaked
        BigHdr* h = g_big_head.next;
        h->prev->next = h->next; h->next->prev = h->prev;
        free(h);
    }
}

// ------------------------------------------------------------- helpers ----
struct Compiled {
    Automata* nfa = nullptr;   // memory-less automaton (Automata::match)
    MFA*      mfa = nullptr;   // memory automaton (MFA::match, non-virtual)
    std::string header;        // what compile() printed
};

static Compiled compile_mode(const std::string& mode, std::string regex) {
    Compiled c;
    std::ostringstream sink;
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    g_in_ref = true;
    Regexp* re = Regexp::parse_regexp(regex);
    if (mode == "plain" || mode == "bnf" || mode == "reverse" || mode == "ssnf" || mode == "all") {
        bool is_mfa = false;
        bool rev = (mode == "reverse" || mode == "all"), bnf = (mode == "bnf" || mode == "reverse" || mode == "all");
        Automata* a = re->compile(is_mfa, rev, bnf, mode == "ssnf" || mode == "all");
        if (is_mfa) c.mfa = static_cast<MFA*>(a); else c.nfa = a;
    } else if (mode == "thompson") {
        c.nfa = re->to_binary_tree()->toThomson();
    } else if (mode == "glushkov") {
        c.nfa = re->to_binary_tree()->toGlushkov();
    } else if (mode == "mfa") {
        re->is_backref_correct();
        c.mfa = re->to_binary_tree()->toMFA();
    } else {
        g_in_ref = false; std::cout.rdbuf(old);
        fprintf(stderr, "unknown mode %s\n", mode.c_str()); exit(2);
    }
    g_in_ref = false;
    std::cout.rdbuf(old);
    c.header = sink.str();
    arena_set_mark();
    return c;
}

static std::string hexlabel(const std::string& by) {
    if (by.empty() || by == "\xce\xb5") return "-";      // epsilon (mfa.cpp:40-42 rewrites "" to the UTF-8 letter)
    static const char* d = "0123456789abcdef";
    std::string h;
    for (unsigned char ch : by) { h += d[ch >> 4]; h += d[ch & 15]; }
    return h;
}

template <class NodeT>
static std::map<const void*, int> index_nodes(const std::list<NodeT*>& nodes, std::vector<const void*>& order) {
    std::map<const void*, int> idx;
    for (auto* n : nodes) if (!idx.count(n)) { idx[n] = (int)order.size(); order.push_back(n); }
    return idx;
}

static void dump(const Compiled& c) {
    std::vector<const void*> order;
    if (c.mfa) {
        MFA* m = c.mfa;
        auto idx = index_nodes(m->nodes, order);
        // any edge target / start / finish missing from the list is appended
        auto add = [&](const void* p) { if (!idx.count(p)) { idx[p] = (int)order.size(); order.push_back(p); } };
        add(m->start); add(m->finish);
        for (size_t k = 0; k < order.size(); k++)
            for (auto* e : ((MemoryNode*)order[k])->edges) add(e->to);
        std::vector<const void*> sorted(order); std::sort(sorted.begin(), sorted.end());
        printf("kind mfa\nreversed %d\nnodes %zu\nstart %d\nfinish %d\n", (int)m->is_reversed, order.size(),
               idx[m->start], idx[m->finish]);
        for (size_t k = 0; k < order.size(); k++) {
            auto* n = (MemoryNode*)order[k];
            int rank = (int)(std::lower_bound(sorted.begin(), sorted.end(), order[k]) - sorted.begin());
            printf("node %zu %d %zu\n", k, rank, n->edges.size());
            for (auto* e : n->edges) {
                printf("edge %s %d", hexlabel(e->by).c_str(), idx[e->to]);
                for (auto& a : e->memoryActions) printf(" %c%s", a.second == open ? 'o' : 'c', a.first.c_str());
                printf("\n");
            }
        }
    } else {
        Automata* m = c.nfa;
        auto idx = index_nodes(m->nodes, order);
        auto add = [&](const void* p) { if (!idx.count(p)) { idx[p] = (int)order.size(); order.push_back(p); } };
        add(m->start); add(m->finish);
        for (size_t k = 0; k < order.size(); k++)
            for (auto* e : ((Node*)order[k])->edges) add(e->to);
        std::vector<const void*> sorted(order); std::sort(sorted.begin(), sorted.end());
        printf("kind nfa\nreversed %d\nnodes %zu\nstart %d\nfinish %d\n", (int)m->is_reversed, order.size(),
               idx[m->start], idx[m->finish]);
        for (size_t k = 0; k < order.size(); k++) {
            auto* n = (Node*)order[k];
            int rank = (int)(std::lower_bound(sorted.begin(), sorted.end(), order[k]) - sorted.begin());
            printf("node %zu %d %zu\n", k, rank, n->edges.size());
            for (auto* e : n->edges) printf("edge %s %d\n", hexlabel(e->by).c_str(), idx[e->to]);
        }
    }
}

static bool run_match(const Compiled& c, const std::string& s) {
    if (!g_glibc) arena_rewind();
    g_in_ref = true;
    bool r = c.mfa ? c.mfa->match(s) : c.nfa->match(s);
    g_in_ref = false;
    return r;
}

static std::vector<std::string> read_lines() {
    std::vector<std::string> v; std::string line;
    while (std::getline(std::cin, line)) v.push_back(line);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_harness dump|match|time <mode> <regex> | front|frontlog <regex>\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "front" || cmd == "frontlog") {
        // body of the BNF/Reverse REPL, main.cpp:55-66; frontlog: with -log (the rewrite trace goes to log.txt in the working directory)
        std::string regex = argv[2];
        g_in_ref = true;
        Regexp* re = Regexp::parse_regexp(regex);
        re->is_backref_correct();
        Regexp* b = re->bnf(cmd == "frontlog");
        if (!b->is_bad_bnf) {
            std::string bs = b->to_string();
            Regexp* r = b->reverse();
            std::string rs = r->to_string();
            g_in_ref = false;
            printf("BNF: %s\nReverse: %s\n", bs.c_str(), rs.c_str());
        } else { g_in_ref = false; printf("BAD\n"); }
        return 0;
    }
    if (argc < 4) return 2;
    Compiled c = compile_mode(argv[2], argv[3]);
    if (cmd == "dump") { dump(c); return 0; }
    if (cmd == "header") { fputs(c.header.c_str(), stdout); return 0; }
    std::vector<std::string> in = read_lines();
    if (cmd == "match") {
        std::string out; out.reserve(in.size() * 2);
        for (auto& s : in) { out += run_match(c, s) ? '1' : '0'; out += '\n'; }
        fputs(out.c_str(), stdout);
        return 0;
    }
    if (cmd == "time") {
        size_t bytes = 0; for (auto& s : in) bytes += s.size();
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        unsigned acc = 0;
        for (auto& s : in) acc += run_match(c, s);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        printf("%zu %zu %.6f %u\n", in.size(), bytes, sec, acc);
        return 0;
    }
    return 2;
}
