// TEST INFRASTRUCTURE (tests/test_integration_binding.py): the reference's front-end builds an automaton, gpu_match.cpp freezes
// it, the blob goes to stdout as hex.  usage: freeze_check plain|bnf|reverse|thompson|glushkov <regex>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "regex/regex.h"
#include "bt/binary_tree.h"
#include "automata.h"

// Canonical allocation-order mode, as in oracle/ref_harness.cpp: every allocation comes from a monotonic
// arena, so pointer order == allocation order and the node numbering does not depend on glibc's free lists.
#include <cstdlib>
#include <new>
static char* g_arena = nullptr;
static size_t g_top = 0;
static const size_t kArena = size_t(1) << 30;
void* operator new(size_t n) {
    if (!g_arena) g_arena = static_cast<char*>(std::calloc(kArena, 1));
    n = (n + 15) & ~size_t(15);
    if (!g_arena || g_top + n > kArena) throw std::bad_alloc();
    void* p = g_arena + g_top;
    g_top += n;
    return p;
}
void operator delete(void*) noexcept {}
void operator delete(void*, size_t) noexcept {}
// the reference hands one object it got from `new` to ::free (regex/bnf.cpp:226)
extern "C" void __libc_free(void*);
extern "C" void free(void* p) {
    if (g_arena && static_cast<char*>(p) >= g_arena && static_cast<char*>(p) < g_arena + kArena) return;
    __libc_free(p);
}

namespace diploma_gpu {
std::vector<uint8_t> freeze(MFA* m);
std::vector<uint8_t> freeze(Automata* a);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    std::string regex = argv[2];
    std::streambuf* keep = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf());                       // compile() prints its header lines
    Regexp* re = Regexp::parse_regexp(regex);
    std::vector<uint8_t> blob;
    if (mode == "thompson" || mode == "glushkov") {
        BinaryTree* bt = re->to_binary_tree();
        blob = diploma_gpu::freeze(mode == "thompson" ? bt->toThomson() : bt->toGlushkov());
    } else {
        bool is_mfa = false;
        Automata* a = re->compile(is_mfa, mode == "reverse", mode != "plain", false);
        blob = is_mfa ? diploma_gpu::freeze(static_cast<MFA*>(a)) : diploma_gpu::freeze(a);
    }
    std::cout.rdbuf(keep);
    for (uint8_t b : blob) std::printf("%02x", b);
    std::printf("\n");
    return 0;
}
