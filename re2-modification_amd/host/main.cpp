// `diploma` command line (reference main.cpp:9-87).
//   diploma -match [-bnf] [-reverse] [-ssnf] [-all] [-log]    regex token, then string tokens until `exit`
//   diploma -match N                                          timing series over test/example_N (example_runner.cpp)
//   diploma -dump  [-thompson|-glushkov|-mfa]                 regex token -> automaton image as text
//   diploma -match-file <gt|mfa> <file>                       matchers/match_mfa.cpp counterparts
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "diploma_api.h"

namespace {

string hexlabel(const string& by) {
    if (by.empty() || by == "\xce\xb5") return "-";
    static const char* d = "0123456789abcdef";
    string h;
    for (unsigned char c : by) { h += d[c >> 4]; h += d[c & 15]; }
    return h;
}

template <class NodeT, class EdgeFn>
void dump_graph(const char* kind, bool reversed, const list<NodeT*>& listed, NodeT* start, NodeT* finish, EdgeFn print_extra) {
    vector<NodeT*> all;
    map<NodeT*, size_t> idx;
    auto add = [&](NodeT* n) { if (!idx.count(n)) { idx[n] = all.size(); all.push_back(n); } };
    for (NodeT* n : listed) add(n);
    add(start); add(finish);
    for (size_t k = 0; k < all.size(); k++)
        for (auto* e : all[k]->edges) add(e->to);
    vector<uint64_t> seqs;
    for (NodeT* n : all) seqs.push_back(n->seq);
    std::sort(seqs.begin(), seqs.end());
    cout << "kind " << kind << "\nreversed " << (reversed ? 1 : 0) << "\nnodes " << all.size() << "\nstart " << idx[start]
         << "\nfinish " << idx[finish] << "\n";
    for (size_t k = 0; k < all.size(); k++) {
        size_t rank = std::lower_bound(seqs.begin(), seqs.end(), all[k]->seq) - seqs.begin();
        cout << "node " << k << " " << rank << " " << all[k]->edges.size() << "\n";
        for (auto* e : all[k]->edges) {
            cout << "edge " << hexlabel(e->by) << " " << idx[e->to];
            print_extra(e);
            cout << "\n";
        }
    }
}

int do_dump(int argc, char** argv) {
    string mode = argc > 2 ? argv[2] : "";
    string regex;
    cin >> regex;
    std::ostringstream sink;                       // compile() prints its header lines: keep the dump clean
    std::streambuf* old = cout.rdbuf(sink.rdbuf());
    Regexp* re = Regexp::parse_regexp(regex);
    Automata* nfa = nullptr;
    MFA* mfa = nullptr;
    if (mode == "-thompson") nfa = re->to_binary_tree()->toThomson();
    else if (mode == "-glushkov") nfa = re->to_binary_tree()->toGlushkov();
    else if (mode == "-mfa") { re->is_backref_correct(); mfa = re->to_binary_tree()->toMFA(); }
    else {
        bool is_mfa = false;
        const bool all = mode == "-all";                   // -all = -bnf -reverse -ssnf (main.cpp:25-29)
        Automata* a = re->compile(is_mfa, mode == "-reverse" || all, mode == "-bnf" || mode == "-reverse" || all, mode == "-ssnf" || all);
        if (is_mfa) mfa = static_cast<MFA*>(a); else nfa = a;
    }
    cout.rdbuf(old);
    if (mfa)
        dump_graph("mfa", mfa->is_reversed, mfa->nodes, mfa->start, mfa->finish, [](MemoryEdge* e) {
            for (const auto& kv : e->memoryActions) cout << " " << (kv.second == open ? 'o' : 'c') << kv.first;
        });
    else
        dump_graph("nfa", nfa->is_reversed, nfa->nodes, nfa->start, nfa->finish, [](Edge*) {});
    return 0;
}

}  // namespace

int main(int argc, char* argv[]) {
    try {
        if (argc > 1 && std::strcmp(argv[1], "-dump") == 0) return do_dump(argc, argv);
        if (argc > 2 && std::strcmp(argv[1], "-match-mixed") == 0) {
            // diploma -match-mixed FILE...: every file holds a regex (line 1) and strings (one per line); all of them are matched by
            // ONE device call; prints the 0/1 lines of file 1, then of file 2, ...
            vector<MFA*> automata;
            vector<vector<string>> strs;
            std::ostringstream sink;
            for (int a = 2; a < argc; a++) {
                std::ifstream f(argv[a]);
                if (!f.is_open()) { cout << "ERROR\n"; return 1; }
                string regex, line;
                std::getline(f, regex);
                std::streambuf* old = cout.rdbuf(sink.rdbuf());      // compile() prints its header lines
                Regexp* re = Regexp::parse_regexp(regex);
                re->is_backref_correct();
                automata.push_back(re->to_binary_tree()->toMFA());
                cout.rdbuf(old);
                strs.emplace_back();
                while (std::getline(f, line)) strs.back().push_back(line);
            }
            for (const auto& r : match_mixed(automata, strs))
                for (bool b : r) cout << (b ? 1 : 0) << "\n";
            return 0;
        }
        if (argc > 3 && std::strcmp(argv[1], "-match-file") == 0) {
            string regex;
            cin >> regex;
            if (std::strcmp(argv[2], "gt") == 0) match_gt(regex, argv[3]); else match_mfa(regex, argv[3]);
            return 0;
        }
        if (argc > 1 && std::strcmp(argv[1], "-match") == 0) {
            if (argc > 2 && argv[2][0] != '-') {
                run_configuration_examples(argv[2]);      // main.cpp:11-13: growth curve of test/example_N
                return 0;
            }
            bool bnf = false, reverse = false, ssnf = false, use_log = false;
            set<string> flags;
            for (int i = 2; i < argc; i++) flags.insert(argv[i]);
            if (argc > 2 && std::strcmp(argv[2], "-all") == 0) bnf = reverse = ssnf = true;      // main.cpp:25-29
            if (flags.count("-bnf")) bnf = true;
            if (flags.count("-reverse")) reverse = bnf = true;
            if (flags.count("-ssnf")) ssnf = true;
            if (flags.count("-log")) use_log = true;
            string regex;
            cin >> regex;
            match(regex, reverse, bnf, ssnf, use_log);
            return 0;
        }
        // main.cpp:50-85: no mode flag -> one regex token per round: its backreference normal form and the reversal of that.
        // (The reference never leaves this loop: parse_regexp erases the token before it is compared with "exit", and at end of
        // input it spins.  Here `exit` and end of input end it.)
        const bool use_log = argc > 2 && std::strcmp(argv[2], "-log") == 0;
        string token;
        while (cin >> token && token != "exit") {
            Regexp* regexp = Regexp::parse_regexp(token);
            regexp->is_backref_correct();
            Regexp* normal = regexp->bnf(use_log);
            if (normal->is_bad_bnf) continue;
            cout << "BNF: " << normal->to_string() << endl;
            cout << "Reverse: " << normal->reverse()->to_string() << endl;
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "diploma: " << e.what() << "\n";
        return 1;
    }
}
