// The match loops of the reference's matchers/ directory, batched onto the GPU.
//   match()      matchers/match.cpp:10-32      `./diploma -match`: regex compiled once, then one 0/1 line per
//                                              whitespace-separated token of stdin, until the token `exit`
//   match_gt()   matchers/match_mfa.cpp:13-56  strings from a file against the Glushkov automaton
//   match_mfa()  matchers/match_mfa.cpp:58-97  strings from a file against toMFA()'s automaton
#include <chrono>
#include <fstream>
#include <iostream>

#include "diploma_api.h"

extern "C" int isatty(int);   // <unistd.h> would clash with the reference's `enum MemoryAction { open, close }`

namespace {

struct Batch {
    vector<uint8_t> bytes;
    vector<uint64_t> offsets{0};
    void add(const string& s) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        offsets.push_back(bytes.size());
    }
    size_t size() const { return offsets.size() - 1; }
    void clear() { bytes.clear(); offsets.assign(1, 0); }
};

void flush(Automata* a, MFA* m, Batch& b) {
    if (b.size() == 0) return;
    vector<uint8_t> res(b.size());
    b.bytes.resize(b.bytes.size() + 16);
    if (m) m->match_packed(b.bytes.data(), b.offsets.data(), b.size(), res.data());
    else a->match_packed(b.bytes.data(), b.offsets.data(), b.size(), res.data());
    string out;
    for (uint8_t r : res) { out += r ? '1' : '0'; out += '\n'; }
    cout << out << std::flush;
    b.clear();
}

}  // namespace

void match(string regexp_str, bool reverse, bool bnf, bool ssnf, bool use_log) {
    Regexp* regexp = Regexp::parse_regexp(regexp_str);
    bool is_mfa = false;
    Automata* automata = regexp->compile(is_mfa, reverse, bnf, ssnf, use_log);
    MFA* mfa = is_mfa ? static_cast<MFA*>(automata) : nullptr;
    // The reference answers token by token (and spins forever on end of input without `exit`).  Here
    // tokens are collected into batches -- one token per batch when stdin is a terminal, so interactive
    // use still answers immediately -- and end of input ends the loop.
    const bool interactive = isatty(0) != 0;
    const size_t kFlushBytes = size_t(256) << 20;
    Batch batch;
    string text;
    while (cin >> text && text != "exit") {
        batch.add(text);
        if (interactive || batch.bytes.size() >= kFlushBytes) flush(automata, mfa, batch);
    }
    flush(automata, mfa, batch);
}

namespace {

void match_file(Automata* a, MFA* m, const string& input_path, const string& csv_path, bool print_result) {
    std::ifstream file(input_path);
    if (!file.is_open()) { cout << "ERROR\n"; exit(1); }
    Batch batch;
    string line;
    while (getline(file, line)) batch.add(line);
    vector<uint8_t> res(batch.size() + 1);
    batch.bytes.resize(batch.bytes.size() + 16);
    auto t0 = std::chrono::steady_clock::now();
    if (m) m->match_packed(batch.bytes.data(), batch.offsets.data(), batch.size(), res.data());
    else a->match_packed(batch.bytes.data(), batch.offsets.data(), batch.size(), res.data());
    double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the reference prints one time per string; a batch has one time for all of them
    std::ofstream out(csv_path);
    cout << seconds << endl;
    out << seconds << ",";
    if (print_result)
        for (size_t k = 0; k < batch.size(); k++) cout << (res[k] ? 1 : 0) << endl;
}

}  // namespace

void match_gt(string regexp_str, const string& input_path) {
    Regexp* regexp = Regexp::parse_regexp(regexp_str);
    BinaryTree* bt = regexp->to_binary_tree();
    Automata* glushkov = bt->toGlushkov();
    Automata* thomson = bt->toThomson();
    thomson->draw("thomson");
    glushkov->draw("glushkov");
    match_file(glushkov, nullptr, input_path, "results.txt", false);
}

void match_mfa(string regexp_str, const string& input_path) {
    Regexp* regexp = Regexp::parse_regexp(regexp_str);
    BinaryTree* bt = regexp->to_binary_tree();
    MFA* mfa = bt->toMFA();
    mfa->draw("mfa");
    match_file(nullptr, mfa, input_path, "results7.txt", true);
}
