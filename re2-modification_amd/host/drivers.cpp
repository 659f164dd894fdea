// The match loops of the reference's matchers/ directory, batched onto the GPU.
//   match()      matchers/match.cpp:10-32      `./diploma -match`: regex compiled once, then one 0/1 line per
//                                              whitespace-separated token of stdin, until the token `exit`
//   match_gt()   matchers/match_mfa.cpp:13-56  strings from a file against the Glushkov automaton
//   match_mfa()  matchers/match_mfa.cpp:58-97  strings from a file against toMFA()'s automaton
//   run_configuration_examples()  matchers/example_runner.cpp:84-151  `./diploma -match N`: growth curve of one example
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdlib>
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

// ---- `./diploma -match N` ------------------------------------------------------------------------------
// The reference's experiment (matchers/example_runner.cpp:84-151): the regex of test/example_N/regexp.txt (line 1), compiled
// three ways -- plain, `-bnf`, `-reverse` (all with ssnf, which is a no-op on these paths) -- against
// prefix + pumped_string(pump_size) + suffix for growing pump sizes; one "len seconds" line per string and automaton in
// test/example_N/diploma_results.txt, diploma_bnf_results.txt, diploma_reverse_results.txt, each series until a match takes
// >= 0.5 s.  Kept from the reference: the files and their format, the three compile() calls on the one parsed tree in the
// reference's order (they share nodes, so the order is part of the result), the pump-size schedule (500, doubling every
// round and once more whenever the count of rounds the reversed automaton has run is a multiple of ten -- which is every
// round once it has stopped at such a count), the prefix that accumulates the previous string (example_runner.cpp:123
// appends in place), the 0.5 s stop rule.  Different: the time is wall time of one GPU match of one string (copy in, launch,
// copy out) instead of clock() around the CPU loop; a running match is not interrupted, a series ends after its first slow
// one; strings beyond the device limit (16 MiB) end all series.
// DIPLOMA_FRESH_PREFIX=1 uses the file's prefix for every string (matcher.py:58), which is what bench.py measures.
std::string pumped_string(int n, vector<string> pump_v) {
    const int parts = (int)pump_v.size() / 2 + 1, joints = (int)pump_v.size() - parts;
    string unit = pump_v[0];
    while ((int)(unit.size() + pump_v[0].size()) < (n - joints) / parts) unit += pump_v[0];
    string out;
    out.reserve((unit.size() + (joints ? pump_v[1].size() : 0)) * (size_t)(joints + 1));
    for (int k = 0; k < joints; k++) out += unit + pump_v[1];
    return out + unit;
}

void run_configuration_examples(const string& number) {
    const string dir = "test/example_" + number + "/";
    std::ifstream regex_file(dir + "regexp.txt"), pump_file(dir + "pump.txt");
    if (!regex_file.is_open() || !pump_file.is_open()) return;          // like the reference: nothing to do
    string regexp_str, pump_line, suffix, prefix;
    std::getline(regex_file, regexp_str);
    std::getline(pump_file, pump_line);
    std::getline(pump_file, suffix);
    std::getline(pump_file, prefix);
    vector<string> pump;
    for (size_t b = 0;;) {
        const size_t e = pump_line.find(',', b);
        pump.push_back(pump_line.substr(b, e == string::npos ? string::npos : e - b));
        if (e == string::npos) break;
        b = e + 1;
    }
    cout << regexp_str << endl;
    Regexp* regexp = Regexp::parse_regexp(regexp_str);
    regexp->is_backref_correct();                                       // example_runner.cpp:107 (compile() analyses again, like there)
    bool is_mfa = true;
    struct Series { Automata* automata; bool is_mfa; bool stopped; std::ofstream file; };
    Series series[3];
    const char* names[3] = {"diploma_results.txt", "diploma_bnf_results.txt", "diploma_reverse_results.txt"};
    const bool rev[3] = {false, false, true}, bnf[3] = {false, true, true};
    for (int k = 0; k < 3; k++) {
        series[k].automata = regexp->compile(is_mfa, rev[k], bnf[k], true, false);
        series[k].is_mfa = is_mfa;
        series[k].stopped = false;
        series[k].file.open(dir + names[k], std::ofstream::out | std::ofstream::trunc);
    }
    const char* fresh_env = std::getenv("DIPLOMA_FRESH_PREFIX");
    const bool fresh = fresh_env && fresh_env[0] == '1';
    const size_t len_limit = size_t(INT32_MAX / 10), device_limit = 0x00ffffffu;
    long long pump_size = 500;
    int count = 0;
    string grown = prefix;
    size_t len = prefix.size() + (size_t)pump_size + suffix.size();
    while (!(series[0].stopped && series[1].stopped && series[2].stopped) && len < len_limit) {
        const string input = (fresh ? prefix : grown) + pumped_string((int)pump_size, pump) + suffix;
        grown = input;
        len = input.size();
        pump_size += pump_size;
        if (len > device_limit) break;
        for (int k = 0; k < 3; k++) {
            Series& s = series[k];
            if (s.stopped) continue;
            const auto t0 = std::chrono::steady_clock::now();
            if (s.is_mfa) static_cast<MFA*>(s.automata)->match(input); else s.automata->match(input);
            const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (seconds >= 0.5) s.stopped = true;
            if (seconds < 1) s.file << len << " " << seconds << endl;
            if (k == 2) count++;
        }
        if (count % 10 == 0) pump_size *= 2;
    }
}
