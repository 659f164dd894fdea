// Host-side mirror of the reference's C++ API for the match path, backed by the MI355X
// kernels through the C-ABI of include/mfa_hip.h.
//
// Same class names, public fields and call signatures as the reference, so code written
// against it (matchers/match.cpp:10-32, matchers/match_mfa.cpp:13-97,
// matchers/example_runner.cpp:106-111) compiles against this header unchanged:
//
//   reference                              here
//   ---------------------------------------------------------------------------------------
//   variable.h:8-41    Variable            Variable            (kept for source parity; the
//                                                               kernels hold cells as spans)
//   edge.h:13-49       Edge, MemoryEdge,   same                + `seq`
//                      MemoryAction
//   node.h:11-38       Node, MemoryNode    same                + `seq`
//   automata.h:18-84   Automata, MFA       same                match() runs on the GPU;
//                                                              + match_batch(), image_blob()
//   bt/binary_tree.h   BinaryTree          same                toThomson/toGlushkov/toMFA
//   regex/regex.h      Regexp, RegexpType  same                parse_regexp/to_binary_tree/
//                                                              compile/reverse (memory-less)
//
// What is different, and why:
//  * `seq`: the reference orders nodes, edges and states by raw heap pointers
//    (automata.h:12-13, bt/bt_mfa.cpp:51,87,91,122-128, bt/bt_thomson.cpp:26-27).  Under the
//    canonical allocation-order model a pointer comparison is an
//    allocation-sequence comparison; every graph object here carries the sequence number the
//    reference's allocation would have had, and every place the reference compares pointers
//    compares `seq` instead.  Results are then independent of the host heap.
//  * match() does not walk the graph on the CPU.  The graph is frozen into an automaton
//    image (include/mfa_image_format.h) and handed to libmfa_hip.so; there is no CPU fallback:
//    without a usable GPU match() throws std::runtime_error.
//  * The BNF rewriter and the reversal of memory regexes (regex/bnf.cpp, helpers.cpp, reverse.cpp) are restated in
//    bnf_rewrite.cpp.  Two paths of the reference's rewriter read freed or uninitialised memory (a default argument that
//    is an iterator into a destroyed temporary, bnf.cpp:805 with 544-552; an uninitialised iterator, bnf.cpp:409-418):
//    regexes that reach them throw std::runtime_error here instead of repeating an accident.
#ifndef DIPLOMA_API_H
#define DIPLOMA_API_H

#include <cstdint>
#include <list>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

using namespace std;   // the reference's headers do this (automata.h:15, node.h:9, edge.h:8)

struct mfa_image;      // include/mfa_hip.h

namespace diploma {
uint64_t next_seq();   // allocation sequence number (monotonic, process-wide)
}

// ---- variable.h -----------------------------------------------------------------------------------
class Variable {
public:
    bool is_open = false;
    bool is_read = false;
    string value;
    Variable() = default;
    Variable(bool open_, string v, bool read_ = false) : is_open(open_), is_read(read_), value(std::move(v)) {}
    void open() { is_open = true; is_read = false; value.clear(); }
    void close() { is_open = false; }
    string& read() { is_read = true; return value; }
    void write(const string& a) { value += a; }
};

// ---- edge.h / node.h --------------------------------------------------------------------------------
class Node;
class MemoryNode;

class Edge {
public:
    string by;
    Node* to = nullptr;
    bool drawn = false;
    uint64_t seq = diploma::next_seq();
    Edge() = default;
    Edge(string label, Node* target) : by(std::move(label)), to(target) {}
};

enum MemoryAction { open, close };

class MemoryEdge : public Edge {
public:
    map<string, MemoryAction> memoryActions;
    MemoryNode* to = nullptr;
    MemoryEdge() = default;
    MemoryEdge(string label, MemoryNode* target) : to(target) { by = std::move(label); }
    void addAction(const string& var, MemoryAction action) { memoryActions[var] = action; }
};

class Node {
public:
    list<Edge*> edges;
    string name;
    Node* finish_for = nullptr;
    Node* start_for = nullptr;
    uint64_t seq = diploma::next_seq();
    Node() = default;
    explicit Node(string n) : name(std::move(n)) {}
};

class MemoryNode {
public:
    list<MemoryEdge*> edges;
    string name;
    uint64_t seq = diploma::next_seq();
    MemoryNode() = default;
    explicit MemoryNode(string n) : name(std::move(n)) {}
};

// ---- automata.h ---------------------------------------------------------------------------------------
#define Memory map<string, Variable*>
#define MemoryState pair<int, pair<MemoryNode*, Memory>>

class MFA;
vector<vector<bool>> match_mixed(const vector<MFA*>& automata, const vector<vector<string>>& strs);

class Automata {
public:
    Node* start;
    Node* finish;
    int last_idx;
    list<Node*> nodes;
    bool is_reversed = false;

    Automata();
    virtual ~Automata();

    void makeDOTFile(const string& filename);
    bool draw(const string& filename);
    void changeFinalState(Node* new_final);
    bool isDeterministic();

    // reference automata.cpp:177-210, on the GPU (a batch of one)
    bool match(const string& str);
    // the same for many strings in one launch; result[k] is what match(strs[k]) returns
    vector<bool> match_batch(const vector<string>& strs);
    // packed form: string k is bytes[offsets[k], offsets[k+1]); results gets n bytes of 0/1
    void match_packed(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* results);

    // the automaton image handed to the device (include/mfa_image_format.h)
    virtual vector<uint8_t> image_blob() const;
    int device = 0;    // HIP device the match calls run on (the first one, when a batch is spread over several)
    // How many HIP devices a packed batch is spread over: 0 = all the node has, 1 = `device` only.  Strings are independent: the batch is
    // cut into contiguous ranges of about equal BYTES (diploma_partition_by_bytes), one host thread per device copies its range in,
    // matches it through the C-ABI and copies the answers back into place; no exchange between devices.  Small batches (under 4 MiB, or
    // fewer strings than twice the devices) go to `device` alone.  Environment: DIPLOMA_DEVICES=N overrides this field.
    int devices = 0;

protected:
    friend vector<vector<bool>> match_mixed(const vector<MFA*>& automata, const vector<vector<string>>& strs);
    mfa_image* image_for_match();
    mfa_image* cached_image_ = nullptr;
    vector<uint8_t> cached_blob_;
};

class MFA : public Automata {
public:
    MemoryNode* start;
    MemoryNode* finish;
    list<MemoryNode*> nodes;
    bool is_reversed = false;

    MFA();

    void changeFinalState(MemoryNode* new_final);
    void makeDOTFile(const string& filename);
    bool draw(const string& filename);

    // reference mfa.cpp:215-236, on the GPU.  Non-virtual like the reference's: callers that hold an
    // Automata* static_cast to MFA* when compile() set is_mfa (matchers/match.cpp:16-19).
    bool match(string str);
    vector<bool> match_batch(const vector<string>& strs);
    void match_packed(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* results);

    vector<uint8_t> image_blob() const override;
};

// ---- regex/regex.h ---------------------------------------------------------------------------------------
class BinaryTree;

enum RegexpType {
    leftParenthesis, rightParenthesis, leftBrace, rightBrace, leftSquareBr, rightSquareBr, dash,
    kleeneStar, kleenePlus, alternation, literal, epsilon,
    alternationExpr, concatenationExpr, backreferenceExpr, reference, rootExpr
};

class Regexp {
public:
    RegexpType regexp_type = rootExpr;
    string regexp_str;
    char rune = 0;                    // literal
    Regexp* sub_regexp = nullptr;     // kleene, backreference body
    list<Regexp*> sub_regexps;        // alternation, concatenation
    string variable;                  // backreference / reference name
    bool is_read = false;
    Regexp* reference_to = nullptr;
    bool have_backreference = false;
    bool is_one_unamb = false;
    bool is_bad_bnf = false;

    bool is_slided = false;           // an iteration over a read/write block that has been unrolled already (regex.h:97)

    // variable-flow sets (regex.h:99-121), maintained while a tree is analysed or rebuilt
    map<string, list<Regexp*>> initialized;   // cells initialised on every path, with their initialisations in order
    set<string> read;                         // cells read on every path
    set<string> maybe_initialized;            // cells some path may initialise
    set<string> maybe_read;                   // cells some path may read
    set<string> uninited_read;                // reads some path reaches without an initialisation before them
    set<string> unread_init;                  // initialisations some path leaves unread
    set<string> definitely_unread_init;       // initialisations nothing behind them reads (cleared from the result)
    set<string> definitely_uninit_read;       // reads no initialisation precedes (under an iteration)
    set<string> rw_vars;                      // cells that form a read..write block at this level

    Regexp() = default;
    explicit Regexp(RegexpType t) : regexp_type(t) {}

    static Regexp* parse_regexp(string& s);           // consumes s, like the reference (parser.cpp:8)
    string to_string();
    bool is_backref_correct();
    BinaryTree* to_binary_tree();
    Regexp* reverse();                                // regex/reverse.cpp:104-113
    Regexp* bnf(bool is_log = false);                 // regex/bnf.cpp:894-919
    bool is_acreg();                                  // regex/helpers.cpp:4-21
    Automata* compile(bool& is_mfa, bool use_reverse, bool use_bnf, bool use_ssnf, bool use_log = false);

private:
    void close_alternative();                         // '|'
    void close_group();                               // ')'
    void wrap_kleene(char c);
    void close_backreference(string name);
    void close_enumeration();
    Regexp* mirrored();

    // ---- bnf_rewrite.cpp: the reference's regex/helpers.cpp, regex/bnf.cpp, regex/reverse.cpp and the flow analysis of
    //      regex/regex.cpp:91-147 restated (same order of every list and set operation, since the result is order sensitive)
    void analyse(set<string>& seen_inits);
    bool same_shape(Regexp* other);
    Regexp* last_init(const string& var);
    Regexp* prefix_last_init(const string& var, list<Regexp*>::iterator it);
    bool crosses_references();
    map<string, list<string>> reads_inside_inits();
    void put(Regexp* r, bool front);
    void put_sequence(Regexp* r, bool front);
    void put_choice(Regexp* r, bool front);
    void add(Regexp* r, bool front = false);
    void flow_after(Regexp* r);                       // concat_vars
    void flow_beside(Regexp* r);                      // alt_vars
    void flow_under_star(Regexp* r);                  // star_kleene_vars
    void flow_same(Regexp* r);                        // copy_vars
    void flow_replace(Regexp* r);                     // change_vars
    static Regexp* clone(Regexp* r);                  // copy
    Regexp* unwrap_single();                          // simplify_conc_alt
    set<string> choice_free_reads();
    map<string, int> choice_init_counts();
    Regexp* strip_dead_memory(set<string> init_vars, set<string> read_vars);
    Regexp* unroll_plus(set<string> vars, bool strip = true);
    Regexp* unroll(set<string> vars, bool strip = true);
    Regexp* unroll_for_reads(set<string> vars, Regexp* parent, list<Regexp*>::iterator where);
    list<string> order_choice_vars(bool smallest_first);
    Regexp* split_choice_under_star(bool smallest_first = false);
    Regexp* split_choice_on(const string& var);
    Regexp* denest(Regexp* a_alt, Regexp* b_alt);
    Regexp* slide_last_init(set<string> vars);
    Regexp* rw_sequence_under_star(Regexp* parent, bool have_where, list<Regexp*>::iterator where);
    Regexp* rw_choice_under_star();
    Regexp* rw_under_star(Regexp* parent, bool have_where, list<Regexp*>::iterator where);
    Regexp* hoist_choice_out_of_init();
    Regexp* spread_right(int alt_pos);
    Regexp* spread_left(int alt_pos);
    Regexp* spread_both(int alt_pos);
    Regexp* sequence_fix_unread_inits();
    Regexp* sequence_fix_free_reads();
    Regexp* normalise(Regexp* parent, bool under_star, bool have_where, list<Regexp*>::iterator where);
    Regexp* swap_reads_and_inits(set<Regexp*>& done);
    void bind_reads(map<string, Regexp*>& init);
};

// ---- bt/binary_tree.h ---------------------------------------------------------------------------------
class BinaryTree {
public:
    BinaryTree() {}
    explicit BinaryTree(RegexpType t) : type(t) {}

    RegexpType type = epsilon;
    BinaryTree* left = nullptr;     // concatenation and alternation
    BinaryTree* right = nullptr;
    BinaryTree* child = nullptr;    // Kleene, backreference body
    char rune = 0;                  // literal
    string variable;                // reference and backreferenceExpr
    string name;

    bool epsilonProducing();
    list<string> linearize(int&);
    list<string> doFIRST();
    list<string> doLAST();
    set<pair<string, string>> doFOLLOW();
    bool is_one_unambiguity();
    BinaryTree* toSSNF();           // star normal form (bt/bt_ssnf.cpp:73-108)
    BinaryTree* starBody();         // the same for a subtree that sits under an iteration (bt/bt_ssnf.cpp:18-71); nullptr: nothing left

    Automata* toThomson();
    Automata* toGlushkov();
    MFA* toMFA();
};

std::string substr(std::string originalString, int maxLength);

// matchers/match.cpp:10 -- the `-match` loop: compile once, then 0/1 per whitespace-separated token of
// stdin until the token `exit` (or end of input), batched onto the GPU
void match(string regexp_str, bool reverse, bool bnf, bool ssnf, bool use_log = false);
// matchers/match_mfa.cpp:13,58 counterparts: strings from a file (one per line), one batch, results and
// timing on stdout
// The cut points of a batch over `parts` devices: cuts[r] .. cuts[r+1]-1 are the strings of part r (parts + 1 values, cuts[0] = 0,
// cuts[parts] = n), balanced by bytes -- the rule of mfa_amd/sharding.py: part r starts at the first string whose offset is at least
// r / parts of the batch's bytes.  Returns 0, or -1 for bad arguments.
extern "C" int diploma_partition_by_bytes(const uint64_t* offsets, uint64_t n, uint32_t parts, uint64_t* cuts);

// several automata, one batch: strs[k] are matched against automata[k] (memory automata only) by ONE device call
// (mfa_match_mixed: the region pre-pass and the walks of all of them scheduled together); results[k][j] = automata[k] matches strs[k][j]
vector<vector<bool>> match_mixed(const vector<MFA*>& automata, const vector<vector<string>>& strs);
void match_gt(string regexp_str, const string& input_path = "input_strings.txt");
void match_mfa(string regexp_str, const string& input_path = "mfa_str.txt");

// matchers/example_runner.cpp:15 -- the attack-string generator: only pump parts 0 and 1 are used
std::string pumped_string(int n, vector<string> pump_v);
// matchers/example_runner.cpp:84 (`./diploma -match N`): reads test/example_N/regexp.txt and pump.txt below
// the working directory, matches ever longer pumped strings one at a time and writes "len seconds" lines to
// test/example_N/diploma_results.txt until one match takes 0.5 s or longer (see drivers.cpp for what differs)
void run_configuration_examples(const string& number);

#endif  // DIPLOMA_API_H
