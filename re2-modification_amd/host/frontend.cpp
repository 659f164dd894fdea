// Regex front-end of the host mirror: parser, regex tree, binary tree and the three automaton
// constructions.  Restates (does not copy) the reference's
//     regex/parser.cpp:8-210      parse_regexp and its stack reductions
//     regex/regex.cpp:162-343     to_string, to_binary_tree, compile
//     regex/reverse.cpp:30-57     reversal (memory-less part)
//     bt/binary_tree.cpp:7-60     epsilonProducing, substr, is_one_unambiguity
//     bt/bt_glushkov.cpp:8-142    linearize / FIRST / LAST / FOLLOW / toGlushkov
//     bt/bt_thomson.cpp:8-60      toThomson
//     bt/bt_mfa.cpp:7-143         toMFA
// The automata must come out IDENTICAL to the reference's, edge order included, because edge
// order decides tie-breaks in the match ("front-end pointer-order dependence").
// Wherever the reference sorts or merges by raw pointer, the code below sorts or merges by
// allocation sequence number (`seq`), and objects are created in the reference's order.
// tests/test_frontend.py checks image equality against the reference's dumps.
#include <algorithm>
#include <cstdio>
#include <iostream>
#include <stdexcept>

#include "diploma_api.h"

namespace diploma {
uint64_t next_seq() {
    static uint64_t counter = 0;
    return ++counter;
}
}  // namespace diploma

namespace {

bool by_seq_edge(const Edge* a, const Edge* b) { return a->seq < b->seq; }
bool by_seq_medge(const MemoryEdge* a, const MemoryEdge* b) { return a->seq < b->seq; }
bool by_seq_node(const Node* a, const Node* b) { return a->seq < b->seq; }
bool by_seq_mnode(const MemoryNode* a, const MemoryNode* b) { return a->seq < b->seq; }

bool is_marker(const Regexp* r, RegexpType t) { return r->regexp_type == t; }

}  // namespace

// =====================================================================================================
// parser (reference regex/parser.cpp).  One pass over the text with the root's child list used
// as an operand stack; brackets push marker nodes, closers reduce back to the marker.
// =====================================================================================================

std::string substr(std::string s, int max_chars) {          // bt/binary_tree.cpp:29-53: first UTF-8 characters
    int chars = 0;
    size_t bytes = 0;
    for (; bytes < s.size() && s[bytes]; bytes++) {
        if ((static_cast<unsigned char>(s[bytes]) & 0xc0) != 0x80) chars++;
        if (chars > max_chars) break;
    }
    return s.substr(0, bytes);
}

Regexp* Regexp::parse_regexp(string& s) {
    Regexp* root = new Regexp(rootExpr);
    root->regexp_str = s;
    auto& stack = root->sub_regexps;
    while (!s.empty()) {
        const char c = s[0];
        switch (c) {
            case '*': case '+': root->wrap_kleene(c); break;
            case '(': stack.push_back(new Regexp(leftParenthesis)); break;
            case '|': root->close_alternative(); break;
            case ')': root->close_group(); break;
            case '[': stack.push_back(new Regexp(leftSquareBr)); break;
            case ']': root->close_enumeration(); break;
            case '-': stack.push_back(new Regexp(dash)); break;
            case '{': stack.push_back(new Regexp(leftBrace)); break;
            case '}': {
                s.erase(0, 1);
                if (substr(s, 1) != ":") printf("Expected :");                     // parser.cpp:44-47
                s.erase(0, 1);
                root->close_backreference(substr(s, 1));
                break;
            }
            case '&': {
                s.erase(0, 1);
                Regexp* ref = new Regexp(reference);
                ref->variable = substr(s, 1);
                stack.push_back(ref);
                root->have_backreference = true;
                break;
            }
            default:
                if ((c >= 'a' && c <= 'z') || c == '.') {
                    Regexp* lit = new Regexp(literal);
                    lit->rune = c;
                    stack.push_back(lit);
                } else {
                    printf("Unexpected literal %c", c);                          // parser.cpp:65-67
                }
        }
        s.erase(0, 1);
    }
    if (stack.size() == 1 && stack.front()->regexp_type == literal) {
        root->rune = stack.back()->rune;
        root->regexp_type = literal;
    } else if (stack.size() > 1) {
        root->regexp_type = concatenationExpr;
    }
    if (root->regexp_type == rootExpr && stack.size() == 1) {
        Regexp* only = stack.back();
        only->regexp_str = root->regexp_str;
        only->have_backreference = root->have_backreference;
        return only;
    }
    return root;
}

// '|': everything since the last '(' or '|' becomes one alternative (parser.cpp:87-104)
void Regexp::close_alternative() {
    Regexp* seqn = new Regexp();
    while (!sub_regexps.empty() && !is_marker(sub_regexps.back(), leftParenthesis) &&
           !is_marker(sub_regexps.back(), alternation)) {
        seqn->sub_regexps.push_front(sub_regexps.back());
        sub_regexps.pop_back();
    }
    if (seqn->sub_regexps.size() == 1) seqn = seqn->sub_regexps.back();
    else seqn->regexp_type = concatenationExpr;
    sub_regexps.push_back(seqn);
    sub_regexps.push_back(new Regexp(alternation));
}

// ')': reduce to the matching '(' -- a concatenation, or an alternation of concatenations (parser.cpp:106-141)
void Regexp::close_group() {
    Regexp* seqn = new Regexp(concatenationExpr);
    Regexp* alts = new Regexp(alternationExpr);
    auto finish_alternative = [&]() {
        if (seqn->sub_regexps.size() == 1) seqn = seqn->sub_regexps.back();
        alts->sub_regexps.push_front(seqn);
    };
    while (!sub_regexps.empty() && !is_marker(sub_regexps.back(), leftParenthesis)) {
        Regexp* top = sub_regexps.back();
        sub_regexps.pop_back();
        if (!is_marker(top, alternation)) {
            seqn->sub_regexps.push_front(top);
        } else {
            finish_alternative();
            seqn = new Regexp(concatenationExpr);
        }
    }
    if (sub_regexps.empty()) throw std::runtime_error("parse_regexp: ')' without '('");
    sub_regexps.pop_back();
    if (alts->sub_regexps.empty()) {
        if (seqn->sub_regexps.size() == 1) seqn = seqn->sub_regexps.back();
        sub_regexps.push_back(seqn);
    } else {
        finish_alternative();
        sub_regexps.push_back(alts);
    }
}

void Regexp::wrap_kleene(char c) {                                               // parser.cpp:143-153
    if (sub_regexps.empty()) throw std::runtime_error("parse_regexp: nothing to repeat");
    Regexp* body = sub_regexps.back();
    sub_regexps.pop_back();
    Regexp* k = new Regexp(c == '*' ? kleeneStar : kleenePlus);
    k->sub_regexp = body;
    sub_regexps.push_back(k);
}

void Regexp::close_backreference(string name) {                                  // parser.cpp:155-178
    Regexp* br = new Regexp(backreferenceExpr);
    br->variable = std::move(name);
    while (!sub_regexps.empty() && !is_marker(sub_regexps.back(), leftBrace)) {
        br->sub_regexps.push_front(sub_regexps.back());
        sub_regexps.pop_back();
    }
    if (sub_regexps.empty()) throw std::runtime_error("parse_regexp: '}' without '{'");
    sub_regexps.pop_back();
    br->sub_regexps.push_front(new Regexp(leftParenthesis));      // reuse the ')' reduction for the body
    br->close_group();
    if (br->sub_regexps.size() == 1) {
        br->sub_regexp = br->sub_regexps.back();
        br->sub_regexps.clear();
    } else {
        printf("PARSER: Backreference can not have more than one subregex");
    }
    sub_regexps.push_back(br);
    have_backreference = true;
}

void Regexp::close_enumeration() {                                               // parser.cpp:180-210
    Regexp* alts = new Regexp(alternationExpr);
    while (!sub_regexps.empty() && !is_marker(sub_regexps.back(), leftSquareBr)) {
        Regexp* top = sub_regexps.back();
        if (top->regexp_type == literal) {
            alts->sub_regexps.push_front(top);
        } else if (top->regexp_type == dash) {
            Regexp* hi = alts->sub_regexps.front();
            alts->sub_regexps.pop_front();
            sub_regexps.pop_back();
            Regexp* lo = sub_regexps.back();
            for (char ch = lo->rune; ch <= hi->rune; ch = char(int(ch) + 1)) {
                Regexp* lit = new Regexp(literal);
                lit->rune = ch;
                alts->sub_regexps.push_front(lit);            // the range ends up in DESCENDING order
            }
        } else {
            printf("PARSER: Expected dash or literal inside enumeration");
        }
        sub_regexps.pop_back();
    }
    if (sub_regexps.empty()) throw std::runtime_error("parse_regexp: ']' without '['");
    sub_regexps.pop_back();
    if (alts->sub_regexps.size() == 1) alts = alts->sub_regexps.back();
    sub_regexps.push_back(alts);
}

// =====================================================================================================
// regex tree
// =====================================================================================================

string Regexp::to_string() {                                                     // regex.cpp:162-207
    switch (regexp_type) {
        case epsilon: return "\xce\xb5";
        case literal: return string(1, rune);
        case reference: return "&" + variable;
        case concatenationExpr: {
            string out;
            for (Regexp* r : sub_regexps) out += r->to_string();
            return out;
        }
        case alternationExpr: {
            if (sub_regexps.empty()) return "(\xce\xb5)";
            string out = "(";
            for (Regexp* r : sub_regexps) out += r->to_string() + "|";
            out.back() = ')';
            return out;
        }
        case backreferenceExpr: return "{" + sub_regexp->to_string() + "}:" + variable;
        case kleeneStar: {
            RegexpType t = sub_regexp->regexp_type;
            bool bare = t == alternationExpr || t == literal || t == epsilon || t == kleeneStar;
            return bare ? sub_regexp->to_string() + "*" : "(" + sub_regexp->to_string() + ")*";
        }
        case kleenePlus:
            return sub_regexp->regexp_type == alternationExpr ? sub_regexp->to_string() + "+"
                                                              : "(" + sub_regexp->to_string() + ")+";
        default: return "";
    }
}

BinaryTree* Regexp::to_binary_tree() {                                           // regex.cpp:223-264
    BinaryTree* t = new BinaryTree(regexp_type);
    switch (regexp_type) {
        case epsilon: break;
        case literal: t->rune = rune; break;
        case reference: t->variable = variable; break;
        case kleeneStar: case kleenePlus: t->child = sub_regexp->to_binary_tree(); break;
        case backreferenceExpr:
            t->child = sub_regexp->to_binary_tree();
            t->variable = variable;
            break;
        default: {                                        // n-ary alternation / concatenation -> right-nested pairs
            if (sub_regexps.empty()) throw std::runtime_error("to_binary_tree: empty operator node");
            if (sub_regexps.size() == 1) return sub_regexps.front()->to_binary_tree();
            t->left = sub_regexps.front()->to_binary_tree();
            if (sub_regexps.size() == 2) {
                t->right = sub_regexps.back()->to_binary_tree();
            } else {
                Regexp* rest = new Regexp(regexp_type);
                rest->sub_regexps.assign(std::next(sub_regexps.begin()), sub_regexps.end());
                t->right = rest->to_binary_tree();
            }
        }
    }
    return t;
}

Automata* Regexp::compile(bool& is_mfa, bool use_reverse, bool use_bnf, bool use_ssnf, bool use_log) {   // regex.cpp:266-343
    is_backref_correct();
    BinaryTree* bt = to_binary_tree();
    if (!maybe_initialized.empty() || !maybe_read.empty()) {
        cout << "\xd0\x98\xd1\x81\xd0\xbf\xd0\xbe\xd0\xbb\xd1\x8c\xd0\xb7\xd1\x83\xd0\xb5\xd1\x82\xd1\x81\xd1\x8f "
                "\xd0\xbf\xd0\xb0\xd0\xbc\xd1\x8f\xd1\x82\xd1\x8c" << endl;
        is_mfa = true;
        const bool one_unamb = bt->is_one_unambiguity();
        // use_ssnf on these paths: the reference computes toSSNF() and throws the result away (regex.cpp:284,292,309)
        if (!one_unamb && use_reverse) {
            Regexp* normal = bnf(use_log);
            if (normal->is_bad_bnf) {                                // not reversible in this version: the plain automaton
                MFA* m = bt->toMFA();
                m->draw("mfa");
                return m;
            }
            cout << "BNF: " << normal->to_string() << endl;
            Regexp* back = normal->reverse();
            cout << "Reverse: " << back->to_string() << endl;
            MFA* m = back->to_binary_tree()->toMFA();
            m->is_reversed = true;
            m->draw("reverse_mfa");
            return m;
        }
        if (one_unamb) {
            cout << "1-\xd0\xbe\xd0\xb4\xd0\xbd\xd0\xbe\xd0\xb7\xd0\xbd\xd0\xb0\xd1\x87\xd0\xbd\xd0\xbe\xd1\x81\xd1\x82\xd1\x8c" << endl;
            is_one_unamb = true;
        }
        if (use_bnf) bt = bnf(use_log)->to_binary_tree();
        MFA* m = bt->toMFA();
        m->draw("mfa");
        return m;
    }
    is_mfa = false;
    cout << "\xd0\x91\xd0\xb5\xd0\xb7 \xd0\xb8\xd1\x81\xd0\xbf\xd0\xbe\xd0\xbb\xd1\x8c\xd0\xb7\xd0\xbe\xd0\xb2\xd0\xb0\xd0\xbd\xd0\xb8\xd1\x8f "
            "\xd0\xbf\xd0\xb0\xd0\xbc\xd1\x8f\xd1\x82\xd0\xb8" << endl;
    if (bt->is_one_unambiguity()) {
        cout << "1-\xd0\xbe\xd0\xb4\xd0\xbd\xd0\xbe\xd0\xb7\xd0\xbd\xd0\xb0\xd1\x87\xd0\xbd\xd0\xbe\xd1\x81\xd1\x82\xd1\x8c" << endl;
        return bt->toGlushkov();                                     // (toSSNF result discarded, regex.cpp:320-321)
    }
    BinaryTree* rbt = reverse()->to_binary_tree();
    if (use_ssnf) rbt = rbt->toSSNF();                              // the one place where the normal form is used (regex.cpp:326-328)
    Automata* rev = rbt->toGlushkov();
    rev->is_reversed = true;
    rev->draw("reverse");
    if (rev->isDeterministic()) return rev;
    return bt->toThomson();
}

// =====================================================================================================
// star normal form (reference bt/bt_ssnf.cpp): iterations of subtrees that produce the empty word are flattened,
// e.g. (a*b*)* -> (a|b)*.  Only the reversed Glushkov automaton of a memory-less regex is built from it.
// =====================================================================================================
namespace {
// a binary node one of whose children vanished is that child's sibling (both vanished: nothing)
BinaryTree* without_empty_child(BinaryTree* t) {
    if (!t->left) return t->right;
    if (!t->right) return t->left;
    return t;
}
BinaryTree* leaf_copy(const BinaryTree* t) {
    BinaryTree* c = new BinaryTree(t->type);
    c->rune = t->rune;
    c->variable = t->variable;
    return c;
}
}  // namespace

BinaryTree* BinaryTree::toSSNF() {
    switch (type) {
        case literal: case reference: return leaf_copy(this);
        case concatenationExpr: case alternationExpr: {
            BinaryTree* t = new BinaryTree(type);
            t->left = left->toSSNF();
            t->right = right->toSSNF();
            return without_empty_child(t);
        }
        case kleeneStar: case kleenePlus: {
            BinaryTree* t = new BinaryTree(type);
            t->child = child->starBody();
            return t;
        }
        default: return this;                                       // epsilon, a named group: as they are
    }
}

BinaryTree* BinaryTree::starBody() {
    switch (type) {
        case epsilon: return nullptr;
        case literal: case reference: return leaf_copy(this);
        case concatenationExpr: {
            const bool le = left->epsilonProducing(), re = right->epsilonProducing();
            BinaryTree* t = new BinaryTree(le && re ? alternationExpr : concatenationExpr);
            if (!le && !re) { t->left = left; t->right = right; return t; }      // cannot be empty: the iteration around it stays as it is
            if (le && re) {                                                       // both may be empty: under the star the order does not matter
                t->left = left->starBody();
                t->right = right->starBody();
                return without_empty_child(t);
            }
            t->left = left->toSSNF();
            t->right = right->toSSNF();
            return t;
        }
        case alternationExpr: {
            BinaryTree* t = new BinaryTree(alternationExpr);
            t->left = left->toSSNF();
            t->right = right->toSSNF();
            return without_empty_child(t);
        }
        case kleeneStar: case kleenePlus: return child->starBody();              // an iteration under an iteration: its body
        case backreferenceExpr: {
            BinaryTree* t = new BinaryTree(backreferenceExpr);                    // (the reference does not carry the name over)
            t->child = child->starBody();
            return t;
        }
        default: return this;
    }
}

// =====================================================================================================
// binary tree: position sets (reference bt/bt_glushkov.cpp) and helpers (bt/binary_tree.cpp)
// =====================================================================================================

bool BinaryTree::epsilonProducing() {
    switch (type) {
        case epsilon: case kleeneStar: return true;
        case literal: case reference: return false;
        case kleenePlus: case backreferenceExpr: return child->epsilonProducing();
        case alternationExpr: return left->epsilonProducing() || right->epsilonProducing();
        case concatenationExpr: return left->epsilonProducing() && right->epsilonProducing();
        default: printf("UNKNOWN BINARY TREE TYPE!!!!!"); return true;
    }
}

// The reference keeps these sets in std::list<string> and combines them with list::merge, i.e. a
// stable merge by string order of two lists that are not necessarily sorted (bt_glushkov.cpp:17-21,
// 36-46,57-67).  std::list::merge is used here too so the element order comes out the same.
list<string> BinaryTree::linearize(int& next_index) {
    list<string> out;
    if (type == literal) {
        name = string(1, rune) + std::to_string(next_index++);
        out.push_back(name);
    } else if (type == alternationExpr || type == concatenationExpr) {
        out.merge(left->linearize(next_index));
        out.merge(right->linearize(next_index));
    } else if (type == kleeneStar || type == kleenePlus) {
        out.merge(child->linearize(next_index));
    }
    return out;
}

list<string> BinaryTree::doFIRST() {
    list<string> out;
    switch (type) {
        case reference: out.push_back(variable); break;
        case literal: out.push_back(name); break;
        case alternationExpr:
            out.merge(left->doFIRST());
            out.merge(right->doFIRST());
            break;
        case concatenationExpr:
            out.merge(left->doFIRST());
            if (left->epsilonProducing()) out.merge(right->doFIRST());
            break;
        case kleeneStar: case kleenePlus: case backreferenceExpr: out.merge(child->doFIRST()); break;
        default: break;
    }
    return out;
}

list<string> BinaryTree::doLAST() {
    list<string> out;
    switch (type) {
        case literal: out.push_back(name); break;
        case alternationExpr:
            out.merge(left->doLAST());
            out.merge(right->doLAST());
            break;
        case concatenationExpr:
            out.merge(right->doLAST());
            if (right->epsilonProducing()) out.merge(left->doLAST());
            break;
        case kleeneStar: out.merge(child->doLAST()); break;       // kleenePlus is not handled by the reference either
        default: break;
    }
    return out;
}

set<pair<string, string>> BinaryTree::doFOLLOW() {
    set<pair<string, string>> out;
    auto cross = [&](const list<string>& from, const list<string>& to) {
        for (const auto& f : from)
            for (const auto& t : to) out.insert({f, t});
    };
    auto absorb = [&](BinaryTree* t) { auto s = t->doFOLLOW(); out.insert(s.begin(), s.end()); };
    if (type == alternationExpr) {
        absorb(left); absorb(right);
    } else if (type == concatenationExpr) {
        cross(left->doLAST(), right->doFIRST());
        absorb(left); absorb(right);
    } else if (type == kleeneStar) {
        cross(child->doLAST(), child->doFIRST());
        absorb(child);
    }
    return out;
}

bool BinaryTree::is_one_unambiguity() {                                          // binary_tree.cpp:55-60
    list<string> first = doFIRST();
    return set<string>(first.begin(), first.end()).size() == first.size();
}

// =====================================================================================================
// Glushkov position automaton (reference bt/bt_glushkov.cpp:107-142)
// =====================================================================================================

Automata* BinaryTree::toGlushkov() {
    Automata* a = new Automata();
    int next_index = 0;
    list<string> positions = linearize(next_index);
    map<string, Node*> node_of;
    for (const string& p : positions) {
        node_of[p] = new Node(p);
        a->nodes.push_back(node_of[p]);
    }
    list<string> first = doFIRST(), last = doLAST();
    set<pair<string, string>> follow = doFOLLOW();
    for (const string& p : first) a->start->edges.push_back(new Edge(substr(p, 1), node_of[p]));
    if (last.size() == 1) {
        a->finish = node_of[last.back()];
    } else {
        for (const string& p : last) node_of[p]->edges.push_back(new Edge("", a->finish));
    }
    for (const auto& pr : follow) node_of[pr.first]->edges.push_back(new Edge(substr(pr.second, 1), node_of[pr.second]));
    return a;
}

// =====================================================================================================
// Thompson construction (reference bt/bt_thomson.cpp:8-60).  The outer automaton's start/finish are
// created BEFORE the operands' (the reference allocates `new Automata()` first), which fixes the
// node order that set<Node*> iteration follows at match time.
// =====================================================================================================

Automata* BinaryTree::toThomson() {
    Automata* a = new Automata();
    auto pair_up = [](Automata* x) { x->finish->finish_for = x->start; x->start->start_for = x->finish; };
    if (type == literal) {
        a->start->edges.push_back(new Edge(string(1, rune), a->finish));
        pair_up(a);
    } else if (type == alternationExpr) {
        Automata* l = left->toThomson();
        Automata* r = right->toThomson();
        a->start->edges.push_back(new Edge("", l->start));
        a->start->edges.push_back(new Edge("", r->start));
        l->finish->edges.push_back(new Edge("", a->finish));
        r->finish->edges.push_back(new Edge("", a->finish));
        a->nodes.merge(l->nodes, by_seq_node);
        a->nodes.merge(r->nodes, by_seq_node);
        pair_up(a);
    } else if (type == concatenationExpr) {
        Automata* l = left->toThomson();
        Automata* r = right->toThomson();
        r->start->finish_for = l->finish->finish_for;
        l->changeFinalState(r->start);
        l->nodes.merge(r->nodes, by_seq_node);
        l->finish = r->finish;
        a = l;
    } else if (type == kleeneStar) {
        Automata* body = child->toThomson();
        body->finish->edges.push_back(new Edge("", body->start));
        body->finish->edges.push_back(new Edge("", a->finish));
        a->nodes.merge(body->nodes, by_seq_node);
        a->start->edges.push_back(new Edge("", body->start));
        a->start->edges.push_back(new Edge("", a->finish));
        pair_up(a);
    } else {
        printf("UNKNOWN BINARY TREE TYPE!!!!!");
    }
    return a;
}

// =====================================================================================================
// Memory automaton (reference bt/bt_mfa.cpp:7-143).  Sub-automata are glued by rewriting the
// edges that enter the left operand's `finish`: each is combined with every start edge of the right
// operand.  All edges into `finish` are epsilon edges, so "combined label" = the start edge's label.
// =====================================================================================================

namespace {

void inherit_actions(MemoryEdge* dst, const MemoryEdge* src) {
    for (const auto& kv : src->memoryActions) dst->memoryActions[kv.first] = kv.second;
}

// drop `old_start` (always the first element) and fold the remaining nodes of b into a (bt_mfa.cpp:51-55,91-94)
void absorb_nodes(MFA* a, MFA* b) {
    b->nodes.erase(b->nodes.begin());
    a->nodes.merge(b->nodes, by_seq_mnode);
    a->changeFinalState(b->finish);
    a->finish = b->finish;
}

}  // namespace

MFA* BinaryTree::toMFA() {
    MFA* a = new MFA();
    switch (type) {
        case epsilon:
            a->start->edges.push_back(new MemoryEdge("", a->finish));
            break;
        case literal: {
            MemoryNode* mid = new MemoryNode();
            a->start->edges.push_back(new MemoryEdge(string(1, rune), mid));
            mid->edges.push_back(new MemoryEdge("", a->finish));
            a->nodes.push_back(mid);
            break;
        }
        case reference: {
            MemoryNode* mid = new MemoryNode();
            MemoryEdge* rd = new MemoryEdge(variable, mid);
            rd->addAction(variable, close);
            a->start->edges.push_back(rd);
            mid->edges.push_back(new MemoryEdge("", a->finish));
            a->nodes.push_back(mid);
            break;
        }
        case backreferenceExpr: {
            a = child->toMFA();
            for (MemoryEdge* e : a->start->edges) e->addAction(variable, open);
            for (MemoryNode* n : a->nodes)
                for (MemoryEdge* e : n->edges)
                    if (e->to == a->finish) e->addAction(variable, close);     // overwrites an `open` set just above
            break;
        }
        case alternationExpr: {
            a = left->toMFA();
            MFA* b = right->toMFA();
            for (MemoryEdge* e : b->start->edges) a->start->edges.push_back(e);
            absorb_nodes(a, b);
            break;
        }
        case concatenationExpr: {
            a = left->toMFA();
            MFA* b = right->toMFA();
            vector<MemoryEdge*> heads(b->start->edges.begin(), b->start->edges.end());
            MemoryEdge* last_head = heads.back();
            for (MemoryNode* n : a->nodes) {
                list<MemoryEdge*> extra;
                for (MemoryEdge* e : n->edges) {
                    if (e->to != a->finish) continue;
                    for (size_t k = 0; k + 1 < heads.size(); k++) {           // one new edge per head but the last ...
                        MemoryEdge* ne = new MemoryEdge(e->by + heads[k]->by, heads[k]->to);
                        inherit_actions(ne, heads[k]);
                        extra.push_back(ne);
                    }
                    e->to = last_head->to;                                   // ... the last head reuses the edge itself
                    e->by = e->by + last_head->by;
                    inherit_actions(e, last_head);
                }
                n->edges.merge(extra, by_seq_medge);
            }
            absorb_nodes(a, b);
            break;
        }
        case kleeneStar: {                                                    // r* = (r+ | epsilon)
            BinaryTree* plus = new BinaryTree(kleenePlus);
            plus->child = child;
            BinaryTree* alt = new BinaryTree(alternationExpr);
            alt->left = plus;
            alt->right = new BinaryTree(epsilon);
            a = alt->toMFA();
            break;
        }
        case kleenePlus: {
            a = child->toMFA();
            vector<MemoryEdge*> heads(a->start->edges.begin(), a->start->edges.end());
            vector<pair<MemoryNode*, MemoryEdge*>> tails;
            for (MemoryNode* n : a->nodes)
                for (MemoryEdge* e : n->edges)
                    if (e->to == a->finish) tails.push_back({n, e});
            // the reference collects (tail, head) pairs in a std::set keyed by the three pointers
            // (bt_mfa.cpp:122-128): iteration order = (node, tail edge, head edge) in allocation order
            struct Loop { MemoryNode* n; MemoryEdge* tail; MemoryEdge* head; };
            vector<Loop> loops;
            for (MemoryEdge* h : heads)
                for (auto& t : tails) loops.push_back({t.first, t.second, h});
            std::sort(loops.begin(), loops.end(), [](const Loop& x, const Loop& y) {
                if (x.n->seq != y.n->seq) return x.n->seq < y.n->seq;
                if (x.tail->seq != y.tail->seq) return x.tail->seq < y.tail->seq;
                return x.head->seq < y.head->seq;
            });
            for (const Loop& l : loops) {
                MemoryEdge* ne = new MemoryEdge(l.tail->by + l.head->by, l.head->to);
                inherit_actions(ne, l.tail);
                inherit_actions(ne, l.head);
                l.n->edges.push_back(ne);
            }
            break;
        }
        default:
            printf("UNKNOWN BINARY TREE TYPE!!!!!");
    }
    return a;
}
