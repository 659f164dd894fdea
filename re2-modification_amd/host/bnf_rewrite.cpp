// Backreference normal form and reversal of regexes with memory cells.
//
// Restates, for the host mirror of the reference's API (diploma_api.h), what the reference does in
//     regex/regex.cpp:9-147      is_equal, the flow analysis _is_backref_correct, simplify_conc_alt, set helpers
//     regex/helpers.cpp:4-262    is_acreg, cross-reference check, push_* (flattening inserts), *_vars (flow sets), copy
//     regex/bnf.cpp:13-919       the rewriting: distribute, open Kleene, denesting, sliding, clearing, bnf()
//     regex/reverse.cpp:5-113    bind_init_to_read, _reverse, replace_read_write, reverse()
// `./diploma -match -bnf|-reverse` feeds the rewritten tree to toMFA, so the automaton -- and through the match loop's
// tie-breaks the 0/1 answers -- depend on the exact shape of the result: every list is built in the reference's order,
// every set operation is applied in the reference's sequence, and the places where the reference's outcome rests on a
// quirk are kept and marked "(sic)".  Checked against the reference itself: tests/golden/front/bnf_reverse.txt, the 20
// `-bnf`/`-reverse` automaton images under tests/golden/images, and tests/test_frontend_fuzz.py (random regexes).
//
// Not repeated: two paths on which the reference reads memory it does not own (see diploma_api.h); they throw.
#include <algorithm>
#include <fstream>
#include <iostream>
#include <iterator>
#include <queue>
#include <stdexcept>

#include "diploma_api.h"

namespace {

std::ofstream g_trace;          // the `-log` trace, log.txt in the working directory (reference: bnf.cpp:10, 894-897)

void unite(set<string>& a, const set<string>& b) { a.insert(b.begin(), b.end()); }

void meet(set<string>& a, const set<string>& b) {
    for (auto it = a.begin(); it != a.end();) it = b.count(*it) ? std::next(it) : a.erase(it);
}

set<string> common(const set<string>& a, const set<string>& b) {
    set<string> out(a);
    meet(out, b);
    return out;
}

// regex.cpp:77-89: per cell, the second map's initialisations go behind the first's
void append_inits(map<string, list<Regexp*>>& first, const map<string, list<Regexp*>>& second) {
    for (const auto& el : second) {
        auto& dst = first[el.first];
        dst.insert(dst.end(), el.second.begin(), el.second.end());
    }
}

bool is_seq(const Regexp* r) { return r->regexp_type == concatenationExpr; }
bool is_alt(const Regexp* r) { return r->regexp_type == alternationExpr; }
bool is_star(const Regexp* r) { return r->regexp_type == kleeneStar; }
bool is_iter(const Regexp* r) { return r->regexp_type == kleeneStar || r->regexp_type == kleenePlus; }
bool is_init(const Regexp* r) { return r->regexp_type == backreferenceExpr; }

void mark_init(Regexp* r) {            // what every node that initialises r->variable carries (regex.cpp:139-144 and elsewhere)
    r->initialized[r->variable].push_back(r);
    r->maybe_initialized.insert(r->variable);
    r->unread_init.insert(r->variable);
    r->definitely_unread_init.insert(r->variable);
}

}  // namespace

// =====================================================================================================
// flow sets
// =====================================================================================================

void Regexp::flow_after(Regexp* r) {                                     // helpers.cpp:118-165 concat_vars
    if (r->is_bad_bnf) is_bad_bnf = true;
    if (sub_regexps.empty()) { flow_same(r); return; }
    for (const string& v : r->maybe_initialized)
        if (definitely_uninit_read.count(v)) rw_vars.insert(v);
    set<string> free_reads(r->uninited_read);
    for (const auto& known : initialized) free_reads.erase(known.first);
    unite(uninited_read, free_reads);
    unite(unread_init, r->unread_init);
    for (const string& v : r->read) unread_init.erase(v);
    set<string> surely_free(r->uninited_read);
    for (const string& v : maybe_initialized) surely_free.erase(v);
    unite(definitely_uninit_read, surely_free);
    for (const string& v : r->maybe_read) definitely_unread_init.erase(v);
    unite(definitely_unread_init, r->definitely_unread_init);
    append_inits(initialized, r->initialized);
    unite(read, r->read);
    unite(maybe_read, r->maybe_read);
    unite(maybe_initialized, r->maybe_initialized);
}

void Regexp::flow_beside(Regexp* r) {                                    // helpers.cpp:167-178 alt_vars
    if (r->is_bad_bnf) is_bad_bnf = true;
    unite(uninited_read, r->uninited_read);
    unite(unread_init, r->unread_init);
    unite(definitely_unread_init, r->definitely_unread_init);
    unite(definitely_uninit_read, r->definitely_uninit_read);
    meet(read, r->read);
    unite(maybe_read, r->maybe_read);
    unite(maybe_initialized, r->maybe_initialized);
}

void Regexp::flow_under_star(Regexp* r) {                                // helpers.cpp:180-189 star_kleene_vars
    if (r->is_bad_bnf) is_bad_bnf = true;
    maybe_initialized = r->maybe_initialized;
    maybe_read = r->maybe_read;
    uninited_read = r->uninited_read;
    unread_init = r->unread_init;
    definitely_unread_init = r->definitely_unread_init;
    definitely_uninit_read = r->definitely_uninit_read;
}

void Regexp::flow_same(Regexp* r) {                                      // helpers.cpp:191-206 copy_vars
    if (r->is_bad_bnf) is_bad_bnf = true;
    if (regexp_type == reference && r->regexp_type == reference) reference_to = r->reference_to;
    unite(uninited_read, r->uninited_read);
    unite(rw_vars, r->rw_vars);
    unite(definitely_unread_init, r->definitely_unread_init);
    unite(unread_init, r->unread_init);
    if (regexp_type != alternationExpr) append_inits(initialized, r->initialized);
    unite(read, r->read);
    unite(maybe_read, r->maybe_read);
    unite(maybe_initialized, r->maybe_initialized);
    unite(definitely_uninit_read, r->definitely_uninit_read);
}

void Regexp::flow_replace(Regexp* r) {                                   // helpers.cpp:208-212 change_vars
    initialized = r->initialized;
    read = r->read;
    flow_under_star(r);
}

// regex.cpp:91-147.  Additive like the reference's: analysing a tree twice doubles its initialisation lists.
void Regexp::analyse(set<string>& seen_inits) {
    switch (regexp_type) {
        case reference:
            read.insert(variable); maybe_read.insert(variable);
            uninited_read.insert(variable); definitely_uninit_read.insert(variable);
            break;
        case alternationExpr: case concatenationExpr: {
            int k = 0;
            for (Regexp* sub : sub_regexps) {
                set<string> sub_seen(seen_inits);
                sub->analyse(sub_seen);
                seen_inits = sub_seen;
                if (is_seq(this)) flow_after(sub);          // the list is never empty here: also the first child takes the general path
                else if (k == 0) flow_same(sub);
                else flow_beside(sub);
                k++;
            }
            break;
        }
        case kleeneStar: case kleenePlus: case backreferenceExpr:
            sub_regexp->analyse(seen_inits);
            if (is_star(this)) flow_under_star(sub_regexp);
            else flow_same(sub_regexp);
            if (is_init(this)) mark_init(this);
            break;
        default: break;
    }
}

bool Regexp::is_backref_correct() {                                     // regex.cpp:209-221 (nothing ever fills the two sets it tests)
    set<string> seen;
    analyse(seen);
    return true;
}

// =====================================================================================================
// small helpers on trees
// =====================================================================================================

// regex.cpp:9-39 is_equal.  (sic) for sequences and alternations the loop walks THIS node's list with both iterators, so
// two of them are "equal" as soon as they have the same number of children.
bool Regexp::same_shape(Regexp* other) {
    if (!other || regexp_type != other->regexp_type) return false;
    switch (regexp_type) {
        case epsilon: return true;
        case literal: return other->rune == rune;
        case reference: return other->variable == variable;
        case concatenationExpr: case alternationExpr: return sub_regexps.size() == other->sub_regexps.size();
        case kleeneStar: case kleenePlus: return sub_regexp->same_shape(other->sub_regexp);
        case backreferenceExpr: return sub_regexp->same_shape(other->sub_regexp) && variable == other->variable;
        default: return false;
    }
}

Regexp* Regexp::last_init(const string& var) {                           // regex.h:125-130
    auto it = initialized.find(var);
    return it == initialized.end() || it->second.empty() ? nullptr : it->second.back();
}

// regex.h:132-140: the last initialisation of `var` in the children from `it` back to the first one
Regexp* Regexp::prefix_last_init(const string& var, list<Regexp*>::iterator it) {
    if (it == sub_regexps.end()) return nullptr;
    for (;;) {
        if (Regexp* found = (*it)->last_init(var)) return found;
        if (it == sub_regexps.begin()) return nullptr;
        --it;
    }
}

Regexp* Regexp::unwrap_single() {                                        // regex.cpp:149-160 simplify_conc_alt
    if ((is_seq(this) || is_alt(this)) && sub_regexps.size() == 1) return sub_regexps.front();
    if ((is_seq(this) || is_alt(this)) && sub_regexps.empty()) return new Regexp(epsilon);
    return this;
}

set<string> Regexp::choice_free_reads() {                                // regex.cpp:41-47
    set<string> out;
    for (Regexp* sub : sub_regexps) unite(out, sub->uninited_read);
    return out;
}

map<string, int> Regexp::choice_init_counts() {                          // regex.cpp:49-62
    map<string, int> out;
    for (Regexp* sub : sub_regexps)
        for (const auto& var : sub->initialized) out[var.first] += 1;
    return out;
}

bool Regexp::is_acreg() {                                                // helpers.cpp:4-21
    switch (regexp_type) {
        case alternationExpr: case concatenationExpr:
            for (Regexp* sub : sub_regexps)
                if (!sub->is_acreg()) return false;
            return true;
        case kleeneStar: case kleenePlus: return sub_regexp->is_acreg();
        case backreferenceExpr: return sub_regexp->is_acreg() && !maybe_read.count(variable);
        default: return true;
    }
}

map<string, list<string>> Regexp::reads_inside_inits() {                 // helpers.cpp:23-36 get_inner_reads
    map<string, list<string>> out;
    if (regexp_type == kleenePlus || is_init(this)) {
        for (auto& inner : sub_regexp->reads_inside_inits()) out[inner.first].merge(inner.second);
        if (is_init(this))
            for (const string& v : maybe_read) out[variable].push_back(v);
    }
    return out;
}

// helpers.cpp:38-65: an initialisation that depends on a read whose cell is initialised again before the first is used
bool Regexp::crosses_references() {
    map<string, list<string>> inner;
    set<string> inits;
    for (Regexp* sub : sub_regexps) {
        for (const string& v : sub->maybe_read) {
            auto hit = inner.find(v);
            if (hit == inner.end()) continue;
            for (const string& dep : hit->second)
                if (inits.count(dep)) return true;
            inner.erase(hit);
        }
        if (!sub->initialized.empty())
            for (auto& more : sub->reads_inside_inits()) inner[more.first].merge(more.second);
        unite(inits, sub->maybe_initialized);
    }
    return false;
}

// ---- flattening inserts (helpers.cpp:68-116) ---------------------------------------------------------
void Regexp::put(Regexp* r, bool front) {
    if (front) sub_regexps.push_front(r);
    else sub_regexps.push_back(r);
}

void Regexp::put_sequence(Regexp* r, bool front) {
    if (!is_seq(this)) { put(r, front); return; }
    if (front)                                          // (sic) the children go in twice when inserted at the front
        for (auto it = r->sub_regexps.rbegin(); it != r->sub_regexps.rend(); ++it) put(*it, front);
    for (Regexp* inner : r->sub_regexps) put(inner, front);
}

void Regexp::put_choice(Regexp* r, bool front) {
    if (is_alt(this))
        for (Regexp* inner : r->sub_regexps) put(inner, false);
    else put(r, front);
}

void Regexp::add(Regexp* r, bool front) {
    if (is_seq(this)) flow_after(r);
    else if (is_alt(this)) {
        if (sub_regexps.empty()) flow_same(r);
        else flow_beside(r);
    }
    if (is_seq(r)) put_sequence(r, front);
    else if (is_alt(r)) put_choice(r, front);
    else put(r, front);
}

// helpers.cpp:214-262 copy: every initialisation of the result is a node of its own
Regexp* Regexp::clone(Regexp* r) {
    switch (r->regexp_type) {
        case reference: {
            Regexp* out = new Regexp(reference);
            out->variable = r->variable;
            out->reference_to = r->reference_to;
            out->flow_same(r);
            return out;
        }
        case backreferenceExpr: {
            Regexp* out = new Regexp(backreferenceExpr);
            out->variable = r->variable;
            out->sub_regexp = clone(r->sub_regexp);
            out->flow_same(out->sub_regexp);
            out->definitely_unread_init.insert(r->variable);
            out->unread_init.insert(r->variable);
            out->maybe_initialized.insert(r->variable);
            out->initialized[r->variable].push_back(out);
            return out;
        }
        case kleeneStar: case kleenePlus: {
            Regexp* out = new Regexp(r->regexp_type);               // (sic) no flow sets on the copy of an iteration
            out->sub_regexp = clone(r->sub_regexp);
            return out;
        }
        case alternationExpr: case concatenationExpr: {
            Regexp* out = new Regexp(r->regexp_type);
            int k = 0;
            for (Regexp* sub : r->sub_regexps) {
                Regexp* c = clone(sub);
                if (is_seq(r)) out->flow_after(c);
                else if (k == 0) out->flow_same(c);
                else out->flow_beside(c);
                out->sub_regexps.push_back(c);
                k++;
            }
            return out;
        }
        default: return r;                                         // epsilon and literals are shared
    }
}

// =====================================================================================================
// binding reads to initialisations, clearing what is never used (reverse.cpp:5-28, bnf.cpp:39-127)
// =====================================================================================================

void Regexp::bind_reads(map<string, Regexp*>& init) {
    switch (regexp_type) {
        case reference: {
            auto it = init.find(variable);
            if (it != init.end()) { reference_to = it->second; reference_to->is_read = true; }
            break;
        }
        case concatenationExpr: case alternationExpr:
            for (Regexp* sub : sub_regexps) {
                sub->bind_reads(init);
                if (is_seq(this))
                    for (const auto& el : sub->initialized) init[el.first] = el.second.back();      // the last one before a possible read
            }
            break;
        case kleeneStar: case kleenePlus: case backreferenceExpr: sub_regexp->bind_reads(init); break;
        default: break;
    }
}

Regexp* Regexp::strip_dead_memory(set<string> init_vars, set<string> read_vars) {
    Regexp* out = new Regexp();
    switch (regexp_type) {
        case epsilon: out->regexp_type = epsilon; break;
        case literal: out->regexp_type = literal; out->rune = rune; break;
        case reference:
            if (!reference_to && read_vars.count(variable)) out->regexp_type = epsilon;       // a read nothing can have initialised
            else { out->regexp_type = reference; out->reference_to = reference_to; out->variable = variable; }
            break;
        case backreferenceExpr:
            if (is_read) {                                           // stays an initialisation; only its body is cleaned
                sub_regexp = sub_regexp->strip_dead_memory(init_vars, read_vars);
                return this;
            }
            // never read: the node turns into its body
            out->regexp_type = sub_regexp->regexp_type;
            out->flow_same(this);
            out->initialized.erase(variable);
            out->maybe_initialized.erase(variable);
            out->unread_init.erase(variable);
            out->definitely_unread_init.erase(variable);
            if (is_alt(out) || is_seq(out)) out->sub_regexps = sub_regexp->sub_regexps;
            else if (is_iter(out)) out->sub_regexp = sub_regexp->sub_regexp;
            else if (out->regexp_type == reference) { out->variable = sub_regexp->variable; out->reference_to = sub_regexp->reference_to; }
            else if (is_init(out)) throw std::runtime_error("bnf: an unread initialisation directly inside an unread initialisation "
                                                            "(the reference dereferences a null pointer here, bnf.cpp:71-86)");
            else out->rune = sub_regexp->rune;
            out = out->strip_dead_memory(init_vars, read_vars);
            break;
        case concatenationExpr: case alternationExpr:
            out->regexp_type = regexp_type;
            for (Regexp* sub : sub_regexps) {
                if (is_seq(this)) {
                    Regexp* cleaned = sub->strip_dead_memory(init_vars, read_vars);
                    if (cleaned->regexp_type != epsilon) out->add(cleaned);
                } else {
                    Regexp* cleaned = sub->strip_dead_memory(common(init_vars, sub->definitely_unread_init), common(read_vars, sub->uninited_read));
                    out->sub_regexps.push_back(cleaned);
                    out->flow_beside(cleaned);
                }
            }
            if (is_seq(out) && out->sub_regexps.empty()) out->sub_regexps.push_back(new Regexp(epsilon));
            break;
        case kleeneStar: case kleenePlus:
            out->regexp_type = regexp_type;
            out->sub_regexp = sub_regexp->strip_dead_memory(init_vars, read_vars);
            if (is_star(this)) out->flow_under_star(out->sub_regexp);
            else out->flow_same(out->sub_regexp);
            break;
        default: break;
    }
    return out;
}

// =====================================================================================================
// the rewriting steps
// =====================================================================================================

Regexp* Regexp::hoist_choice_out_of_init() {                             // bnf.cpp:13-37: {(a|b)}:1 == ({a}:1|{b}:1)
    if (!(is_init(this) && is_alt(sub_regexp))) return this;
    Regexp* out = new Regexp(alternationExpr);
    for (Regexp* branch : sub_regexp->sub_regexps) {
        Regexp* one = new Regexp(backreferenceExpr);
        one->variable = variable;
        one->sub_regexp = branch;
        one->flow_same(branch);
        mark_init(one);
        out->sub_regexps.push_back(one);
    }
    out->flow_same(this);
    if (g_trace.is_open()) g_trace << to_string() << "->" << out->to_string() << endl;
    return out;
}

Regexp* Regexp::spread_right(int alt_pos) {                              // bnf.cpp:130-188: (a|b)c -> (ac|bc)
    if (!is_seq(this)) return this;
    const size_t n = sub_regexps.size();
    if (!(n - 1 > (size_t)alt_pos)) return this;
    Regexp* out = new Regexp(concatenationExpr);
    auto it = sub_regexps.begin();
    for (int k = 0; k < alt_pos; k++) out->add(*it++);
    Regexp* choice = *it++;
    if (it == sub_regexps.end()) return this;
    Regexp* tail = new Regexp(concatenationExpr);
    for (size_t k = (size_t)alt_pos + 1; k < n; k++) tail->add(*it++);
    Regexp* spread = new Regexp(alternationExpr);
    for (Regexp* branch : choice->sub_regexps) {
        Regexp* one = new Regexp(concatenationExpr);
        one->add(branch);
        for (Regexp* t : tail->sub_regexps) one->add(clone(t));
        spread->add(one);
    }
    out->add(spread);
    if (g_trace.is_open()) g_trace << "distribute to right" << endl << to_string() << " -> " << spread->to_string() << endl;
    (void)spread->normalise(nullptr, false, false, {});                 // (sic) the normalised form is dropped; its side effects on shared nodes are not
    if (out->sub_regexps.size() == 1) {
        Regexp* only = out->sub_regexps.front();
        out->regexp_type = alternationExpr;
        out->flow_replace(only);
        out->sub_regexps = list<Regexp*>(only->sub_regexps);
    }
    return out;
}

Regexp* Regexp::spread_left(int alt_pos) {                               // bnf.cpp:190-246: a(b|c) -> (ab|ac)
    if (!is_seq(this)) { cout << "Expected concatenation in distribute()" << endl; return this; }
    const size_t n = sub_regexps.size();
    if (alt_pos <= 0) return this;
    Regexp* out = new Regexp(concatenationExpr);
    Regexp* head = new Regexp(concatenationExpr);
    auto it = sub_regexps.begin();
    for (int k = 0; k < alt_pos; k++) head->add(*it++);
    Regexp* choice = *it;
    Regexp* spread = new Regexp(alternationExpr);
    for (Regexp* branch : choice->sub_regexps) {
        Regexp* one = new Regexp(concatenationExpr);
        for (Regexp* h : head->sub_regexps) one->add(clone(h));
        one->add(branch);
        spread->add(one);
    }
    if (g_trace.is_open()) g_trace << "distribute to left" << endl << "prefix -> " << spread->to_string() << endl;
    out->add(spread->normalise(nullptr, false, false, {}));
    ++it;
    for (size_t k = (size_t)alt_pos + 1; k < n; k++) out->add(*it++);
    if (out->sub_regexps.size() == 1) {
        Regexp* only = out->sub_regexps.front();
        out->regexp_type = alternationExpr;
        out->flow_replace(only);
        out->sub_regexps = list<Regexp*>(only->sub_regexps);
    }
    return out;
}

Regexp* Regexp::spread_both(int alt_pos) {                               // bnf.cpp:248-258
    if (alt_pos == 0) return spread_right(0);
    if ((size_t)alt_pos == sub_regexps.size() - 1) return spread_left(alt_pos);
    Regexp* out = spread_right(alt_pos);
    return out->spread_left((int)out->sub_regexps.size() - 1);
}

Regexp* Regexp::unroll_plus(set<string> vars, bool strip) {              // bnf.cpp:260-273: a+ == a* a (here: this, then a copy of the body)
    Regexp* iteration = this;
    if (strip) {
        map<string, Regexp*> none;
        bind_reads(none);
        iteration = strip_dead_memory(std::move(vars), {});
    }
    Regexp* body = clone(sub_regexp);
    Regexp* out = new Regexp(concatenationExpr);
    out->add(iteration);
    out->add(body);
    return out;
}

Regexp* Regexp::unroll(set<string> vars, bool strip) {                   // bnf.cpp:275-293: a* == (eps | a* a)
    if (is_star(this)) {
        Regexp* out = new Regexp(alternationExpr);
        out->sub_regexps.push_back(new Regexp(epsilon));
        out->add(unroll_plus(vars, strip));
        if (g_trace.is_open()) g_trace << "open kleene" << endl << to_string() << " -> " << out->to_string() << endl;
        return out;
    }
    if (regexp_type == kleenePlus) return unroll_plus(vars);            // (sic) always with clearing
    cout << "Open kleene: Expected kleene" << endl;
    throw std::runtime_error("bnf: open_kleene on a node that is not an iteration (the reference returns garbage here, bnf.cpp:291)");
}

// bnf.cpp:295-351: (a|b)* -> (a* | a* b a* (b a*)*) where the b branches read one of `vars`
Regexp* Regexp::unroll_for_reads(set<string> vars, Regexp* parent, list<Regexp*>::iterator where) {
    if (!is_alt(sub_regexp)) {
        if (common(sub_regexp->definitely_uninit_read, sub_regexp->maybe_initialized).empty()) return unroll(vars);
        return rw_sequence_under_star(parent, true, where);
    }
    Regexp* quiet = new Regexp(alternationExpr);
    Regexp* reading = new Regexp(alternationExpr);
    for (Regexp* branch : sub_regexp->sub_regexps)
        (common(branch->maybe_read, vars).empty() ? quiet : reading)->sub_regexps.push_back(branch);
    quiet = quiet->unwrap_single();
    reading = reading->unwrap_single();
    Regexp* quiet_star = new Regexp(kleeneStar);
    quiet_star->sub_regexp = quiet;
    quiet_star->flow_under_star(quiet);
    Regexp* step = new Regexp(concatenationExpr);
    step->add(reading);
    step->add(clone(quiet_star));
    Regexp* more = new Regexp(kleeneStar);
    Regexp* step_copy = clone(step);
    more->sub_regexp = step_copy;
    more->flow_under_star(step_copy);
    Regexp* seq = new Regexp(concatenationExpr);
    seq->add(quiet_star);
    seq->add(step);
    seq->add(more);
    Regexp* out = new Regexp(alternationExpr);
    Regexp* alone = clone(quiet_star);
    out->sub_regexps.push_back(alone);
    quiet->flow_same(alone);                                             // (sic) lands on the branch set, not on the result
    out->sub_regexps.push_back(seq);
    out->flow_beside(seq);
    return out;
}

// bnf.cpp:353-393: the cells with free reads among the branches, most often initialised first.  (sic) the reference's two
// comparators are the same, so "min order" and "max order" agree.
list<string> Regexp::order_choice_vars(bool) {
    struct ByCount {
        bool operator()(const pair<string, int>& a, const pair<string, int>& b) const { return a.second < b.second; }
    };
    std::priority_queue<pair<string, int>, vector<pair<string, int>>, ByCount> pq;
    const map<string, int> counts = choice_init_counts();
    const set<string> free_reads = choice_free_reads();
    if (!free_reads.empty())
        for (const auto& c : counts)
            if (free_reads.count(c.first)) pq.emplace(c);
    list<string> out;
    for (; !pq.empty(); pq.pop()) out.push_back(pq.top().first);
    return out;
}

Regexp* Regexp::denest(Regexp* a_alt, Regexp* b_alt) {                   // bnf.cpp:395-421: (a|b)* = a*(b a*)*
    Regexp* a_star = new Regexp(kleeneStar);
    a_star->sub_regexp = a_alt;
    a_star->flow_under_star(a_alt);
    Regexp* second = new Regexp(kleeneStar);
    Regexp* step = new Regexp(concatenationExpr);
    Regexp* a_copy = clone(a_star);
    step->add(b_alt);
    step->add(a_copy);
    second->sub_regexp = step;
    second->flow_under_star(step);
    Regexp* out = new Regexp(concatenationExpr);
    out->add(a_star);
    out->add(second);
    if (g_trace.is_open()) g_trace << "denesting" << endl << to_string() << " -> " << out->to_string() << endl;
    return out->normalise(nullptr, false, false, {});
}

Regexp* Regexp::split_choice_on(const string& var) {                     // bnf.cpp:423-446
    Regexp* a_alt = new Regexp(alternationExpr);
    Regexp* b_alt = new Regexp(alternationExpr);
    for (Regexp* branch : sub_regexp->sub_regexps) {
        const bool free_or_unrelated = branch->uninited_read.count(var) ||
                                       (!branch->maybe_initialized.count(var) && !branch->maybe_read.count(var));
        (free_or_unrelated ? a_alt : b_alt)->add(branch);
    }
    a_alt = a_alt->unwrap_single();
    b_alt = b_alt->unwrap_single();
    if (b_alt->regexp_type == epsilon) return rw_choice_under_star();   // (rw | rw | f | f)*
    return denest(a_alt, b_alt);
}

Regexp* Regexp::split_choice_under_star(bool smallest_first) {           // bnf.cpp:448-470
    if (!is_alt(sub_regexp)) return this;
    const list<string> vars = sub_regexp->order_choice_vars(smallest_first);
    return vars.empty() ? this : split_choice_on(vars.front());
}

// bnf.cpp:472-553: denesting + sliding, (ab)* = (a (ba)* b | eps): a holds reads, b the LAST initialisation of `vars`
Regexp* Regexp::slide_last_init(set<string> vars) {
    Regexp* star = new Regexp(kleeneStar);
    Regexp* a_seq = new Regexp(concatenationExpr);
    Regexp* b_seq = new Regexp(concatenationExpr);
    auto it = sub_regexp->sub_regexps.rbegin();
    const auto stop = sub_regexp->sub_regexps.rend();
    while (it != stop) {                                                // from the back up to and including the last initialisation
        b_seq->add(*it, true);
        const bool found = !common(vars, (*it)->maybe_initialized).empty();
        ++it;
        if (found) break;
    }
    for (; it != stop; ++it) a_seq->add(*it, true);
    a_seq = a_seq->unwrap_single();
    b_seq = b_seq->unwrap_single();
    Regexp* with_eps = new Regexp(alternationExpr);
    with_eps->sub_regexps.push_back(new Regexp(epsilon));
    Regexp* seq = new Regexp(concatenationExpr);
    Regexp* turn = new Regexp(concatenationExpr);
    Regexp* b_copy = clone(b_seq);
    Regexp* a_copy = clone(a_seq);
    turn->add(b_copy);
    turn->add(a_copy);
    star->sub_regexp = turn;
    star->flow_under_star(turn);
    seq->add(a_seq);
    seq->add(star);
    seq->add(b_seq);
    Regexp* out = seq;
    if (is_star(this)) { with_eps->add(seq); out = with_eps; }
    // (sic) the iteration inside `out` stays as built: what follows makes new nodes and marks THOSE as slid
    Regexp* follow = star->sub_regexp->rw_vars.empty() ? star->normalise(nullptr, true, false, {}) : star->unroll({}, false);
    follow->is_slided = true;
    if (g_trace.is_open()) g_trace << "denesting + sliding" << endl << to_string() << " -> " << out->to_string() << endl;
    return out;
}

// bnf.cpp:555-636: an iteration over a sequence in which a read comes before its initialisation.  `where` is the iteration's
// place in `parent` when it is a child of a sequence or alternation being normalised (have_where); under an initialisation or
// another iteration the reference passes an iterator into a list that no longer exists and only gets away with it while the
// body has no read..write cells.
Regexp* Regexp::rw_sequence_under_star(Regexp* parent, bool have_where, list<Regexp*>::iterator where) {
    if (sub_regexp->crosses_references()) {
        is_bad_bnf = true;
        cout << to_string() << endl;
        cout << "\xd0\x9e\xd0\xb1\xd1\x80\xd0\xb0\xd1\x89\xd0\xb5\xd0\xbd\xd0\xb8\xd0\xb5 \xd1\x80\xd0\xb5\xd0\xb3\xd1\x83\xd0\xbb\xd1\x8f\xd1\x80\xd0\xbe\xd0\xba "
                "\xd1\x82\xd0\xb8\xd0\xbf\xd0\xb0 ({&j}:i {}:j &i)* \xd0\xbd\xd0\xb5 \xd0\xbf\xd0\xbe\xd0\xb4\xd0\xb4\xd0\xb5\xd1\x80\xd0\xb6\xd0\xb8\xd0\xb2\xd0\xb0\xd0\xb5\xd1\x82\xd1\x81\xd1\x8f, "
                "\xd1\x82\xd0\xb0\xd0\xba \xd0\xba\xd0\xb0\xd0\xba \xd1\x82\xd1\x80\xd0\xb5\xd0\xb1\xd1\x83\xd0\xb5\xd1\x82 \xd0\xb2\xd0\xb2\xd0\xb5\xd0\xb4\xd0\xb5\xd0\xbd\xd0\xb8\xd0\xb5 "
                "\xd0\xb2\xd1\x81\xd0\xbf\xd0\xbe\xd0\xbc\xd0\xbe\xd0\xb3\xd0\xb0\xd1\x82\xd0\xb5\xd0\xbb\xd1\x8c\xd0\xbd\xd1\x8b\xd1\x85 \xd1\x8f\xd1\x87\xd0\xb5\xd0\xb5\xd0\xba" << endl;
        return this;
    }
    const set<string> read_before_init = common(sub_regexp->definitely_uninit_read, sub_regexp->maybe_initialized);
    // cells whose last initialisation in the body differs from the last one in front of the iteration
    set<string> differing;
    if (parent && !have_where) {
        if (!sub_regexp->rw_vars.empty())
            throw std::runtime_error("bnf: an iteration with read..write cells directly under an initialisation or another iteration "
                                     "(the reference walks a destroyed list here, bnf.cpp:805 with 572-579)");
    } else if (parent && where != parent->sub_regexps.begin()) {
        for (const string& var : sub_regexp->rw_vars) {
            Regexp* mine = sub_regexp->last_init(var);
            if (!mine) continue;
            Regexp* before = parent->prefix_last_init(var, where);
            // (sic) the reference post-decrements its iterator inside the loop: every cell that is looked up moves the place one
            // child to the front, and libstdc++'s list is circular: from the first child to end(), from end() to the last child
            if (where == parent->sub_regexps.begin()) where = parent->sub_regexps.end();
            else --where;
            if (!mine->same_shape(before)) differing.insert(var);
        }
    } else {
        differing = sub_regexp->rw_vars;
    }
    if (!differing.empty()) return slide_last_init(differing);
    if (!sub_regexp->rw_vars.empty()) {
        Regexp* out = unroll({}, false);                                 // (eps | a* a)
        Regexp* last = out->sub_regexps.back();
        if (last->sub_regexps.empty())
            throw std::runtime_error("bnf: unrolling a + iteration over a single node (the reference reads an empty list here, bnf.cpp:589)");
        last->sub_regexps.front()->is_slided = true;
        return out;
    }
    if (!read_before_init.empty()) {
        // the initialisation sits in the same child as the read, (c(&1|{a*}:1&1)b)*: spread that alternation over the sequence
        int pos = (int)sub_regexp->sub_regexps.size() - 1;
        auto it = sub_regexp->sub_regexps.rbegin();
        for (; it != sub_regexp->sub_regexps.rend(); ++it, --pos)
            if (!common(common(read_before_init, (*it)->definitely_uninit_read), (*it)->maybe_initialized).empty()) break;
        if (it == sub_regexp->sub_regexps.rend())
            throw std::runtime_error("bnf: no child holds both the read and its initialisation (the reference uses an uninitialised iterator here, bnf.cpp:596-618)");
        if (is_alt(*it)) sub_regexp = sub_regexp->spread_both(pos);
        if (g_trace.is_open()) g_trace << "open alt in conc under kleene" << endl << " -> " << to_string() << endl;
        return this;
    }
    return this;
}

Regexp* Regexp::rw_choice_under_star() {                                 // bnf.cpp:638-656
    Regexp* first = new Regexp(alternationExpr);
    Regexp* others = new Regexp(alternationExpr);
    for (Regexp* branch : sub_regexp->sub_regexps) {
        const bool reads_first = !common(branch->definitely_uninit_read, branch->maybe_initialized).empty();
        if (reads_first && first->sub_regexps.empty()) first = branch;      // (sic) a branch without children is replaced by the next one
        else others->sub_regexps.push_back(branch);
    }
    others = others->unwrap_single();
    return denest(others, first);
}

Regexp* Regexp::rw_under_star(Regexp* parent, bool have_where, list<Regexp*>::iterator where) {      // bnf.cpp:658-667
    if (is_seq(sub_regexp)) return rw_sequence_under_star(parent, have_where, where);
    if (is_alt(sub_regexp)) return rw_choice_under_star();
    return this;
}

// bnf.cpp:669-739: an initialisation that a later alternative (or iteration) may leave unread, {}:1(&1|a): spread the
// alternative to the left so that every branch carries its own copy of the initialisation
Regexp* Regexp::sequence_fix_unread_inits() {
    Regexp* seq = this;
    set<string> met_init;
    const size_t n = seq->sub_regexps.size();
    size_t j = 0;
    auto it = seq->sub_regexps.begin();
    while (j < n) {
        const set<string> sub_init = common((*it)->unread_init, seq->unread_init);
        const set<string> may_read = common((*it)->uninited_read, met_init);
        if (is_init(*it) && is_alt((*it)->sub_regexp) && !may_read.empty()) *it = (*it)->hoist_choice_out_of_init();
        if ((is_alt(*it) || is_star(*it)) && !may_read.empty()) {
            if (is_star(*it) && !(*it)->is_slided) *it = (*it)->unroll_for_reads(may_read, seq, it);
            if (is_alt(*it)) {
                if (g_trace.is_open()) g_trace << "init without read " << to_string() << endl;
                Regexp* spread = seq->spread_left((int)j);
                seq->regexp_type = spread->regexp_type;
                seq->sub_regexps = list<Regexp*>(spread->sub_regexps);
                seq->flow_replace(spread);
                break;
            }
        }
        ++j; ++it;
        unite(met_init, sub_init);
    }
    return seq;
}

// bnf.cpp:741-803: a read that an earlier alternative may leave uninitialised, ({}:1&1|a)&1: spread to the right
Regexp* Regexp::sequence_fix_free_reads() {
    Regexp* seq = this;
    set<string> met_read;
    int j = (int)seq->sub_regexps.size() - 1;
    auto it = seq->sub_regexps.end();
    if (j >= 0) --it;
    long guard = 0;
    while (j >= 0) {
        if (++guard > 100000) throw std::runtime_error("bnf: the reference does not terminate on this regex (bnf.cpp:741-803)");
        const set<string> sub_read = common(seq->uninited_read, (*it)->uninited_read);
        const set<string> maybe_init = common((*it)->maybe_initialized, met_read);
        if (is_init(*it) && (is_alt((*it)->sub_regexp) || is_star((*it)->sub_regexp)) && !maybe_init.empty()) {
            if (is_star((*it)->sub_regexp) && !(*it)->sub_regexp->is_slided) (*it)->sub_regexp = (*it)->sub_regexp->unroll({});
            *it = (*it)->hoist_choice_out_of_init();
        }
        bool restarted = false;
        if ((is_alt(*it) || is_star(*it)) && !maybe_init.empty()) {
            if (is_star(*it) && !(*it)->is_slided) *it = (*it)->unroll({});
            if (is_alt(*it)) {
                if (g_trace.is_open()) g_trace << "read without init" << to_string() << endl;
                seq = seq->spread_right(j);
                if (is_alt(seq)) break;
                // the sequence now ends in the normalised alternative: go over it again from the back
                j = (int)seq->sub_regexps.size() - 1;
                it = seq->sub_regexps.end();
                if (j >= 0) --it;
                met_read.clear();
                restarted = true;
            }
        }
        if (!restarted) {
            --j;
            if (j >= 0) --it;
        }
        unite(met_read, sub_read);
    }
    return seq;
}

// bnf.cpp:805-892 _bnf
Regexp* Regexp::normalise(Regexp* parent, bool under_star, bool have_where, list<Regexp*>::iterator where) {
    (void)under_star;
    if (regexp_type == epsilon || regexp_type == literal || regexp_type == reference || is_bad_bnf) return this;
    if (is_alt(this) || is_seq(this)) {
        Regexp* out = new Regexp(regexp_type);
        for (auto it = sub_regexps.begin(); it != sub_regexps.end(); ++it) out->add((*it)->normalise(this, false, true, it));
        if (out->is_bad_bnf) return out;
        if (is_seq(out)) {
            if (!out->unread_init.empty()) out = out->sequence_fix_unread_inits();              // ({}:1(&1|a))*
            else if (!out->uninited_read.empty()) out = out->sequence_fix_free_reads();         // ({}:1&1|a)&1
        }
        return out;
    }
    if (is_iter(this)) {
        if (is_slided) return this;
        Regexp* out = new Regexp(regexp_type);
        Regexp* body = sub_regexp->normalise(this, true, false, {});
        out->sub_regexp = body;
        if (is_star(this)) out->flow_under_star(body);
        else out->flow_same(body);
        if (is_alt(out->sub_regexp)) return out->split_choice_under_star();
        Regexp* before = clone(out);
        out = out->rw_under_star(parent, have_where, where);
        if (!before->same_shape(out) && !out->is_bad_bnf) out = out->normalise(parent, false, have_where, where);
        return out;
    }
    if (is_init(this)) {
        Regexp* out = new Regexp(backreferenceExpr);
        out->variable = variable;
        out->sub_regexp = sub_regexp->normalise(this, false, false, {});
        out->flow_same(out->sub_regexp);
        mark_init(out);
        return out;
    }
    return this;
}

Regexp* Regexp::bnf(bool is_log) {                                       // bnf.cpp:894-919
    if (is_log) g_trace.open("log.txt", std::ofstream::out | std::ofstream::trunc);
    if (!is_acreg()) {
        cout << "\xd0\xa0\xd0\xb5\xd0\xb3\xd1\x83\xd0\xbb\xd1\x8f\xd1\x80\xd0\xbd\xd0\xbe\xd0\xb5 \xd0\xb2\xd1\x8b\xd1\x80\xd0\xb0\xd0\xb6\xd0\xb5\xd0\xbd\xd0\xb8\xd0\xb5 \xd0\xbd\xd0\xb5 "
                "\xd1\x83\xd0\xb4\xd0\xbe\xd0\xb2\xd0\xbb\xd0\xb5\xd1\x82\xd0\xb2\xd0\xbe\xd1\x80\xd1\x8f\xd0\xb5\xd1\x82 \xd1\x83\xd1\x81\xd0\xbb\xd0\xbe\xd0\xb2\xd0\xb8\xd1\x8e "
                "\xd0\xb0\xd1\x86\xd0\xb8\xd0\xba\xd0\xbb\xd0\xb8\xd1\x87\xd0\xbd\xd0\xbe\xd1\x81\xd1\x82\xd0\xb8 \xd0\xb8 \xd0\xbd\xd0\xb5 \xd0\xbc\xd0\xbe\xd0\xb6\xd0\xb5\xd1\x82 "
                "\xd0\xb1\xd1\x8b\xd1\x82\xd1\x8c \xd0\xbd\xd0\xbe\xd1\x80\xd0\xbc\xd0\xb0\xd0\xbb\xd0\xb8\xd0\xb7\xd0\xbe\xd0\xb2\xd0\xb0\xd0\xbd\xd0\xbe" << endl;
        is_bad_bnf = true;
        return this;
    }
    Regexp* out = normalise(nullptr, false, false, {});
    if (out->is_bad_bnf) return out;
    map<string, Regexp*> none;
    out->bind_reads(none);
    out = out->strip_dead_memory(out->definitely_unread_init, out->uninited_read);
    if (g_trace.is_open()) g_trace.close();
    return out;
}

// =====================================================================================================
// reversal (reverse.cpp:30-113)
// =====================================================================================================

Regexp* Regexp::mirrored() {                                             // reverse.cpp:30-57 _reverse
    switch (regexp_type) {
        case backreferenceExpr:
            sub_regexp = sub_regexp->mirrored();                          // in place: reads stay bound to this very node
            return this;
        case kleeneStar: case kleenePlus: {
            Regexp* r = new Regexp(regexp_type);
            r->sub_regexp = sub_regexp->mirrored();
            return r;
        }
        case concatenationExpr: case alternationExpr: {
            Regexp* r = new Regexp(regexp_type);
            for (Regexp* s : sub_regexps) {
                if (is_seq(this)) r->sub_regexps.push_front(s->mirrored());
                else r->sub_regexps.push_back(s->mirrored());
            }
            return r;
        }
        default: return this;
    }
}

// reverse.cpp:59-102: going through the mirrored tree from the left, the first time a cell's initialisation or one of its
// reads is met it becomes (stays) the initialisation, every later meeting becomes a read
Regexp* Regexp::swap_reads_and_inits(set<Regexp*>& done) {
    switch (regexp_type) {
        case reference:
            if (done.count(reference_to)) return this;
            if (!reference_to)
                throw std::runtime_error("reverse: a read that no initialisation is bound to (the reference dereferences a null pointer here, reverse.cpp:64-71)");
            done.insert(reference_to);
            reference_to->sub_regexp = reference_to->sub_regexp->swap_reads_and_inits(done);
            return reference_to;
        case backreferenceExpr: {
            if (!done.count(this)) { done.insert(this); return this; }
            Regexp* r = new Regexp(reference);
            r->variable = variable;
            return r;
        }
        case kleeneStar: case kleenePlus: {
            Regexp* r = new Regexp(regexp_type);
            r->sub_regexp = sub_regexp->swap_reads_and_inits(done);
            return r;
        }
        case concatenationExpr: case alternationExpr: {
            Regexp* r = new Regexp(regexp_type);
            for (Regexp* s : sub_regexps) r->sub_regexps.push_back(s->swap_reads_and_inits(done));
            return r;
        }
        default: return this;
    }
}

Regexp* Regexp::reverse() {                                              // reverse.cpp:104-113
    if (is_bad_bnf) cout << "Regex is not reversable in this version";
    set<Regexp*> done;
    return mirrored()->swap_reads_and_inits(done);
}
