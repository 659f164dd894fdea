// Specialised-kernel generator for small memory automata.
//
// The generic MFA kernel (kernels.hip: mfa_walk_kernel) interprets the automaton: node and edge
// tables are read at run time and the per-string slots live in LDS.  Its step latency is a chain of
// dependent scalar loads and LDS round trips, and a batch is only as fast as its longest string.
// For automata whose slots fit the register file, this file emits the SAME algorithm as straight-line
// HIP source for one automaton: every node and edge is unrolled with its constants, slots are VGPRs,
// and there is nothing to load per step but the input byte.  The source is compiled for gfx950 with
// hipcc (--genco) once per automaton and cached as a code object (jit.hip).
//
// Semantics are those documented at the top of kernels.hip (slot per node, minimum under the
// reference's set order).  The three-word key (P,Q,R) of the generic kernel is replaced by emission
// order: candidates are emitted in the reference's evaluation order, and a later candidate replaces a
// slot only if its (pos, first cell name) is STRICTLY smaller, so ties keep the older state:
//   phase A  waiting states re-insert themselves (mfa.cpp:195-197): oldest, written first;
//   phase B  states with pos == i, in node order, edges in list order, depth first (mfa.cpp:161-194);
//   phase C  states created by "unset cell" edges of waiting states (mfa.cpp:148-160): such a state
//            keeps its source's pos, so two of them tie only when their sources have equal pos, and
//            then node order is the reference's (pos, node) evaluation order.
//
// Run acceleration.  The step function is emitted as a template over the value type (device_common.h:
// plain uint32_t or Dual = value + change per step).  Inside a long run of equal input bytes a lane
//   1. saves its slots, takes one plain step, and forms d = slots now - slots saved;
//   2. takes one DUAL step from there with direction d.  That step yields the next slots, the direction
//      the step map sends d to, and TB = for how many consecutive steps every comparison made in the
//      step keeps its outcome if the slots keep moving by d per step;
//   3. if the slots moved by exactly d again and d was mapped to itself, then -- the step map being
//      affine for fixed comparison outcomes -- the slots move by d in every one of the next TB-1 steps
//      too, and the lane adds (TB-1)*d and advances i by TB-1 without executing them.
// TB also covers "the byte stays the same" (i < end of run) and "not the final pass" (i != len), because
// those are comparisons of the step like any other.  Nothing is approximated: a jump is taken only over
// steps whose every branch outcome is proven, and any data-dependent comparison of string bytes inside
// the step forces TB = 1.
#include <cstdio>
#include <sstream>
#include <string>

#include "mfa_internal.h"

namespace mfa {

namespace {

const char* kPrelude =
#include "device_common.inc"
    ;

// Automata whose two slot sets do not fit the register file ("huge": > 272 slot registers) keep every slot set
// in LDS and let fewer lanes of the wave carry strings, as many as one CU's 160 KiB allow for the seven images
// a dual step and its probe need (current and next set with directions, three probe images), plus one column
// that the idle lanes scribble on.  The idle lanes still take part in the cooperative scans.
static uint32_t huge_lanes(uint32_t n_words) {
    // per column: current and next set with directions (the probe images live in the wave's scratch area, which only the
    // sparse loops of the main loop touch), the lane's Input and its region-table cache; 1 KiB for everything else
    const size_t column = (size_t)4 * n_words * 4 + 160 /* >= sizeof(Input), asserted in the generated source */ + 8 * 2 /* MFA_RT_CACHED */;
    for (uint32_t lanes = 64; lanes >= 8; lanes -= 2)          // even: the stride lanes + 1 is odd
        if (column * (lanes + 1) + 1024 <= 160 * 1024) return lanes;
    return 0;
}

// development knobs (folded into the cache key by jit.hip)
static int knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

struct Gen {
    const HostImage& g;
    std::ostringstream o;
    int K;
    bool rev;
    int tmp = 0;
    bool sparse = false;      // huge automata: the step only touches slots that the wave's occupancy mask marks (emit_step_chunked)

    explicit Gen(const HostImage& img) : g(img), K((int)(img.h.n_cells ? img.h.n_cells : 1)), rev(img.h.is_reversed != 0) {}

    bool is_finish(uint32_t n) const { return n == g.h.finish; }
    uint32_t deg(uint32_t n) const { return g.edge_begin[n + 1] - g.edge_begin[n]; }
    const mfa_blob_edge& edge(uint32_t n, uint32_t k) const { return g.edges[g.edge_begin[n] + k]; }
    static bool eps(const mfa_blob_edge& e) { return e.flags & MFA_EDGE_EPS; }
    static int digit(const mfa_blob_edge& e) { return (!eps(e) && e.label >= '1' && e.label <= '9') ? e.label - '1' : -1; }
    static std::string num(long v) { return std::to_string(v); }

    // a symbolic state: names of the variables (type U) that hold it; known[c]: cell c is statically present
    struct Sym { std::string pos; std::vector<std::string> S, L, F; std::vector<bool> known; };

    std::string present(const Sym& s, int c) { return "(flagv(U(" + s.F[c] + "), TB) & F_PRESENT)"; }

    std::string fname(const Sym& s) {          // plain uint32_t expression
        std::string e = "0u";
        for (int c = K - 1; c >= 0; c--) e = "(" + present(s, c) + " ? " + num(c + 1) + "u : " + e + ")";
        return e;
    }

    // "this node has an edge that lets a waiting state wait" (mfa.cpp:195-197 is reached by letter, dot and
    // present-cell edges only)
    std::string qualifies(uint32_t n, const Sym& s) {
        std::string dyn;
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e)) continue;
            int d = digit(e);
            if (d < 0) return "true";
            if (s.known[d]) return "true";
            dyn += (dyn.empty() ? "" : " || ") + present(s, d);
        }
        return dyn.empty() ? "false" : "(" + dyn + ")";
    }

    // The outcome of a cell read on the lanes in `rd`: `ok`.  Run extents and byte-wise comparisons are done by the whole wave, one
    // lane's at a time.
    void emit_read_check(const std::string& ind, const std::string& rd, const std::string& vs, const std::string& vl, const std::string& vf,
                         const std::string& ok, const std::string& id) {
                o << ind << "    {                                                  // run extent for one-byte-repeated values: found by the whole wave\n";
                o << ind << "      bool nr = " << rd << " && uni_needs_run(in, val(i), ch, val(" << vl << "), " << vf << ");\n";
                o << ind << "      if (nr && in.rt != nullptr) {                      // the pre-pass knows every long run\n"
                  << ind << "        uint32_t rh;\n"
                  << ind << "        if (rt_run<REV>(in, val(i), rh) || run_end_bounded<REV>(in, val(i), ch, 192u, rh)) { in.run_lo = val(i); in.run_hi = rh; in.run_ch = ch; nr = false; }\n"
                  << ind << "      }\n";
                o << ind << "      for (unsigned long long sb = __ballot(nr); sb; sb &= sb - 1ull) {\n";
                o << ind << "        const int L = __builtin_ctzll(sb);\n";
                o << ind << "        const uint32_t r = coop_run_end<REV>(in.bytes, ((uint64_t)__shfl((uint32_t)(in.base >> 32), L) << 32) | __shfl((uint32_t)in.base, L),\n";
                o << ind << "                                            __shfl(in.len, L), __shfl(val(i), L), threadIdx.x & 63u);\n";
                // a run is a periodic region too: remember it as one, so that the main loop's look does not measure it again
                // (unless a region that reaches at least as far is already known -- a probe may be relying on it)
                o << ind << "        if ((threadIdx.x & 63u) == (uint32_t)L) {\n"
                  << ind << "          in.run_lo = val(i); in.run_hi = r; in.run_ch = ch;\n"
                  << ind << "          if (!(in.per_q != 0u && in.per_lo <= val(i) && val(i) < in.per_hi && in.per_hi >= r)) { in.per_lo = val(i); in.per_hi = r; in.per_q = 1u; }\n"
                  << ind << "        }\n";
                o << ind << "      }\n";
                o << ind << "    }\n";
                o << ind << "    bool " << ok << " = false, cmp" << id << " = false;\n";
                o << ind << "    if (" << rd << ") " << ok << " = read_pre_u<REV, U>(in, i, ch, " << vs << ", " << vl << ", " << vf << ", TB, cmp" << id << ");\n";
                o << ind << "    for (unsigned long long sb = __ballot(cmp" << id << "); sb; sb &= sb - 1ull) {      // byte-wise comparisons: one lane's at a time, whole wave\n";
                o << ind << "      const int L = __builtin_ctzll(sb);\n";
                o << ind << "      const uint32_t ca = val(" << vs << "), cb = val(i), cl = val(" << vl << ");\n";
                o << ind << "      const uint64_t pa = in.base + (REV ? (uint64_t)(in.len - ca - cl) : (uint64_t)ca), pb = in.base + (REV ? (uint64_t)(in.len - cb - cl) : (uint64_t)cb);\n";
                o << ind << "      const bool r = coop_mem_equal(in.bytes, ((uint64_t)__shfl((uint32_t)(pa >> 32), L) << 32) | __shfl((uint32_t)pa, L),\n";
                o << ind << "                                    ((uint64_t)__shfl((uint32_t)(pb >> 32), L) << 32) | __shfl((uint32_t)pb, L), __shfl(cl, L), threadIdx.x & 63u);\n";
                o << ind << "      if ((threadIdx.x & 63u) == (uint32_t)L) " << ok << " = r;\n";
                o << ind << "    }\n";
    }

    bool outline_reads = false;

    void insert(uint32_t m, const std::string& pred, const std::string& P, const Sym& t, const std::string& ind) {
        std::string w = "w" + num(tmp++), pv = "p" + num(tmp++);
        o << ind << "{ const U " << pv << " = " << P << ";\n";
        o << ind << "  const bool " << w << " = (" << pred << ") && lt(" << pv << ", U(n.P" << m << "), TB);\n";
        o << ind << "  if (!NextSet<U>::in_lds || __any(" << w << ")) {       // a next set in LDS is only touched when some lane wins\n";
        if (sparse) o << ind << "  occn" << (m < 64 ? 0 : 1) << " |= 1ull << " << (m & 63) << "; any_next = any_next || " << w << ";\n";
        o << ind << "  n.P" << m << " = sel(" << w << ", " << pv << ", U(n.P" << m << "));\n";
        for (int c = 0; c < K; c++) {
            std::string sfx = num(m) + "_" + num(c);
            o << ind << "  assign_if(n.S" << sfx << ", " << w << ", U(" << t.S[c] << "));";
            o << " assign_if(n.L" << sfx << ", " << w << ", U(" << t.L[c] << "));";
            o << " assign_if(n.F" << sfx << ", " << w << ", U(" << t.F[c] << "));\n";
        }
        o << ind << "  }\n" << ind << "}\n";
    }

    // declare a copy of `s` as fresh variables; returns the new symbolic state
    Sym copy_of(const Sym& s, const std::string& ind) {
        Sym t = s;
        int id = tmp++;
        for (int c = 0; c < K; c++) {
            t.S[c] = "tS" + num(id) + "_" + num(c);
            t.L[c] = "tL" + num(id) + "_" + num(c);
            t.F[c] = "tF" + num(id) + "_" + num(c);
            o << ind << "U " << t.S[c] << " = U(" << s.S[c] << "), " << t.L[c] << " = U(" << s.L[c] << "), " << t.F[c] << " = U(" << s.F[c] << ");\n";
        }
        return t;
    }

    // MFA::doMemoryWriteActions on `t` for the lanes in `pred`
    void apply_actions(const Sym& t, uint32_t actions, const std::string& pred, const std::string& ts, const std::string& tl,
                       const std::string& tuni, const std::string& tch, const std::string& ind) {
        for (int c = 0; c < K; c++) {
            uint32_t act = (actions >> (2 * (c + 1))) & 3u;
            std::string a = t.S[c] + ", " + t.L[c] + ", " + t.F[c];
            if (act == MFA_ACT_OPEN) o << ind << "act_open_u<U>(" << a << ", " << ts << ", " << tl << ", " << tuni << ", " << tch << ");\n";
            else if (act == MFA_ACT_CLOSE) o << ind << "if (" << pred << ") act_close_u<U>(" << a << ", TB);\n";
            else o << ind << "if (" << pred << ") act_none_u<U>(" << a << ", " << ts << ", " << tl << ", " << tuni << ", " << tch << ", TB);\n";
        }
    }

    // Edges of node `n` for the lanes in `pred`, whose state is `s`.
    //   current = true : the lanes sit at pos == i and may consume (phase B)
    //   current = false: the lanes are waiting (pos > i) or at pos == len in the final pass (phase C):
    //                    only unset-cell recursion, epsilon acceptance and (level > 0) self-insertion
    void edges(uint32_t n, Sym& s, const std::string& pred, int level, bool current, const std::string& ind) {
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e)) {
                if (!current) o << ind << "accept = accept || ((" << pred << ") && eq(" << s.pos << ", len, TB));\n";
                continue;
            }
            const int d = digit(e);
            std::string other = pred;
            if (d >= 0 && !s.known[d]) {
                // unset cell (mfa.cpp:148-160): create it, recurse into the target with the same pos
                if (level < K && !is_finish(e.target)) unset_cell(e, s, pred, level, current, ind);
                other = "(" + pred + ") && " + present(s, d);
            }
            if (!current) continue;
            // consume (mfa.cpp:161-194): the literal test comes first, also for digit labels
            std::string lit = "l" + num(tmp++);
            o << ind << "{ const bool " << lit << " = (" << other << ")" << (e.label == '.' ? "" : " && ch == " + num(e.label) + "u") << ";\n";
            o << ind << "  if (__any(" << lit << ")) {\n";
            {
                Sym t = copy_of(s, ind + "    ");
                apply_actions(t, e.actions, lit, "i", "konst<U>(1u)", "true", "ch", ind + "    ");
                insert(e.target, lit, "mkp(add(i, konst<U>(1u)), " + fname(t) + ")", t, ind + "    ");
            }
            o << ind << "  }\n";
            if (d >= 0 && e.label != '.') {
                std::string rd = "r" + num(tmp++);
                o << ind << "  const bool " << rd << " = (" << other << ") && !" << lit << ";\n";
                o << ind << "  if (__any(" << rd << ")) {\n";
                Sym t = copy_of(s, ind + "    ");                      // copy BEFORE read() marks the source (mfa.cpp:167/177)
                std::string id = num(tmp++);
                std::string vs = "vs" + id, vl = "vl" + id, vf = "vf" + id, ok = "k" + id;
                o << ind << "    const U " << vs << " = U(" << s.S[d] << "), " << vl << " = U(" << s.L[d] << "); const uint32_t " << vf << " = flagv(U("
                  << s.F[d] << "), TB);\n";
                o << ind << "    " << s.F[d] << " = sel(" << rd << ", konst<U>(" << vf << " | F_READ), U(" << s.F[d] << "));\n";
                if (outline_reads) {
                    // huge automata: the whole check is one out-of-line function (a read site inlined is ~10 KB of code, eight of them per node)
                    o << ind << "    bool " << ok << " = false;\n"
                      << ind << "    { const ReadOut ro = huge_cell_read<U>(i, ch, " << vs << ", " << vl << ", " << vf << ", " << rd << ", TB); " << ok << " = ro.ok; TB = ro.TB; }\n";
                } else emit_read_check(ind, rd, vs, vl, vf, ok, id);
                o << ind << "    if (__any(" << ok << ")) {\n";
                apply_actions(t, e.actions, ok, "i", vl, "((" + vf + " & F_UNI) != 0u)", "((" + vf + " >> 8) & 0xffu)", ind + "      ");
                insert(e.target, ok, "mkp(add(i, " + vl + "), " + fname(t) + ")", t, ind + "      ");
                o << ind << "    }\n" << ind << "  }\n";
            }
            o << ind << "}\n";
        }
    }

    void unset_cell(const mfa_blob_edge& e, Sym& s, const std::string& pred, int level, bool current, const std::string& ind) {
        const int d = digit(e);
        std::string p2 = "a" + num(tmp++);
        o << ind << "{ const bool " << p2 << " = (" << pred << ") && !" << present(s, d) << ";\n";
        o << ind << "  if (__any(" << p2 << ")) {\n";
        Sym t = copy_of(s, ind + "    ");
        uint32_t act = (e.actions >> (2 * (d + 1))) & 3u;
        o << ind << "    " << t.S[d] << " = " << s.pos << "; " << t.L[d] << " = konst<U>(0u); " << t.F[d] << " = konst<U>(F_PRESENT | F_UNI"
          << (act == MFA_ACT_OPEN ? " | F_OPEN" : "") << ");\n";
        t.known[d] = true;
        edges(e.target, t, p2, level + 1, current, ind + "    ");
        if (!current) {
            // the new state waits at the target if the target lets it (mfa.cpp:195-197)
            std::string q = qualifies(e.target, t);
            if (q != "false")
                insert(e.target, p2 + " && !final_pass && gt(" + s.pos + ", i, TB) && " + q, "mkp(" + s.pos + ", " + fname(t) + ")", t, ind + "    ");
        }
        o << ind << "  }\n" << ind << "}\n";
    }

    Sym cur_sym(uint32_t n, const std::string& prefix) {
        Sym s;
        s.pos = "pos" + num(n);
        for (int c = 0; c < K; c++) {
            s.S.push_back(prefix + "S" + num(n) + "_" + num(c));
            s.L.push_back(prefix + "L" + num(n) + "_" + num(c));
            s.F.push_back(prefix + "F" + num(n) + "_" + num(c));
            s.known.push_back(false);
        }
        return s;
    }

    std::vector<std::string> slot_words() {           // member names of SlotSet, in a fixed order
        std::vector<std::string> w;
        for (uint32_t n = 0; n < g.h.n_nodes; n++) {
            if (is_finish(n)) continue;
            w.push_back("P" + num(n));
            for (int c = 0; c < K; c++) {
                w.push_back("S" + num(n) + "_" + num(c));
                w.push_back("L" + num(n) + "_" + num(c));
                w.push_back("F" + num(n) + "_" + num(c));
            }
        }
        return w;
    }

    // ---- pieces of one step -----------------------------------------------------------------------------
    void emit_classify(uint32_t n) {                      // pos / live / here / wait of slot n
        std::string s = num(n);
        o << "  const U pos" << s << " = posof(U(c.P" << s << "), TB);\n";
        o << "  bool live" << s << " = active && ne(U(c.P" << s << "), konst<U>(MFA_EMPTY), TB) && ge(pos" << s << ", i, TB);\n";
        if (rev) {                                               // mfa.cpp:116-133
            o << "  if (live" << s << ") { U need = konst<U>(0u);";
            for (int c = 0; c < K; c++) {
                std::string f = "flagv(U(c.F" + s + "_" + num(c) + "), TB)";
                o << " if ((" << f << " & F_PRESENT) && ((" << f << " & F_OPEN) || !(" << f << " & F_READ))) need = add(need, U(c.L" << s << "_" << c << "));";
            }
            o << " live" << s << " = le(need, sub(len, i), TB); }\n";
        }
        o << "  const bool here" << s << " = live" << s << " && !final_pass && eq(pos" << s << ", i, TB);\n";
        o << "  const bool wait" << s << " = live" << s << " && !final_pass && !here" << s << ";\n";
        o << "  (void)wait" << s << ";\n";
    }

    void emit_phase_a(uint32_t n) {                       // epsilon acceptance of the slot state, waiting states carry over
        Sym s = cur_sym(n, "c.");
        bool has_eps = false;
        for (uint32_t k = 0; k < deg(n); k++) has_eps = has_eps || eps(edge(n, k));
        if (has_eps) o << "  accept = accept || (live" << n << " && eq(pos" << n << ", len, TB));\n";
        std::string q = qualifies(n, s);
        if (sparse) {
            // the next set is all-empty when a step starts (huge_commit leaves it so): only a state that waits is written, with its cells
            if (q == "false") return;
            o << "  { const bool carry = wait" << n << " && " << q << ";\n    if (__any(carry)) {\n      occn" << (n < 64 ? 0 : 1) << " |= 1ull << " << (n & 63)
              << "; any_next = any_next || carry;\n      assign_if(n.P" << n << ", carry, U(c.P" << n << "));";
            for (int c = 0; c < K; c++) {
                std::string sfx = num(n) + "_" + num(c);
                o << " assign_if(n.S" << sfx << ", carry, U(c.S" << sfx << ")); assign_if(n.L" << sfx << ", carry, U(c.L" << sfx << ")); assign_if(n.F" << sfx
                  << ", carry, U(c.F" << sfx << "));";
            }
            o << "\n    }\n  }\n";
            return;
        }
        o << "  n.P" << n << " = sel(wait" << n << " && " << q << ", U(c.P" << n << "), konst<U>(MFA_EMPTY));";
        for (int c = 0; c < K; c++) {
            std::string sfx = num(n) + "_" + num(c);
            o << " n.S" << sfx << " = U(c.S" << sfx << "); n.L" << sfx << " = U(c.L" << sfx << "); n.F" << sfx << " = U(c.F" << sfx << ");";
        }
        o << "\n";
    }

    void emit_phase_b(uint32_t n) {                       // the state at pos == i consumes
        o << "  if (__any(here" << n << ")) {   // node " << n << "\n";
        Sym s = cur_sym(n, "c.");
        edges(n, s, "here" + num(n), 0, true, "    ");
        o << "  }\n";
    }

    bool has_phase_c(uint32_t n) {
        for (uint32_t k = 0; k < deg(n); k++)
            if (digit(edge(n, k)) >= 0 && !is_finish(edge(n, k).target)) return true;
        return false;
    }

    void emit_phase_c(uint32_t n) {                       // unset-cell edges of waiting states (and of pos == len states in the final pass)
        o << "  { const bool late" << n << " = live" << n << " && !here" << n << ";\n";
        o << "    if (__any(late" << n << ")) {\n";
        Sym s = cur_sym(n, "c.");
        for (uint32_t k = 0; k < deg(n); k++) {
            const auto& e = edge(n, k);
            if (eps(e) || digit(e) < 0 || is_finish(e.target)) continue;        // epsilon acceptance of the slot state: phase A
            unset_cell(e, s, "late" + num(n), 0, false, "      ");
        }
        o << "    }\n  }\n";
    }

    void emit_end() {
        o << "  any_next = false;\n";
        for (uint32_t n = 0; n < g.h.n_nodes; n++)
            if (!is_finish(n)) o << "  any_next = any_next || ne(U(n.P" << n << "), konst<U>(MFA_EMPTY), TB);\n";
        for (const auto& w : slot_words()) o << "  c." << w << " = U(n." << w << ");\n";
    }

    // One step as a single inlined function: slot sets are registers, everything stays in one scope.
    void emit_step() {
        const uint32_t N = g.h.n_nodes;
        o << "template <class U>\n__device__ __forceinline__ void mfa_step(SlotSet<U>& c, Input& in, const U i, const U len, const uint32_t ch,\n"
             "                                         const bool final_pass, bool& accept, bool& any_next, tb_t& TB, uint32_t* cur_mem,\n"
             "                                         uint32_t* nxt_mem, const bool active, unsigned long long& occ0, unsigned long long& occ1) {\n";
        o << "  NextSet<U> n(nxt_mem);\n  (void)cur_mem; (void)occ0; (void)occ1;\n";
        for (uint32_t n = 0; n < N; n++) if (!is_finish(n)) emit_classify(n);
        for (uint32_t n = 0; n < N; n++) if (!is_finish(n)) emit_phase_a(n);
        for (uint32_t n = 0; n < N; n++) if (!is_finish(n) && deg(n) != 0) emit_phase_b(n);
        for (uint32_t n = 0; n < N; n++) if (!is_finish(n) && has_phase_c(n)) emit_phase_c(n);
        emit_end();
        o << "}\n\n";
    }

    // Huge automata: the slot sets are in LDS anyway, so the step is cut into out-of-line functions of a few nodes
    // each (one inlined function of several hundred edges takes the compiler tens of minutes).  Phase order is kept:
    // all of A, then B in node order, then C in node order.
    // Huge automata.  Two things keep a step from costing what 76 slots cost when two to eight are occupied:
    //  * occ0/occ1 (wave-uniform, kept by the main loop): bit n set = slot n may be occupied in some lane.  Everything a step does
    //    for a node sits behind a scalar test of its bit -- an empty slot costs a compare and a branch, no LDS access;
    //  * the next set is all-empty when a step starts.  Carried and inserted states set their node's bit in occn; huge_commit then
    //    moves exactly those nodes' words from the next set to the current one (per lane only where the slot is really occupied, so
    //    that the cell words of a lane's empty slots stay what they were: the probes compare whole slot sets), empties them in the
    //    next set again, and empties the current slots that were vacated.
    void emit_step_chunked() {
        const uint32_t N = g.h.n_nodes;
        sparse = knob("MFA_GEN_SPARSE", 1) != 0;
        // What a step works on besides the slot sets travels by value, in and out: by reference it went through the caller's
        // stack, a dozen FLAT accesses with their full latency per call and twenty calls a step.
        const char* sig = "(const StepIO io, const U i, const U len, const uint32_t ch, const bool final_pass, const bool active,\n"
                          "    unsigned long long occ0_arg, unsigned long long occ1_arg)";
        // The slot sets and the input state are reached through the file-scope LDS arrays, not through the pointer
        // and reference parameters: only then does the compiler know the address space and emit DS instructions
        // (through generic pointers every slot access was a FLAT instruction: 4 200 of them per step).
        const char* pre = "  const uint32_t hl_lane = threadIdx.x & 63u, hl_col = hl_lane < LANES ? hl_lane : LANES;\n"
                          "  uint32_t* const cm = huge_lds + hl_col; uint32_t* const nm = cm + 2 * N_WORDS * PSTRIDE;\n"
                          "  Input& in = huge_in[hl_col];\n"
                          "  SlotSet<U> c(cm); NextSet<U> n(nm); (void)c; (void)n; (void)in; (void)ch;\n"
                          "  bool accept = io.accept, any_next = io.any_next; tb_t TB = io.TB;\n"
                          "  const unsigned long long occ0 = uniform64(occ0_arg), occ1 = uniform64(occ1_arg); (void)occ0; (void)occ1;\n"
                          "  unsigned long long occn0 = uniform64(io.occn0), occn1 = uniform64(io.occn1);\n";
        const char* post = "  return StepIO{accept, any_next, TB, occn0, occn1};\n}\n\n";
        outline_reads = true;
        o << "struct ReadOut { bool ok; tb_t TB; };\n"
             "template <class U>\n__device__ __attribute__((noinline)) ReadOut huge_cell_read(const U i, const uint32_t ch, const U vs, const U vl, const uint32_t vf, const bool rd, tb_t TB) {\n"
             "  const uint32_t hl_lane = threadIdx.x & 63u; Input& in = huge_in[hl_lane < LANES ? hl_lane : LANES];\n";
        emit_read_check("", "rd", "vs", "vl", "vf", "ok", "_r");
        o << "  return ReadOut{ok, TB};\n}\n\n";
        o << "struct StepIO { bool accept, any_next; tb_t TB; unsigned long long occn0, occn1; };\n"
             "__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {\n"
             "  return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);\n}\n";
        // Sparse mode: one function per node and phase, called only when the node's bit is set.  A step then runs through the
        // code of the occupied nodes and a compact row of bit tests, not through 700 KB of guarded blocks: with one wave per CU
        // every skipped block was an instruction-cache miss (~230 per step, 80-150 k cycles a step on ex. 8 -reverse).
        std::vector<std::string> calls, call_guard;
        uint64_t fn_mask[2] = {0, 0};
        auto begin_fn = [&](const std::string& name) {
            o << "template <class U>\n__device__ __attribute__((noinline)) StepIO " << name << sig << " {\n" << pre;
            calls.push_back(name);
            fn_mask[0] = fn_mask[1] = 0;
        };
        auto end_fn = [&]() {
            o << post;
            std::string gd;
            if (sparse) {
                char buf[96];
                snprintf(buf, sizeof buf, "((occ0 & 0x%llxull) | (occ1 & 0x%llxull)) != 0ull", (unsigned long long)fn_mask[0], (unsigned long long)fn_mask[1]);
                gd = buf;
            }
            call_guard.push_back(gd);
        };
        auto node_in_fn = [&](uint32_t n) { fn_mask[n >> 6] |= 1ull << (n & 63); };
        const uint32_t chunk_budget = sparse ? 1 : 24;
        if (sparse) {
            for (uint32_t n = 0; n < N; n++) if (!is_finish(n)) { begin_fn("step_a" + num(n)); node_in_fn(n); emit_classify(n); emit_phase_a(n); end_fn(); }
        } else {
            begin_fn("step_a");
            for (uint32_t n = 0; n < N; n++) if (!is_finish(n)) { emit_classify(n); emit_phase_a(n); }
            end_fn();
        }
        int chunk = 0;
        uint32_t budget = 0;
        bool open = false;
        for (uint32_t n = 0; n < N; n++) {
            if (is_finish(n) || deg(n) == 0) continue;
            if (!open) { begin_fn("step_b" + num(chunk++)); open = true; budget = 0; }
            node_in_fn(n);
            emit_classify(n);
            emit_phase_b(n);
            budget += deg(n);
            if (budget >= chunk_budget) { end_fn(); open = false; }
        }
        if (open) { end_fn(); open = false; }
        chunk = 0;
        for (uint32_t n = 0; n < N; n++) {
            if (is_finish(n) || !has_phase_c(n)) continue;
            if (!open) { begin_fn("step_c" + num(chunk++)); open = true; budget = 0; }
            node_in_fn(n);
            emit_classify(n);
            emit_phase_c(n);
            budget += deg(n);
            if (budget >= chunk_budget) { end_fn(); open = false; }
        }
        if (open) { end_fn(); open = false; }
        if (!sparse) {
            begin_fn("step_end");
            emit_end();
            end_fn();
        } else {
            // words of node n start at word (n minus the finish node if it comes before) * (1 + 3 K): P, then S, L, F per cell
            o << "template <class U> struct HugeWord;\n"
                 "template <> struct HugeWord<uint32_t> { static constexpr bool dual = false; };\n"
                 "template <> struct HugeWord<Dual> { static constexpr bool dual = true; };\n";
            o << "template <class U>\n__device__ __attribute__((noinline)) void huge_commit(unsigned long long occ0_arg, unsigned long long occ1_arg, unsigned long long occn0_arg,\n"
                 "                                                          unsigned long long occn1_arg) {\n"
                 "  const uint32_t hl_lane = threadIdx.x & 63u, hl_col = hl_lane < LANES ? hl_lane : LANES;\n"
                 "  uint32_t* const cm = huge_lds + hl_col; uint32_t* const nm = cm + 2 * N_WORDS * PSTRIDE;\n"
                 "  const unsigned long long occ[2] = {uniform64(occ0_arg), uniform64(occ1_arg)}, occn[2] = {uniform64(occn0_arg), uniform64(occn1_arg)};\n"
                 "  constexpr uint32_t WPN = " << (1 + 3 * K) << "u, FIN = " << g.h.finish << "u;\n"
                 "  for (int h = 0; h < 2; h++)\n"
                 "    for (unsigned long long m = occ[h] | occn[h]; m; m &= m - 1ull) {\n"
                 "      const uint32_t node = 64u * (uint32_t)h + (uint32_t)__builtin_ctzll(m);\n"
                 "      const uint32_t k0 = (node - (node > FIN ? 1u : 0u)) * WPN;\n"
                 "      uint32_t* const cw = cm + 2u * k0 * PSTRIDE; uint32_t* const nw = nm + 2u * k0 * PSTRIDE;\n"
                 "      if (!((occn[h] >> (node & 63u)) & 1ull)) { cw[0] = MFA_EMPTY; if (HugeWord<U>::dual) cw[PSTRIDE] = 0u; continue; }      // vacated everywhere\n"
                 "      const uint32_t p = nw[0];\n"
                 "      cw[0] = p; if (HugeWord<U>::dual) cw[PSTRIDE] = nw[PSTRIDE];\n"
                 "      if (p != MFA_EMPTY) {\n"
                 "        for (uint32_t j = 1; j < WPN; j++) { cw[2u * j * PSTRIDE] = nw[2u * j * PSTRIDE]; if (HugeWord<U>::dual) cw[(2u * j + 1u) * PSTRIDE] = nw[(2u * j + 1u) * PSTRIDE]; }\n"
                 "      } else if (HugeWord<U>::dual && !((occ[h] >> (node & 63u)) & 1ull)) {\n"
                 "        for (uint32_t j = 1; j < WPN; j++) cw[(2u * j + 1u) * PSTRIDE] = 0u;      // a node joining in the middle of a dual period: its cells rest in the lanes where it stays empty\n"
                 "      }\n"
                 "      nw[0] = MFA_EMPTY; nw[PSTRIDE] = 0u;\n"
                 "    }\n}\n\n";
        }
        o << "template <class U>\n__device__ __forceinline__ void mfa_step(SlotSet<U>&, Input& in, const U i, const U len, const uint32_t ch,\n"
             "                                         const bool final_pass, bool& accept, bool& any_next, tb_t& TB, uint32_t* cur_mem,\n"
             "                                         uint32_t* nxt_mem, const bool active, unsigned long long& occ0, unsigned long long& occ1) {\n";
        if (sparse) o << "  any_next = false;\n";
        o << "  StepIO io{accept, any_next, TB, 0ull, 0ull};\n  (void)in; (void)cur_mem; (void)nxt_mem;\n";
        for (size_t f = 0; f < calls.size(); f++)
            o << "  " << (call_guard[f].empty() ? "" : "if (" + call_guard[f] + ") ") << "io = " << calls[f]
              << "<U>(io, i, len, ch, final_pass, active, occ0, occ1);\n";
        o << "  accept = io.accept; any_next = io.any_next; TB = io.TB;\n";
        if (sparse) o << "  huge_commit<U>(occ0, occ1, io.occn0, io.occn1);\n  occ0 = io.occn0; occ1 = io.occn1;\n";
        o << "}\n\n";
    }

    // One statement per slot word.  In `body`, K_ is the word's index, CW_ / DW_ the word of the current set as a plain value
    // and as a dual number, ISP_ whether it is a P word.  Huge automata in sparse mode loop at run time over the words of
    // the nodes in pocc0/pocc1 (every node occupied in some lane at some step since the running probes began: all other
    // words have not moved and nobody reads them); everything else is unrolled over all words.
    void each_word(const std::vector<std::string>& words, bool loop, const std::string& ind, const std::string& body) {
        auto subst = [&](const std::string& k, const std::string& cw, const std::string& dw, const std::string& isp) {
            std::string r;
            for (size_t a = 0; a < body.size();) {
                auto take = [&](const char* tok, const std::string& with) {
                    const size_t n = std::char_traits<char>::length(tok);
                    if (body.compare(a, n, tok) == 0) { r += with; a += n; return true; }
                    return false;
                };
                if (take("K_", k) || take("CW_", cw) || take("DW_", dw) || take("ISP_", isp)) continue;
                r += body[a++];
            }
            return r;
        };
        if (!loop) {
            for (size_t k = 0; k < words.size(); k++)
                o << ind << "{ " << subst(num((uint32_t)k), "c." + words[k], "dc." + words[k], words[k][0] == 'P' ? "true" : "false") << " }\n";
            return;
        }
        o << ind << "for (int h_ = 0; h_ < 2; h_++)\n"
          << ind << "  for (unsigned long long m_ = uniform64(h_ ? pocc1 : pocc0) & ~(h_ == " << (g.h.finish < 64 ? 0 : 1) << " ? 1ull << " << (g.h.finish & 63) << " : 0ull); m_; m_ &= m_ - 1ull) {\n"
          << ind << "    const uint32_t node_ = 64u * (uint32_t)h_ + (uint32_t)__builtin_ctzll(m_);\n"
          << ind << "    const uint32_t k0_ = (node_ - (node_ > " << g.h.finish << "u ? 1u : 0u)) * " << (1 + 3 * K) << "u;\n"
          << ind << "    for (uint32_t j = 0; j < " << (1 + 3 * K) << "u; j++) {\n"
          << ind << "      const uint32_t k = k0_ + j; LdsPlain cw{cur_mem + 2u * k * PSTRIDE}; LdsDual dw{cur_mem + 2u * k * PSTRIDE}; (void)cw; (void)dw;\n"
          << ind << "      " << subst("k", "cw", "dw", "(j == 0u)") << "\n"
          << ind << "    }\n" << ind << "  }\n";
    }

    std::string run() {
        const uint32_t N = g.h.n_nodes;
        std::vector<std::string> words = slot_words();
        o << "// generated by re2-modification_amd/csrc/jit_gen.cpp -- do not edit\n#define MFA_PROBE_PERIODS " << knob("MFA_GEN_PROBE_PERIODS", 5) << "u\n#define MFA_SCAN_DEPTH " << knob("MFA_GEN_SCAN_DEPTH", 2) << "\n#define MFA_RUN_DEPTH " << knob("MFA_GEN_RUN_DEPTH", 2) << "\n#ifndef MFA_STATS_BUILD\n#define MFA_STATS_BUILD 0\n#endif\n" << kPrelude;
        o << "\n#define REV " << (rev ? "true" : "false") << "\n#define N_WORDS " << words.size() << "\n#define N_KEYS " << (N - 1) << "\n\n";
        const bool huge = jit_slot_registers(g) > 272;
        const uint32_t lanes = huge ? huge_lanes((uint32_t)words.size()) : 64u;
        const uint32_t stride = huge ? lanes + 1 : 64u;
        const bool lds_next = huge || (int)words.size() > knob("MFA_GEN_LDS_NEXT_MIN", 24);
        o << "#define LANES " << lanes << "u\n#define PSTRIDE " << stride << "u\n#define HUGE " << (huge ? 1 : 0) << "\n";
        o << "#define NEXT_IN_LDS " << (lds_next ? 1 : 0) << "\n";
        // LDS-resident words: [word][v|d][column], one bank per column
        o << "struct LdsDual {\n  uint32_t* p;\n"
             "  __device__ __forceinline__ operator Dual() const { return Dual{p[0], (int32_t)p[PSTRIDE]}; }\n"
             "  __device__ __forceinline__ LdsDual& operator=(Dual d) { p[0] = d.v; p[PSTRIDE] = (uint32_t)d.d; return *this; }\n"
             "  __device__ __forceinline__ LdsDual& operator=(const LdsDual& o) { return *this = (Dual)o; }\n};\n";
        o << "struct LdsPlain {\n  uint32_t* p;\n"
             "  __device__ __forceinline__ operator uint32_t() const { return p[0]; }\n"
             "  __device__ __forceinline__ LdsPlain& operator=(uint32_t v) { p[0] = v; return *this; }\n"
             "  __device__ __forceinline__ LdsPlain& operator=(const LdsPlain& o) { return *this = (uint32_t)o; }\n};\n";
        // conditional assignment of a slot word: a select for registers, a predicated store (no read) for LDS words
        o << "__device__ __forceinline__ void assign_if(uint32_t& x, bool w, uint32_t v) { x = w ? v : x; }\n"
             "__device__ __forceinline__ void assign_if(Dual& x, bool w, Dual v) { x = sel(w, v, x); }\n"
             "__device__ __forceinline__ void assign_if(LdsDual x, bool w, Dual v) { if (w) { x.p[0] = v.v; x.p[PSTRIDE] = (uint32_t)v.d; } }\n"
             "__device__ __forceinline__ void assign_if(LdsPlain x, bool w, uint32_t v) { if (w) x.p[0] = v; }\n";
        auto reg_struct = [&](const char* name, const char* type) {
            o << "template <> struct " << name << "<" << type << "> {\n";
            for (const auto& w : words) o << "  " << type << " " << w << ";\n";
            o << "  static constexpr bool in_lds = false;\n  __device__ __forceinline__ explicit " << name << "(uint32_t*) {}\n};\n";
        };
        auto lds_struct = [&](const char* name, const char* type, const char* proxy) {
            o << "template <> struct " << name << "<" << type << "> {\n";
            for (const auto& w : words) o << "  " << proxy << " " << w << ";\n";
            o << "  static constexpr bool in_lds = true;\n  __device__ __forceinline__ explicit " << name << "(uint32_t* m) :";
            for (size_t k = 0; k < words.size(); k++) o << (k ? ", " : " ") << words[k] << "{m + " << 2 * k << " * PSTRIDE}";
            o << " {}\n};\n";
        };
        // The current set: registers, except for huge automata.  The set a step builds: registers for plain steps of
        // non-huge automata; dual steps of larger automata would need four slot sets in registers at once and spill, so
        // their next set lives in LDS.
        o << "template <class U> struct SlotSet;\ntemplate <class U> struct NextSet;\n";
        if (huge) { lds_struct("SlotSet", "uint32_t", "LdsPlain"); lds_struct("SlotSet", "Dual", "LdsDual"); }
        else { reg_struct("SlotSet", "uint32_t"); reg_struct("SlotSet", "Dual"); }
        if (huge) lds_struct("NextSet", "uint32_t", "LdsPlain"); else reg_struct("NextSet", "uint32_t");
        if (huge) lds_struct("NextSet", "Dual", "LdsDual");
        else if (lds_next) {
            // keys (P words) stay in registers so that the comparison of an insert needs no LDS round trip; S/L/F words in LDS
            o << "template <> struct NextSet<Dual> {\n";
            for (const auto& w : words) o << "  " << (w[0] == 'P' ? "Dual" : "LdsDual") << " " << w << ";\n";
            o << "  static constexpr bool in_lds = true;\n  __device__ __forceinline__ explicit NextSet(uint32_t* m) :";
            bool first = true;
            size_t kk = 0;                       // LDS words are numbered without the keys
            for (size_t k = 0; k < words.size(); k++) {
                if (words[k][0] == 'P') continue;
                o << (first ? " " : ", ") << words[k] << "{m + " << 2 * kk++ << " * PSTRIDE}";
                first = false;
            }
            o << " {}\n};\n";
        } else reg_struct("NextSet", "Dual");
        o << "\n";
        if (huge) o << "__shared__ uint32_t huge_lds[4 * N_WORDS * PSTRIDE];      // cur, next (v and d each)\n__shared__ Input huge_in[LANES + 1];\nstatic_assert(sizeof(Input) <= 160 && MFA_RT_CACHED == 2u, \"huge_lanes() budget\");\n\n";
        if (huge) emit_step_chunked(); else emit_step();
        const bool wl = huge && sparse;      // probe bookkeeping loops over occupied nodes only (each_word)
        // ---- kernel
        // small automata: ask for two waves per SIMD (<= 128 VGPRs); the plain step needs far fewer, only the dual
        // step is register hungry and may then spill a little -- it is rare
        const int lb_waves = knob("MFA_GEN_WAVES", !huge && (int)words.size() <= knob("MFA_GEN_LB2_WORDS", 20) ? 2 : 0);
        o << "extern \"C\" __global__ void __launch_bounds__(64" << (lb_waves > 0 ? ", " + num(lb_waves) : std::string()) << ")\nmfa_jit_kernel(const uint8_t* __restrict__ bytes, "
             "const uint64_t* __restrict__ offsets, uint64_t n,\n               uint8_t* __restrict__ results, "
             "unsigned long long* counter, uint32_t accel, uint32_t* __restrict__ scratch, unsigned long long* stats_arg,\n"
             "               const uint64_t* __restrict__ regions) {\n";
        // the counters cost a dozen VGPRs: they exist only in objects compiled with -DMFA_STATS_BUILD=1 (MFA_STATS=1)
        o << "  unsigned long long* const stats = MFA_STATS_BUILD ? stats_arg : nullptr;\n";
        o << "  unsigned long long st_iter = 0, st_dual = 0, st_skip = 0, st_probe = 0, st_hit = 0, st_scan = 0; uint32_t st_steps = 0;\n";
        o << "  unsigned long long tm_scan = 0, tm_plain = 0, tm_dual = 0, tm_ticket = 0, tm_byte = 0, tm_look = 0, tm_total = stats ? clock64() : 0;\n";
        o << "  const uint32_t lane = threadIdx.x & 63u;\n";
        o << "  const uint32_t col = lane < LANES ? lane : LANES;     // column of this lane in the LDS images (idle lanes share one)\n";
        o << "#if HUGE\n  uint32_t* const cur_mem = huge_lds + col;\n"
             "  uint32_t* const nxt_mem = cur_mem + 2 * N_WORDS * PSTRIDE;\n"
             "#elif NEXT_IN_LDS\n  __shared__ uint32_t nxt_lds[2 * (N_WORDS - N_KEYS) * PSTRIDE];\n  uint32_t* const cur_mem = nullptr;\n  uint32_t* const nxt_mem = nxt_lds + col;\n"
             "#else\n  uint32_t* const cur_mem = nullptr;\n  uint32_t* const nxt_mem = nullptr;\n#endif\n";
        o << "  // probe storage of this wave, [array][word][column]: SB = slots at probe start, from the dual period on the slots at its\n"
             "  // start; SA = the direction d measured over the first period; SD = direction carried between dual steps\n";
        // small automata keep it in LDS, larger ones in an L2-resident scratch buffer (huge ones: LDS is what limits their lanes)
        const bool probe_lds = words.size() <= 68;
        if (!huge && probe_lds) {
            // directions are small numbers: 16 bits each (a probe whose direction does not fit is abandoned).  SD is only
            // live between two dual steps and the LDS next set only inside one: they share memory.
            // LDS budget for two waves per SIMD: 20 KiB a wave.  The next set comes first, then SA, then SB; what does not
            // fit lives in the wave's (L2-resident) scratch area and is touched a few times per probe only.
            const size_t nkeys = (size_t)std::count_if(words.begin(), words.end(), [](const std::string& w) { return w[0] == 'P'; });
            size_t lds = lds_next ? 2 * (words.size() - nkeys) * 256 : words.size() * 128;
            const size_t budget = (size_t)knob("MFA_GEN_LDS_BUDGET", 65536);
            const bool sa_lds = lds + words.size() * 128 <= budget;
            if (sa_lds) lds += words.size() * 128;
            const bool sb_lds = sa_lds && lds + words.size() * 256 <= budget;
            if (sb_lds) o << "  __shared__ uint32_t probe_sb[N_WORDS * 64];\n";
            else o << "  uint32_t* const probe_sb = scratch + (size_t)blockIdx.x * (3u * N_WORDS * 64u);\n";
            if (sa_lds) o << "  __shared__ int16_t probe_sa[N_WORDS * 64];\n";
            else o << "  int16_t* const probe_sa = reinterpret_cast<int16_t*>(probe_sb + N_WORDS * 64);\n";
            o << "#if NEXT_IN_LDS\n  int16_t* const probe_sd = reinterpret_cast<int16_t*>(nxt_lds);\n#else\n  __shared__ int16_t probe_sd[N_WORDS * 64];\n#endif\n";
            o << "  (void)scratch;\n"
                 "#define SA_RD(k) ((int32_t)probe_sa[(k) * 64 + col])\n#define SA_WR(k, v) (probe_sa[(k) * 64 + col] = (int16_t)(v))\n"
                 "#define SD_RD(k) ((int32_t)probe_sd[(k) * 64 + col])\n#define SD_WR(k, v) (probe_sd[(k) * 64 + col] = (int16_t)(v))\n"
                 "#define SB_RD(k) (probe_sb[(k) * 64 + col])\n#define SB_WR(k, v) (probe_sb[(k) * 64 + col] = (v))\n";
        } else {
            o << "  uint32_t* const SAm = scratch + (size_t)blockIdx.x * (3u * N_WORDS * 64u) + col;\n  uint32_t* const SBm = SAm + N_WORDS * 64u;\n"
                 "  uint32_t* const SDm = SBm + N_WORDS * 64u;\n"
                 "#define SA_RD(k) ((int32_t)SAm[(k) * 64])\n#define SA_WR(k, v) (SAm[(k) * 64] = (uint32_t)(v))\n"
                 "#define SD_RD(k) ((int32_t)SDm[(k) * 64])\n#define SD_WR(k, v) (SDm[(k) * 64] = (uint32_t)(v))\n"
                 "#define SB_RD(k) (SBm[(k) * 64])\n#define SB_WR(k, v) (SBm[(k) * 64] = (v))\n";
        }
        o << "#if HUGE\n  Input& in = huge_in[col];\n#else\n  Input in;\n#endif\n";
        o << "  in.bytes = bytes; in.total16 = (offsets[n] + 15u) & ~(uint64_t)15; input_reset(in, 0, 0);\n";
        o << "  in.w0 = in.w1 = in.w2 = in.w3 = in.p0 = in.p1 = in.p2 = in.p3 = 0;\n";
        o << "  bool active = false, exhausted = false, accept = false;\n  uint32_t i = 0, len = 0; uint64_t sid = 0;\n";
        o << "  // run acceleration: phase 0 idle, 1 = pp plain steps after saving the slots, 2 = pp dual steps\n";
        o << "  uint32_t phase = 0, probe_at = 0, backoff = 8, pp = 1, pk = 0, fails = 0, mult = 1, nper = 0;\n  tb_t TBacc = tb_init();\n"
             "  bool fits = true;        // every direction of the running probe fits its 16-bit store\n"
             "  bool stable = false;     // the last two plain periods moved the slots by the same amounts\n"
             "  bool patient = false;    // a dual period right after the first plain one has failed on this string: wait for two equal movements\n"
             "  unsigned long long st_f_unst = 0, st_f_dual = 0, st_f_room = 0;\n";
        o << "  __shared__ uint64_t rt_cache[MFA_RT_CACHED * (HUGE ? LANES + 1u : 64u)];      // first entries of every lane's region table\n"
             "  bool first_round = true;\n  uint32_t warm = 0, turn = 0;\n"
             "  unsigned long long occ0 = 0ull, occ1 = 0ull;      // huge automata: slots that may be occupied in some lane (emit_step_chunked)\n"
             "  unsigned long long pocc0 = 0ull, pocc1 = 0ull;    // ... in some lane at some step since the running probes began (each_word)\n"
             "#if HUGE\n  for (uint32_t k = 0; k < N_WORDS; k++) { nxt_mem[2u * k * PSTRIDE] = MFA_EMPTY; nxt_mem[(2u * k + 1u) * PSTRIDE] = 0u; }      // the next set starts empty\n#endif\n";
        o << "  SlotSet<uint32_t> c(cur_mem);\n";
        for (const auto& w : words) o << "  c." << w << " = " << (w[0] == 'P' ? "MFA_EMPTY" : "0u") << ";\n";
        o << "  for (;;) {\n    const unsigned long long tmA = stats ? clock64() : 0;\n";
        o << "    {\n      // hand strings to idle lanes: one atomic per wave, tickets dealt by lane rank\n"
             "      const bool want = !active && !exhausted && lane < LANES;\n      const unsigned long long wb = __ballot(want);\n"
             "      if (wb) {\n        unsigned long long first = 0;\n"
             "        occ" << (g.h.start < 64 ? 0 : 1) << " |= 1ull << " << (g.h.start & 63) << ";                       // new strings start in the start slot\n"
             "        if (first_round) first = (unsigned long long)blockIdx.x * LANES;      // the first strings of a wave need no ticket\n"
             "        else {\n"
             "          if (lane == (uint32_t)__builtin_ctzll(wb)) first = atomicAdd(counter, (unsigned long long)__builtin_popcountll(wb));\n"
             "          first = ((unsigned long long)__shfl((uint32_t)(first >> 32), __builtin_ctzll(wb)) << 32) | __shfl((uint32_t)first, __builtin_ctzll(wb));\n"
             "          first += (unsigned long long)gridDim.x * LANES;\n"
             "        }\n"
             "        first_round = false;\n"
             "        if (want) {\n          sid = first + (unsigned long long)__builtin_popcountll(wb & ((1ull << lane) - 1ull));\n"
             "          if (sid >= n) exhausted = true;\n          else {\n            uint4 rta, rtb;\n            rt_fetch(regions, sid, rta, rtb);      // the table row and the offsets travel together\n            const uint64_t b = offsets[sid], e = offsets[sid + 1];\n"
             "            if (e - b > MFA_DEV_MAX_LEN) results[sid] = 2;\n            else {\n"
             "              len = (uint32_t)(e - b); input_reset(in, b, len); rt_attach(in, regions, sid, rt_cache + col, HUGE ? LANES + 1u : 64u, warm, rta, rtb);\n"
             "              i = 0; accept = false; active = true; phase = 0; probe_at = 0; backoff = 8; pp = 1; fails = 0; mult = 1; nper = 0; stable = false; patient = false;\n";
        for (const auto& w : words)
            o << "              c." << w << " = " << (w == "P" + num(g.h.start) ? "0u" : (w[0] == 'P' ? "MFA_EMPTY" : "0u")) << ";\n";
        o << "            }\n          }\n        }\n      }\n    }\n    if (!__any(active)) break;\n    st_iter++;\n";
        o << "    const uint32_t tr_i = i, tr_phase = phase, tr_pp = pp, tr_nper = nper; const bool tr_active = active;   // MFA_STATS builds: trace of the first strings\n";
        o << "    const bool final_pass = (i == len);\n    uint32_t ch = 0x100u;\n"
             "    const unsigned long long tmB = stats ? clock64() : 0;\n"
             "    if (turn == 0u) window_turn<REV>(in, i, active && !final_pass);      // the byte windows of all lanes are renewed together\n"
             "    if (active && !final_pass) ch = stream_byte<REV>(in, i, 16u - turn);\n"
             "    turn = (turn + 1u) & 15u;\n"
             "    if (stats) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); tm_byte += clock64() - tmB; }\n"
             "    const unsigned long long tmC = stats ? clock64() : 0;\n";
        // decide whether this lane starts a probe: it must sit in a long run of equal bytes
        o << "    // does this lane sit at the start of a block that looks periodic?  then find how far the periodic region goes\n"
             "    uint32_t q = 0u;\n"
             "    const bool ep_busy = __any(phase != 0u);      // probes run in epochs: all lanes that probe do it in the same iterations\n"
             "    if (!ep_busy) { pocc0 = occ0; pocc1 = occ1; } else { pocc0 |= occ0; pocc1 |= occ1; }\n"
             "    if (accel && active && !final_pass && phase == 0u && i >= probe_at && !ep_busy) {\n"
             "      if (in.per_q != 0u && in.per_lo <= i && i < in.per_hi) q = in.per_q;      // still inside the region found last\n"
             "      else if (in.rt != nullptr) {                  // regions come from the pre-pass table (regions.hip): nothing to measure\n"
             "        uint32_t rl, rh, rq, rn;\n"
             "        if (rt_find<REV>(in, i, mult, rl, rh, rq, rn)) {\n"
             "          if (in.per_q != 0u) { in.prev_lo = in.per_lo; in.prev_hi = in.per_hi; in.prev_q = in.per_q; }\n"
             "          in.per_lo = rl; in.per_hi = rh; in.per_q = rq; q = rq;\n"
             "        } else probe_at = rn;                       // look again where the next region starts (never, if there is none)\n"
             "      } else probe_at = ~0u;                        // no table: every step is executed\n"
             "    }\n"
             "    if (stats) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); tm_look += clock64() - tmC; }\n"
             "    const unsigned long long tm0 = stats ? clock64() : 0;\n"
             "    if (stats) tm_scan += clock64() - tm0;\n"
             "    if (q == 1u) { in.run_lo = i; in.run_hi = in.per_hi; in.run_ch = ch; }\n"
             "    if (q != 0u && q * mult > 16u) mult = 1u;\n"
             "    const unsigned long long cand = __ballot(q != 0u && in.per_hi - i >= 4u * q * mult + 24u);\n"
             "    const uint32_t ep_pp = cand ? __shfl(q * mult, __builtin_ctzll(cand)) : 0u;     // the first candidate's period leads the epoch\n"
             "    if (q != 0u) {\n"
             "      pp = ep_pp;\n"
             "      if (ep_pp != 0u && ep_pp % q == 0u && in.per_hi - i >= 4u * pp + 24u) {\n";
        each_word(words, wl, "        ", "SB_WR(K_, (uint32_t)CW_);");
        o << "        phase = 1u; pk = 0u; nper = 0u; stable = false; st_probe++;\n      } else if (in.per_hi - i < 4u * q * mult + 24u) {\n"
             "        probe_at = in.per_hi > i + 1u ? in.per_hi : i + 1u;      // region too short to be worth a probe: look again behind it\n"
             "      } else {\n        probe_at = i + 1u;                                       // does not fit this epoch's period: next epoch\n      }\n    }\n";
        o << "    bool any_next = false;\n    tb_t TB = tb_init();\n    const unsigned long long tm1 = stats ? clock64() : 0;\n";
        o << "    if (__any(phase == 2u)) {\n";
        o << "      // dual step: lanes in phase 2 carry the direction saved in SD, the others d = 0 (their TB is ignored)\n";
        o << "      const bool p2 = phase == 2u;\n      SlotSet<Dual> dc(cur_mem);\n      st_dual++;\n";
        each_word(words, wl, "      ", "DW_ = Dual{(uint32_t)CW_, p2 ? SD_RD(K_) : 0};");
        o << "      const Dual di{i, (int32_t)pp}, dlen{len, 0};\n";
        o << "      in.dual_p = p2 ? pp : 0u;\n";
        o << "      (void)lt(di, Dual{p2 ? in.per_hi : i + 1u, 0}, TB);     // the byte at this step of the period repeats while i is inside the periodic region\n";
        o << "      (void)eq(di, dlen, TB);\n";
        o << "      mfa_step<Dual>(dc, in, di, dlen, ch, final_pass, accept, any_next, TB, cur_mem, nxt_mem, active, occ0, occ1);\n      pocc0 |= occ0; pocc1 |= occ1;\n";
        o << "      in.dual_p = 0u;\n";
        o << "      uint32_t skip = 0;\n";
        o << "      if (p2) {\n        tb_min(TBacc, TB.a, TB.b);\n        pk++;\n";
        o << "        if (pk == pp) {\n          const int64_t periods = tb_steps(TBacc);\n          bool same = !accept && any_next && periods > 1 && fits;\n";
        // no short-circuit in the comparisons over the words: a chain of `&&` is a chain of branches with one probe-image read (LDS, or
        // the L2-resident scratch area) waited for in each -- the differences are or-ed together instead and the reads go out together
        o << "          if (same) {\n            uint32_t differs = 0u;\n";
        each_word(words, wl, "            ", "const Dual t = DW_; const int32_t sa = SA_RD(K_); differs |= (uint32_t)(t.d ^ sa) | (uint32_t)((int32_t)(t.v - SB_RD(K_)) ^ sa);");
        o << "            same = differs == 0u;\n          }\n";
        o << "          if (same) skip = (uint32_t)(periods - 1 < (int64_t)0x00ffffff ? periods - 1 : (int64_t)0x00ffffff);\n";
        o << "          phase = 0u;\n";
        o << "          if (skip) { backoff = 8u; fails = 0u; st_hit++; st_skip += (unsigned long long)skip * pp; }\n";
        // a failed dual period: after an optimistic start (one plain period) the next probe of this string waits for two
        // equal movements; otherwise the slots may repeat with a multiple of the period
        o << "          else { st_f_dual++; fails++; if (nper == 1u && !patient && pp > 2u) patient = true; else mult = mult % 8u + 1u;\n"
             "            if (fails >= 8u) { fails = 0u; backoff = backoff < 4096u ? backoff * 2u : backoff; } }\n";
        o << "        } else if (tb_is_one(TBacc) || accept || !any_next) {\n"
             "          phase = 0u; fails++; st_f_dual++;                       // cannot succeed any more: stop the probe here\n"
             "          if (nper == 1u && !patient && pp > 2u) patient = true; else mult = mult % 8u + 1u;\n"
             "          if (fails >= 8u) { fails = 0u; backoff = backoff < 4096u ? backoff * 2u : backoff; }\n"
             "        } else {\n";
        each_word(words, wl, "          ", "const int32_t d = Dual(DW_).d; SD_WR(K_, d); fits &= (d == (int32_t)(int16_t)d);");
        o << "        }\n      }\n";
        each_word(words, wl, "      ", "const Dual t = DW_; CW_ = t.v + skip * (uint32_t)t.d;");
        o << "      if (skip) { i += skip * pp; input_drop_window(in); probe_at = i + 1u + pp; }\n";
        o << "      else if (p2 && phase == 0u) probe_at = i + (fails ? 1u : backoff);\n";
        o << "      if (phase == 1u) pk++;\n      if (stats) tm_dual += clock64() - tm1;\n";
        o << "    } else {\n";
        o << "      mfa_step<uint32_t>(c, in, i, len, ch, final_pass, accept, any_next, TB, cur_mem, nxt_mem, active, occ0, occ1);\n      pocc0 |= occ0; pocc1 |= occ1;\n";
        o << "      if (phase == 1u) pk++;\n      if (stats) tm_plain += clock64() - tm1;\n";
        o << "    }\n";
        // plain periods of a probe: after each one the movement of the slots over the period is compared with the previous
        // period's; a lane is ready for the dual period once two consecutive movements agree.  All lanes of an epoch reach
        // their period boundaries in the same iteration and go on together: to the dual period when every one of them is
        // ready or has run out of patience (MFA_PROBE_PERIODS plain periods) or of periodic input.
        o << "    if (phase == 1u && pk == pp) {\n      bool eqd = nper != 0u, occ = nper == 0u && !patient && pp > 2u;      // short periods: waiting for a second one costs next to nothing\n      uint32_t moved = 0u, wide = 0u, vacated = 0u;\n";
        each_word(words, wl, "      ", "const uint32_t v = CW_, b = SB_RD(K_); const int32_t d = (int32_t)(v - b); moved |= (uint32_t)(d ^ SA_RD(K_)); SA_WR(K_, d); "
                                     "SB_WR(K_, v); wide |= (uint32_t)(d ^ (int32_t)(int16_t)d); if (ISP_) vacated |= (uint32_t)((v == MFA_EMPTY) != (b == MFA_EMPTY));");
        // first period: the same slots occupied before and after it is taken as "probably settled already"
        o << "      eqd = eqd && moved == 0u; occ = occ && vacated == 0u; fits = wide == 0u;\n";
        o << "      nper++; pk = 0u; stable = (eqd || occ) && fits;\n    }\n";
        o << "    {\n      const bool at_b = phase == 1u && pk == 0u && nper != 0u;\n"
             "      if (__any(at_b)) {\n"
             "        const bool room = in.per_hi >= i + 1u + 2u * pp;          // the dual period and at least one more to skip\n"
             "        if (!__any(at_b && !stable && room && nper < (pp > 2u ? MFA_PROBE_PERIODS : 3u))) {\n"
             "          if (at_b && stable && room) {\n";
        each_word(words, wl, "            ", "SD_WR(K_, SA_RD(K_));");
        o << "            phase = 2u; TBacc = tb_init();\n"
             "          } else if (at_b) {\n"
             "            phase = 0u;\n"
             "            if (!room) st_f_room++; else st_f_unst++;\n"
             "            if (!room) probe_at = in.per_hi > i + 1u ? in.per_hi : i + 1u;\n"
             "            else {                                                 // never settled: maybe the slots repeat with a multiple of the period\n"
             "              fails++; mult = mult % 8u + 1u;\n"
             "              if (fails >= 8u) { fails = 0u; backoff = backoff < 4096u ? backoff * 2u : backoff; }\n"
             "              probe_at = i + 1u + (fails ? 0u : backoff);\n"
             "            }\n          }\n        }\n      }\n    }\n";
        o << "    if (stats && tr_active && sid < 4u && st_steps < 8192u) {\n"
             "      uint32_t* tr = (uint32_t*)(stats + 16) + ((1u << 20) - 65536u) + (uint32_t)sid * 16384u + 2u * st_steps;\n"
             "      tr[0] = tr_i; tr[1] = tr_phase | (q << 4) | (tr_pp << 8) | (tr_nper << 16) | ((uint32_t)__any(tr_phase == 2u) << 24) | 0x80000000u;\n"
             "    }\n";
        o << "    if (active) {\n      const bool done = accept || final_pass || !any_next;\n      i++;\n"
             "      st_steps++;\n      if (done) { results[sid] = (warm == 0x9e3779b9u && len == 0xffffffffu) ? 3 : (accept ? 1 : 0); active = false; phase = 0u;\n"
             "        if (stats && sid < (1u << 20)) ((uint32_t*)(stats + 16))[sid] = st_steps;\n        st_steps = 0;\n";
        for (const auto& w : words)
            if (w[0] == 'P') o << "        c." << w << " = MFA_EMPTY;\n";
        o << "      }\n    }\n  }\n";
        o << "  if (stats) {\n    if (lane == 0) { atomicAdd(&stats[0], st_iter); atomicAdd(&stats[1], st_dual); atomicAdd(&stats[6], tm_scan); atomicAdd(&stats[7], tm_plain);\n"
             "      atomicAdd(&stats[8], tm_dual); atomicAdd(&stats[9], (unsigned long long)clock64() - tm_total);\n"
             "      atomicAdd(&stats[13], tm_ticket); atomicAdd(&stats[14], tm_byte); atomicAdd(&stats[15], tm_look); }\n"
             "    if (lane < LANES) {                             // lanes that never carry a string hold no meaningful counts\n"
             "      atomicAdd(&stats[2], st_skip); atomicAdd(&stats[3], st_probe); atomicAdd(&stats[4], st_hit); atomicAdd(&stats[5], st_scan);\n"
             "      atomicAdd(&stats[10], st_f_unst); atomicAdd(&stats[11], st_f_dual); atomicAdd(&stats[12], st_f_room);\n    }\n  }\n}\n";
        (void)N;
        return o.str();
    }
};

}  // namespace

// VGPRs the slots of a specialised kernel would take; the launcher uses the generic kernel above this
uint32_t jit_slot_registers(const HostImage& img) {
    uint32_t K = img.h.n_cells ? img.h.n_cells : 1;
    return (img.h.n_nodes - 1) * (1 + 3 * K) * 2;
}

// string-carrying lanes per wave of the specialised kernel (64 unless the automaton is huge), 0 = cannot be specialised
uint32_t jit_lanes(const HostImage& img) {
    if (jit_slot_registers(img) <= 272) return 64;
    return huge_lanes(jit_slot_registers(img) / 2);
}

std::string jit_generate_source(const HostImage& img) {
    Gen gen(img);
    return gen.run();
}

}  // namespace mfa
