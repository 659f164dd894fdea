#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Needs oracle/_ref/ref_harness (the reference's own sources compiled in place from
/root/reference by oracle/Makefile, canonical allocation-order mode) -- so it only
runs in the build container.  What it writes is data only:

  manifest.json                 automata (name, regex, mode) and string sets
  images/<name>.dump            automaton image text dump (reference graph after compile())
  strings/<set>.txt             one input string per line ('' = the empty string)
  results/<name>.<set>.bits     the reference's 0/1 answers, one character per string
  front/bnf_reverse.txt         regex<TAB>BNF<TAB>Reverse as printed by the reference REPL body

Usage: python tests/golden/make_golden.py   (from the repo root)
"""
import itertools
import json
import os
import random
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

# README.md:81-92 / test/example_N/regexp.txt line 1, pump.txt (parts, suffix, prefix)
EXAMPLES = {
    1: ("({a*}:1&1)*", ["a"], "b", ""),
    2: ("{(a|bb)*}:1aaba(&1|bb*aa)*", ["bbaa", "aaba", "bbaa"], "c", ""),
    3: ("{{a*}:1(&1)*}:2b&2a*", ["a", "b", "a"], "aab", ""),
    4: ("({a*}:1&1a*)*", ["a"], "b", ""),
    5: ("{a*}:1c{&1}:2c(&1|&2)*", ["aa"], "b", "aacaac"),
    6: ("({a*}:1b|&1)*", ["a"], "c", "aaab"),
    7: ("({a*}:1)*b&1", ["a", "b", "a"], "b", ""),
    8: ("(({a*}:1|b)(&1|b))*", ["a", "b", "a"], "c", "bb"),
    9: ("(({aa*b}:1(&1)*)|b(b|a*)*)*", ["bbaaa"], "c", ""),
    10: ("({a*}:1b|b&1)*c&1", ["aababba"], "cab", ""),
    # test/example_11..17: regexp.txt line 1, pump.txt (no README row, no timing files)
    11: ("({a*}:1b&1b)*", ["aa", "b", "aa"], "bbc", ""),
    12: ("(({a*}:1b&1b)*)*", ["aa", "b", "aa"], "bbc", ""),
    13: ("ba{aa*}:1a&1*", ["a", "a", "a"], "b", "ba"),
    14: ("{(a*|b*)}:1b(&1|b)*", ["b", "b", "b", "b", "b"], "bc", ""),
    15: ("{a*}:1c{a*}:2c(&1|&2)*", ["aa"], "b", "aacaac"),
    16: ("({a*}:1&1|(a*|b)a)*", ["baaaa"], "b", ""),
    17: ("(&1{a*}:1|(a*|b)a)*", ["baaaa"], "b", ""),
}

# regexes whose side-effect files (the *.dot files compile() leaves in the working directory, regex.cpp:287,294,311,331; the
# rewrite trace `-log` writes to log.txt, bnf.cpp:10,894-897) are kept byte for byte: (name, regex, mode)
SIDE = [("ex1_plain", EXAMPLES[1][0], "plain"), ("ex3_reverse", EXAMPLES[3][0], "reverse"), ("ex5_bnf", EXAMPLES[5][0], "bnf"),
        ("ex6_reverse", EXAMPLES[6][0], "reverse"), ("ex10_plain", EXAMPLES[10][0], "plain"), ("nfa_abb_plain", "(a|b)*abb", "plain"),
        ("ex14_reverse", EXAMPLES[14][0], "reverse")]
LOGGED = ["{a*}:1c{&1}:2c(&1|&2)*", "{{a*}:1(&1)*}:2b&2a*", "({a*}:1b|&1)*", "(({a*}:1|b)(&1|b))*", "({a*}:1b|b&1)*c&1"]

# extra memory regexes (test/bnf_examples.txt column 1, test_inputs.txt) -- plain mode only
EXTRA_MFA = {
    "x1": "({a*}:1|{b*}:1|c)&1",
    "x2": "({a*b}:1)*b&1",
    "x3": "(&1{(a|b)}:1)*",
    "x4": "({a}:1b(&1b)*)*",
    "x5": "{(a|b)}:1(&1b{a*}:1)*",
    "x6": "{a*}:1(&1{a*}:1)*",
    "x7": "{a*}:1b({b*}:2&1{a*}:1&2)*",
    "x8": "{a*}:1b(&1{a*}:1)*",
    "x9": "((&1|b)c(d|{a*}:1&1))*",
    "x10": "((&1|{a}:1&1)b)*",
    "x11": "{.*}:1c&1",
    "x12": "{a*}:1{b*}:2{c*}:3&3&2&1",
}

# memory-less regexes: (regex, [modes]); plain = what compile() picks (regex.cpp:315-342)
# ssnf / all: compile() with the star normal form (bt/bt_ssnf.cpp; regex.cpp:325-334) / with -all = -bnf -reverse -ssnf (main.cpp:25-29)
NFA = {
    "abb": ("(a|b)*abb", ["plain", "thompson", "glushkov", "ssnf", "all"]),
    "third": ("(a|b)*a(a|b)(a|b)", ["plain", "thompson", "glushkov", "ssnf"]),
    "dot": ("a.c*(b|.a)*", ["plain", "thompson", "glushkov", "ssnf"]),
    "enum": ("[a-c]*abc", ["plain", "glushkov", "ssnf"]),
    "alt3": ("(ab|b)(ab|ba)*c*", ["plain", "thompson", "glushkov", "ssnf"]),
    "star1": ("(a*b*)*ab", ["plain", "ssnf"]),
    "star2": ("((a*)*|b*)*a", ["plain", "ssnf"]),
    "star4": ("((ab)*c*)*(a|b)", ["plain", "ssnf"]),
}


def pumped_string(n, pump):
    """matchers/example_runner.cpp:15-29 (== matcher.py:26-38)."""
    pump_count = len(pump) // 2 + 1
    del_count = len(pump) - pump_count
    res = pump[0]
    while len(res) + len(pump[0]) < (n - del_count) // pump_count:
        res += pump[0]
    out = ""
    for _ in range(del_count):
        out += res + pump[1]
    return out + res


def all_strings(alphabet, max_len):
    out = [""]
    for n in range(1, max_len + 1):
        out += ["".join(t) for t in itertools.product(alphabet, repeat=n)]
    return out


def string_sets():
    sets = {}
    sets["abc7"] = all_strings("abc", 7)
    rng = random.Random(0x5EED0001)
    rnd = []
    for k in range(600):
        n = rng.randint(8, 160)
        p = rng.choice([(0.5, 0.3), (0.8, 0.15), (0.34, 0.33), (0.95, 0.04)])
        s = "".join("a" if (x := rng.random()) < p[0] else ("b" if x < p[0] + p[1] else "c") for _ in range(n))
        rnd.append(s)
    # repeated-block strings: long successful cell reads
    for k in range(200):
        blk = "".join(rng.choice("aab") for _ in range(rng.randint(1, 6)))
        reps = rng.randint(2, 40)
        s = blk * reps
        if rng.random() < 0.5:
            s += rng.choice(["b", "c", "ab", "ba", "a"])
        if rng.random() < 0.3:
            s = rng.choice(["b", "c", "aab"]) + s
        rnd.append(s)
    sets["rnd"] = rnd
    # bytes that collide with label syntax: digits and '.' in the INPUT (mfa.cpp:171 compares the raw label first)
    sets["odd"] = ["1", "a1", "11", "a1a", "1a", ".", "a.a", "aa1aa", "a11", "9", "ab1", "1b", "d", "dd", "cdc", "bcd",
                   "adc", "z", "a.c", "abcd"] + ["".join(t) for t in itertools.product("a1.", repeat=4)]
    for ex, (regex, pump, suffix, prefix) in EXAMPLES.items():
        ps = []
        sizes = list(range(1, 40)) + [50, 64, 100, 128, 200, 256, 300, 400, 512, 700, 1000, 1500, 2048]
        for n in sizes:
            core = pumped_string(n, pump)
            ps.append(prefix + core + suffix)
            ps.append(prefix + core)
            ps.append(core + suffix)
        for n in (16, 64, 200):
            core = prefix + pumped_string(n, pump)
            for _ in range(12):
                k = rng.randrange(len(core))
                ps.append(core[:k] + rng.choice("abc") + core[k + 1:] + (suffix if rng.random() < 0.5 else ""))
        sets["pump%d" % ex] = ps
    return sets


def run(args, stdin_text=None, cwd=None):
    p = subprocess.run([HARNESS] + args, input=stdin_text, capture_output=True, text=True, cwd=cwd)
    if p.returncode != 0:
        raise RuntimeError("%s failed: %s" % (args, p.stderr))
    return p.stdout


def main():
    import sys
    new_only = "--new-only" in sys.argv      # keep the answers that exist (they are the reference's: regenerating them changes nothing)
    if not os.path.exists(HARNESS):
        raise SystemExit("build oracle/_ref/ref_harness first (make -C oracle ref)")
    for d in ("images", "strings", "results", "front", "side"):
        os.makedirs(os.path.join(HERE, d), exist_ok=True)
    sets = string_sets()
    for name, strs in sets.items():
        with open(os.path.join(HERE, "strings", name + ".txt"), "w") as f:
            f.write("".join(s + "\n" for s in strs))
    automata = []
    for ex, (regex, pump, suffix, prefix) in EXAMPLES.items():
        for mode in ("plain", "bnf", "reverse"):
            automata.append({"name": "ex%d_%s" % (ex, mode), "regex": regex, "mode": mode,
                             "sets": ["abc7", "rnd", "odd", "pump%d" % ex]})
    for key, regex in EXTRA_MFA.items():
        automata.append({"name": "%s_plain" % key, "regex": regex, "mode": "plain", "sets": ["abc7", "rnd", "odd"]})
    for key, (regex, modes) in NFA.items():
        for mode in modes:
            automata.append({"name": "nfa_%s_%s" % (key, mode), "regex": regex, "mode": mode,
                             "sets": ["abc7", "rnd", "odd"]})
    with tempfile.TemporaryDirectory() as tmp:       # compile() drops *.dot files into cwd
        old = {}
        if new_only and os.path.exists(os.path.join(HERE, "manifest.json")):
            with open(os.path.join(HERE, "manifest.json")) as f:
                old = {a["name"]: a for a in json.load(f)["automata"]}
        for a in automata:
            if a["name"] in old and all(os.path.exists(os.path.join(HERE, "results", "%s.%s.bits" % (a["name"], s))) for s in a["sets"]):
                a["header"] = old[a["name"]]["header"]
                continue
            dump = run(["dump", a["mode"], a["regex"]], cwd=tmp)
            a["header"] = run(["header", a["mode"], a["regex"]], cwd=tmp)
            with open(os.path.join(HERE, "images", a["name"] + ".dump"), "w") as f:
                f.write(dump)
            for s in a["sets"]:
                text = "".join(x + "\n" for x in sets[s])
                out = run(["match", a["mode"], a["regex"]], stdin_text=text, cwd=tmp)
                bits = out.replace("\n", "")
                assert len(bits) == len(sets[s]), (a["name"], s, len(bits), len(sets[s]))
                with open(os.path.join(HERE, "results", "%s.%s.bits" % (a["name"], s)), "w") as f:
                    f.write(bits + "\n")
            print(a["name"], "ok", flush=True)
        # side-effect files of compile(), and the -log trace
        for name, regex, mode in SIDE:
            with tempfile.TemporaryDirectory() as side:
                run(["dump", mode, regex], cwd=side)
                for fn in sorted(os.listdir(side)):
                    with open(os.path.join(side, fn), "rb") as f, open(os.path.join(HERE, "side", "%s.%s" % (name, fn)), "wb") as g:
                        g.write(f.read())
        for k, regex in enumerate(LOGGED):
            with tempfile.TemporaryDirectory() as side:
                run(["frontlog", regex], cwd=side)
                with open(os.path.join(side, "log.txt"), "rb") as f, open(os.path.join(HERE, "side", "log%d.txt" % k), "wb") as g:
                    g.write(f.read())
        with open(os.path.join(HERE, "side", "logged.txt"), "w") as f:
            f.write("".join(r + "\n" for r in LOGGED))
        # front-end KATs: the REPL body on every regex we use (BNF / Reverse strings)
        lines = []
        seen = set()
        for regex in [e[0] for e in EXAMPLES.values()] + list(EXTRA_MFA.values()):
            if regex in seen:
                continue
            seen.add(regex)
            out = run(["front", regex], cwd=tmp).strip().splitlines()
            lines.append("\t".join([regex] + out))
        with open(os.path.join(HERE, "front", "bnf_reverse.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    manifest = {"automata": automata, "sets": {k: len(v) for k, v in sets.items()},
                "generator": "tests/golden/make_golden.py", "oracle_mode": "bump-arena (allocation order)"}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, ensure_ascii=False)


if __name__ == "__main__":
    main()
