"""Host front-end (re2-modification_amd/host): the automata it builds must be IDENTICAL to the
reference's -- node list order, allocation rank, edge order, labels, cell actions -- because edge and
node order decide tie-breaks at match time.  Golden dumps come from the reference itself."""
import json
import os
import subprocess

import pytest

import oracle_lib

DIPLOMA = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)

FLAG = {"plain": [], "thompson": ["-thompson"], "glushkov": ["-glushkov"], "mfa": ["-mfa"], "bnf": ["-bnf"], "reverse": ["-reverse"], "ssnf": ["-ssnf"],
        "all": ["-all"]}
SUPPORTED = [a for a in MANIFEST["automata"] if a["mode"] in FLAG]
assert len(SUPPORTED) == len(MANIFEST["automata"])          # every fixture automaton: plain, -bnf and -reverse images included


def run(args, text, cwd):
    return subprocess.run([DIPLOMA] + args, input=text, capture_output=True, text=True, cwd=cwd)


@pytest.mark.parametrize("auto", SUPPORTED, ids=lambda a: a["name"])
def test_image_equals_reference(auto, tmp_path):
    p = run(["-dump"] + FLAG[auto["mode"]], auto["regex"] + "\n", tmp_path)
    assert p.returncode == 0, p.stderr
    assert p.stdout == oracle_lib.load_dump(auto["name"])


@pytest.mark.parametrize("auto", [a for a in SUPPORTED if a["mode"] == "plain"], ids=lambda a: a["name"])
def test_compile_header_lines(auto, tmp_path):
    """compile() prints the reference's header lines (regex.cpp:272,301,317,319) before any result."""
    p = run(["-match"], auto["regex"] + "\nexit\n", tmp_path)
    if p.returncode == 0:                       # only reachable on a GPU box; header comes first either way
        assert p.stdout == auto["header"]
    else:
        assert p.stdout == auto["header"]
        assert "no usable HIP device" in p.stderr


def front_cases():
    """tests/golden/front/bnf_reverse.txt: `regex<TAB>BNF: ..<TAB>Reverse: ..` lines produced by the reference's REPL
    (main.cpp:50-85) and the ten `regex,bnf,reverse` lines of the reference's own test/bnf_examples.txt"""
    out = []
    with open(os.path.join(oracle_lib.GOLDEN, "front", "bnf_reverse.txt")) as f:
        for line in f:
            line = line.rstrip("\n")
            if "\t" in line:
                regex, b, r = line.split("\t")
                out.append((regex, b[len("BNF: "):], r[len("Reverse: "):]))
            elif line:
                out.append(tuple(line.split(",")))
    return out


@pytest.mark.parametrize("case", front_cases(), ids=lambda c: c[0])
def test_bnf_and_reverse_known_answers(case, tmp_path):
    """the backreference normal form and its reversal, as strings (regex/bnf.cpp:894-919, regex/reverse.cpp:104-113)"""
    regex, want_bnf, want_rev = case
    p = run([], regex + "\nexit\n", tmp_path)
    assert p.returncode == 0, p.stderr
    assert p.stdout == "BNF: %s\nReverse: %s\n" % (want_bnf, want_rev)


def test_reverse_header_lines(tmp_path):
    """`-match -reverse` on a regex that is not 1-unambiguous prints the normal form and its reversal before any result
    (regex.cpp:279-281); a 1-unambiguous one is matched forwards (regex.cpp:299-313)."""
    case = next(c for c in front_cases() if c[0].startswith("{(a|bb)*}:1"))
    p = run(["-match", "-reverse"], case[0] + "\nexit\n", tmp_path)
    assert p.stdout.split("\n")[1:3] == ["BNF: " + case[1], "Reverse: " + case[2]]
    p = run(["-match", "-reverse"], "({a*}:1&1)*\nexit\n", tmp_path)
    assert "BNF" not in p.stdout and "1-" in p.stdout


SIDE = os.path.join(oracle_lib.GOLDEN, "side")


def logged_regexes():
    with open(os.path.join(SIDE, "logged.txt")) as f:
        return [l.rstrip("\n") for l in f if l.strip()]


@pytest.mark.parametrize("k", range(len(logged_regexes())))
def test_rewrite_trace(k, tmp_path):
    """`-log` leaves the rewrite trace in log.txt (bnf.cpp:10,894-897): byte for byte what the reference writes"""
    regex = logged_regexes()[k]
    p = run(["x", "-log"], regex + "\nexit\n", tmp_path)
    assert p.returncode == 0 and p.stdout.startswith("BNF: ")
    with open(os.path.join(SIDE, "log%d.txt" % k), "rb") as f:
        want = f.read()
    assert (tmp_path / "log.txt").read_bytes() == want


def side_files():
    """tests/golden/side/<automaton>.<file>: the files the reference's compile() leaves in its working directory"""
    out = []
    for fn in sorted(os.listdir(SIDE)):
        if fn.endswith(".dot"):
            name, rest = fn.split(".", 1)
            out.append((name, rest))
    # the goldens are part of the repository (.gitignore excepts them): a checkout without them must not pass empty
    assert len(out) >= 7, "tests/golden/side/*.dot missing: %d found" % len(out)
    return out


@pytest.mark.parametrize("name,fn", side_files())
def test_dot_side_effect(name, fn, tmp_path):
    """compile() leaves mfa.dot / reverse_mfa.dot / reverse.dot behind like the reference (regex.cpp:287,294,311,331;
    mfa.cpp:28-61, automata.cpp:34-66), byte for byte -- the MFA version's epsilon labels included."""
    auto = next(a for a in MANIFEST["automata"] if a["name"] == name)
    p = run(["-dump"] + FLAG[auto["mode"]], auto["regex"] + "\n", tmp_path)
    assert p.returncode == 0, p.stderr
    with open(os.path.join(SIDE, "%s.%s" % (name, fn)), "rb") as f:
        want = f.read()
    assert (tmp_path / fn).read_bytes() == want


def test_example_runner_without_files(tmp_path):
    """`./diploma -match N` with no test/example_N below the working directory: like the reference
    (example_runner.cpp:93) it does nothing and exits normally -- and needs no GPU for that."""
    p = subprocess.run([DIPLOMA, "-match", "7"], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 0 and p.stdout == "" and not (tmp_path / "test").exists()
