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

FLAG = {"plain": [], "thompson": ["-thompson"], "glushkov": ["-glushkov"], "mfa": ["-mfa"]}
SUPPORTED = [a for a in MANIFEST["automata"] if a["mode"] in FLAG]
NOT_YET = [a for a in MANIFEST["automata"] if a["mode"] not in FLAG]


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


def test_bnf_and_reverse_fail_loudly(tmp_path):
    """-bnf / -reverse on a regex with memory need the BNF rewriter: not silently ignored."""
    auto = next(a for a in NOT_YET if a["name"] == "ex2_reverse")
    p = run(["-match", "-reverse"], auto["regex"] + "\nab\nexit\n", tmp_path)
    assert p.returncode == 1 and "bnf" in p.stderr.lower()


def test_dot_side_effect(tmp_path):
    """compile() leaves mfa.dot behind like the reference (regex.cpp:311, mfa.cpp:28-61)."""
    run(["-dump"], "({a*}:1&1)*\n", tmp_path)
    dot = (tmp_path / "mfa.dot").read_text()
    assert dot.startswith("digraph g {") and "0 -> 1 [label=\"a/o1/\"]" in dot and "0 -> 3 [label=\"ε/\"]" in dot


def test_example_runner_without_files(tmp_path):
    """`./diploma -match N` with no test/example_N below the working directory: like the reference
    (example_runner.cpp:93) it does nothing and exits normally -- and needs no GPU for that."""
    p = subprocess.run([DIPLOMA, "-match", "7"], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 0 and p.stdout == "" and not (tmp_path / "test").exists()
