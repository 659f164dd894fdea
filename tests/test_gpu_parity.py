"""GPU parity: the HIP path (through the C-ABI) against the golden vectors produced by the
reference and against the CPU restatement, bit-exact 0/1."""
import json
import os

import numpy as np
import pytest

import oracle_lib
from mfa_amd import capi, image

pytestmark = pytest.mark.gpu

with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)


def gpu_match(img, strings):
    import torch
    data, off = oracle_lib.pack(strings)
    d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    d_bytes[:len(data)] = torch.from_numpy(data.copy())
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    res = img.match_tensors(d_bytes, d_off)
    torch.cuda.synchronize()
    return res.cpu().numpy()


def use_engine(monkeypatch, engine):
    """MFA_WALK: "table" = the table-driven walk kernel (walk.hip), "specialised" = the kernel generated for the automaton"""
    monkeypatch.setenv("MFA_WALK", "table" if engine == "table" else "jit")


@pytest.mark.parametrize("jit", ["specialised", "table"])
@pytest.mark.parametrize("auto", MANIFEST["automata"], ids=lambda a: a["name"])
def test_golden(auto, jit, monkeypatch):
    """Both MFA kernels: the table-driven walk (every automaton) and the automaton-specific one, against the reference's answers."""
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    is_mfa = image.blob_info(blob)["kind"] == image.KIND_MFA
    if jit == "table" and not is_mfa:
        pytest.skip("memory-less automata have one kernel")
    use_engine(monkeypatch, jit)
    img = capi.Image(blob)
    for sset in auto["sets"]:
        strings = oracle_lib.load_set(sset)
        want = oracle_lib.load_bits(auto["name"], sset)
        got = gpu_match(img, strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s/%s: %d mismatches, first %r want %d got %d" % (
            auto["name"], sset, bad.size, strings[bad[0]], want[bad[0]], got[bad[0]])
    kern = img.info()["last_kernel"]
    if not is_mfa:
        assert kern == capi.KERNEL_TABLE
    elif jit == "table":
        assert kern == capi.KERNEL_WALK
    else:
        # every fixture automaton has a specialised kernel, except those the generator refuses (hundreds of edges to unroll)
        assert kern == (capi.KERNEL_SPECIALISED if img.specialize() else capi.KERNEL_WALK)
    img.close()


def test_host_entry_point():
    img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    data, off = oracle_lib.pack([b"aa", b"aaa", b"aaaa", b"b", b"aaaaaab", b"ab", b"aaaaaaaa", b""])
    assert list(img.match_host(data, off)) == [1, 1, 1, 0, 0, 0, 1, 1]


def test_cli_match_contract(tmp_path):
    """`./diploma -match`: header lines, then one 0/1 line per token, byte-identical with the reference
    (main.cpp:42-44, matchers/match.cpp:21-31)."""
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    flags = {"plain": [], "bnf": ["-bnf"], "reverse": ["-reverse"], "ssnf": ["-ssnf"], "all": ["-all"]}
    for name in ("ex1_plain", "ex5_plain", "nfa_abb_plain", "nfa_third_plain", "ex3_reverse", "ex6_reverse", "ex2_bnf", "ex1_reverse", "nfa_abb_all",
                 "nfa_star1_ssnf", "ex14_reverse", "ex17_bnf"):
        auto = next(a for a in MANIFEST["automata"] if a["name"] == name)
        strings, want = [], []
        for sset in auto["sets"]:
            bits = oracle_lib.load_bits(name, sset)
            for s, b in zip(oracle_lib.load_set(sset), bits):
                if s:                              # the empty string cannot be a token of `cin >> text`
                    strings.append(s); want.append(b)
        text = auto["regex"].encode() + b"\n" + b"\n".join(strings) + b"\nexit\nnot-read\n"
        p = subprocess.run([diploma, "-match"] + flags[auto["mode"]], input=text, capture_output=True, cwd=tmp_path)
        assert p.returncode == 0, p.stderr
        expect = auto["header"].encode() + b"".join(b"%d\n" % b for b in want)
        assert p.stdout == expect, name


def test_match_file_drivers(tmp_path):
    """`match_mfa` / `match_gt` (matchers/match_mfa.cpp:13-97): strings from a file, one batch on the GPU.  match_mfa prints the
    batch time and then one 0/1 line per string (the reference prints `<seconds> <result>` per string); match_gt prints times
    only, like the reference, and leaves results.txt."""
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    for name in ("ex6_plain", "ex3_plain"):
        auto = next(a for a in MANIFEST["automata"] if a["name"] == name)
        strings, want = [], []
        for sset in auto["sets"]:
            for st, b in zip(oracle_lib.load_set(sset), oracle_lib.load_bits(name, sset)):
                if b"\n" not in st:
                    strings.append(st); want.append(int(b))
        path = tmp_path / (name + ".txt")
        path.write_bytes(b"".join(st + b"\n" for st in strings))
        p = subprocess.run([diploma, "-match-file", "mfa", str(path)], input=auto["regex"].encode() + b"\n", capture_output=True, cwd=tmp_path)
        assert p.returncode == 0, p.stderr
        lines = p.stdout.split()
        assert float(lines[0]) >= 0.0 and [int(x) for x in lines[1:]] == want, name
        assert (tmp_path / "results7.txt").exists()
    auto = next(a for a in MANIFEST["automata"] if a["name"] == "nfa_abb_glushkov")
    path = tmp_path / "gt.txt"
    path.write_bytes(b"abb\naabb\nab\nbbbbabb\n")
    p = subprocess.run([diploma, "-match-file", "gt", str(path)], input=auto["regex"].encode() + b"\n", capture_output=True, cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    assert float(p.stdout.split()[0]) >= 0.0
    # results.txt: times only, `seconds,` like the reference's (match_mfa.cpp:40-41 writes one per string; a batch has one time), and
    # the two drawings it leaves (match_mfa.cpp:18-19)
    import re
    assert re.fullmatch(r"([0-9.eE+-]+,)+", (tmp_path / "results.txt").read_text())
    assert (tmp_path / "glushkov.dot").read_text().startswith("digraph") and (tmp_path / "thomson.dot").read_text().startswith("digraph")


def test_cli_match_mixed(tmp_path):
    """`./diploma -match-mixed FILE...`: several regexes, each with its own strings, matched by ONE device call
    (host/automata_host.cpp: match_mixed -> mfa_match_mixed_host); the answers are the goldens'."""
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    files, want = [], []
    for name in ("ex1_plain", "ex5_plain", "ex9_plain", "x7_plain"):
        auto = next(a for a in MANIFEST["automata"] if a["name"] == name)
        strings = [s for s in oracle_lib.load_set("rnd") if s][:300]
        bits = oracle_lib.load_bits(name, "rnd")
        keep = [k for k, s in enumerate(oracle_lib.load_set("rnd")) if s][:300]
        want += [int(bits[k]) for k in keep]
        path = tmp_path / (name + ".txt")
        path.write_bytes(auto["regex"].encode() + b"\n" + b"".join(s + b"\n" for s in strings))
        files.append(str(path))
    p = subprocess.run([diploma, "-match-mixed"] + files, capture_output=True, cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    assert [int(x) for x in p.stdout.split()] == want


def test_require_generated_kernel(tmp_path, monkeypatch):
    """MFA_REQUIRE_JIT=1: a caller that counts on the specialised kernel gets an error, not silently the other engine; MFA_VERBOSE=1
    says on stderr which kernel walks."""
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    env = dict(os.environ, MFA_REQUIRE_JIT="1", MFA_WALK="table")
    p = subprocess.run([diploma, "-match"], input=b"({a*}:1&1)*\naa\nexit\n", capture_output=True, cwd=tmp_path, env=env)
    assert p.returncode != 0 and b"specialised kernel" in p.stderr
    env = dict(os.environ, MFA_VERBOSE="1", MFA_WALK="table")
    p = subprocess.run([diploma, "-match"], input=b"({a*}:1&1)*\naa\nexit\n", capture_output=True, cwd=tmp_path, env=env)
    assert p.returncode == 0 and b"walk_kernel" in p.stderr and p.stdout.endswith(b"1\n")


def test_cli_example_runner(tmp_path):
    """`./diploma -match N` (main.cpp:11-13, matchers/example_runner.cpp:84-151): "len seconds" lines in
    test/example_N/diploma_results.txt, lengths following the reference's schedule (pump size 500, doubling per
    round and once more every tenth round, prefix accumulating the previous string)."""
    import subprocess
    from mfa_amd import corpus
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    for ex in (1, 5):
        regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
        d = tmp_path / "test" / ("example_%d" % ex)
        d.mkdir(parents=True)
        (d / "regexp.txt").write_text(regex + "\nunused-python-regex\n")
        (d / "pump.txt").write_text(",".join(pump) + "\n" + suffix + "\n" + prefix)
        p = subprocess.run([diploma, "-match", str(ex)], capture_output=True, text=True, cwd=tmp_path, timeout=600)
        assert p.returncode == 0, p.stderr
        assert p.stdout.splitlines()[0] == regex
        lines = (d / "diploma_results.txt").read_text().split("\n")
        assert lines[-1] == "" and len(lines) > 10
        for other in ("diploma_bnf_results.txt", "diploma_reverse_results.txt"):       # the -bnf and -reverse curves (example_runner.cpp:109-111)
            more = (d / other).read_text().split("\n")
            assert more[-1] == "" and [ln.split()[0] for ln in more[:-1]] == [ln.split()[0] for ln in lines[:len(more) - 1]] and len(more) > 10
        want, grown, size, rnd = [], prefix, 500, 0
        while True:
            grown = grown + corpus.pumped_string(size, pump) + suffix
            size += size
            if len(grown) > 0x00ffffff:
                break
            want.append(len(grown))
            rnd += 1
            if rnd % 10 == 0:
                size *= 2
        got = [ln.split() for ln in lines[:-1]]
        assert [int(g[0]) for g in got] == want[:len(got)], ex
        assert all(0.0 <= float(g[1]) < 1.0 for g in got)
        assert len(got) == len(want) or float(got[-1][1]) >= 0.5       # the series ends at the device limit or at the first slow match


def _structured_strings(ex, rng, count, max_len):
    """Attack-like inputs with long runs: pumped strings of random size with a few bytes flipped, and
    concatenations of runs -- what run acceleration jumps over."""
    from mfa_amd import corpus
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    out = []
    for k in range(count):
        kind = k % 4
        n = int(rng.integers(30, max_len))
        if kind < 2:
            s = bytearray((prefix + corpus.pumped_string(n, pump) + (suffix if k % 3 else "")).encode())
            for _ in range(int(rng.integers(0, 4)) if kind == 1 else 0):
                s[int(rng.integers(0, len(s)))] = int(rng.choice(list(b"abc")))
        elif kind == 2:
            s = bytearray()
            while len(s) < n:
                s += bytes([int(rng.choice(list(b"aab")))]) * int(rng.integers(1, max(2, n // 3)))
        else:
            s = bytearray(b"a" * n)
            if rng.random() < 0.5:
                s += bytes(rng.choice(list(b"abc"), size=int(rng.integers(1, 4))).tolist())
        out.append(bytes(s))
    return out


@pytest.mark.parametrize("mode", ["plain", "bnf", "reverse"])
@pytest.mark.parametrize("ex", range(1, 11))
def test_long_runs_against_oracle(ex, mode):
    """Strings far longer than the golden vectors (up to 6 KiB, with long runs of equal bytes) against the
    CPU restatement: covers run acceleration in the specialised kernels and the run cache in both."""
    name = "ex%d_%s" % (ex, mode)
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    rng = np.random.default_rng(1000 * ex + len(mode))
    strings = _structured_strings(ex, rng, 96, 6000)
    want = oracle_lib.OracleImage(blob).match(strings)
    img = capi.Image(blob)
    got = gpu_match(img, strings)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "%s: %d mismatches, first len %d %r... want %d" % (
        name, bad.size, len(strings[bad[0]]), strings[bad[0]][:60], want[bad[0]])


NFA_NAMES = [a["name"] for a in MANIFEST["automata"] if a["name"].startswith("nfa_")]


@pytest.mark.parametrize("name", NFA_NAMES)
def test_table_walk_whole_lines(name):
    """Memory-less automata on batches the golden strings are too short for: 256 strings of exactly 1 KiB (every
    128-byte line of the tiled walk lies wholly inside its string: the mask-free path, forward and reversed) and a
    ragged batch whose lines are partly outside their strings, against the CPU restatement."""
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    rng = np.random.default_rng(len(name) * 7919)
    ora = oracle_lib.OracleImage(blob)
    img = capi.Image(blob)
    for lens in ([1024] * 256, [int(x) for x in rng.integers(0, 700, size=300)]):
        strings = []
        for k, ln in enumerate(lens):
            t = bytes(rng.choice(list(b"ab" if k % 4 else b"abc."), size=ln).tolist())
            if k % 3 == 0 and ln >= 3:
                t = t[:-3] + b"abb"
            strings.append(t)
        want = ora.match(strings)
        got = gpu_match(img, strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s: %d mismatches, first len %d want %d" % (name, bad.size, len(strings[bad[0]]), want[bad[0]])
        assert img.info()["last_kernel"] == capi.KERNEL_TABLE


def _periodic_fuzz(rng, count, max_len, alphabet=b"abc"):
    """Concatenations of periodic regions (period 1..8) with occasional single-byte damage: the inputs run /
    period acceleration keys on, with region boundaries and damage at random phases."""
    out = []
    for _ in range(count):
        s = bytearray()
        target = int(rng.integers(40, max_len))
        while len(s) < target:
            q = int(rng.integers(1, 9))
            word = bytes(rng.choice(list(alphabet), size=q, p=None).tolist())
            reps = int(rng.integers(1, max(2, (target // q) // int(rng.integers(1, 4)) + 1)))
            s += word * reps
            if rng.random() < 0.3:
                s += bytes([int(rng.choice(list(alphabet)))])
        for _ in range(int(rng.integers(0, 3))):
            s[int(rng.integers(0, len(s)))] = int(rng.choice(list(alphabet)))
        out.append(bytes(s[:max_len]))
    return out


MFA_AUTOS = [a for a in MANIFEST["automata"] if not a["name"].startswith("nfa_")]


@pytest.mark.parametrize("auto", MFA_AUTOS, ids=lambda a: a["name"])
def test_periodic_fuzz_against_oracle(auto, monkeypatch):
    """Every memory automaton of the fixture set (plain, -bnf, -reverse images and the extra regexes, up to
    three cells) on periodic inputs, against the CPU restatement: the library's default engine and the table-driven walk (whose probe
    control decides where jumps are tried: shape history, optimistic and chained dual periods).  MFA_FUZZ_SEEDS=n: n more seeds, longer strings
    (MFA_FUZZ_FIRST=k: seeds k .. k + n)."""
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    alphabet = b"abcd" if "d" in auto["regex"] else b"abc"
    ora = oracle_lib.OracleImage(blob)
    first_seed = int(os.environ.get("MFA_FUZZ_FIRST", "0"))
    for seed in range(first_seed, first_seed + 1 + int(os.environ.get("MFA_FUZZ_SEEDS", "0"))):
        rng = np.random.default_rng((int.from_bytes(auto["name"].encode(), "little") + 7919 * seed) % (2 ** 32))
        strings = _periodic_fuzz(rng, 128, 2500 if seed == 0 else 12000, alphabet)
        want = ora.match(strings)
        for engine in ("", "table"):
            if engine:
                monkeypatch.setenv("MFA_WALK", engine)
            else:
                monkeypatch.delenv("MFA_WALK", raising=False)
            got = gpu_match(capi.Image(blob), strings)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, "%s seed %d engine %r: %d mismatches, first len %d %r want %d" % (
                auto["name"], seed, engine, bad.size, len(strings[bad[0]]), strings[bad[0]][:80], want[bad[0]])


@pytest.mark.parametrize("ex", range(1, 11))
def test_full_length_attack_strings(ex):
    """BASELINE-size inputs (pump size 64 KiB, 40 000, 20 000 with a damaged byte) against the CPU
    restatement: the strings the benchmark is made of, at the lengths where run acceleration does most work."""
    from mfa_amd import corpus
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    rng = np.random.default_rng(77 + ex)
    a = (prefix + corpus.pumped_string(65536, pump) + suffix).encode()
    b = (prefix + corpus.pumped_string(40000, pump)).encode()
    c = bytearray((prefix + corpus.pumped_string(20000, pump) + suffix).encode())
    c[int(rng.integers(100, len(c) - 100))] = ord("b") if c[5000] != ord("b") else ord("a")
    strings = [a, b, bytes(c)]
    blob = image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex))
    want = oracle_lib.OracleImage(blob).match(strings)
    got = gpu_match(capi.Image(blob), strings)
    assert list(got) == list(want)


def test_edge_cases():
    """Empty batch, empty strings, ragged lengths, a string at and above the length limit."""
    import torch
    blob = image.blob_from_dump(oracle_lib.load_dump("ex1_plain"))
    img = capi.Image(blob)
    # n = 0: nothing to do, no launch
    off0 = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert img.match_tensors(torch.zeros(64, dtype=torch.uint8, device="cuda"), off0).numel() == 0
    # empty strings between non-empty ones (the empty string matches ({a*}:1&1)*), ragged lengths 0..70
    strings = [b"", b"a", b"", b"aa", b"b", b""] + [b"a" * k for k in range(71)] + [b"a" * k + b"b" for k in range(40)]
    want = oracle_lib.OracleImage(blob).match(strings)
    assert list(gpu_match(img, strings)) == list(want)
    assert want[0] == 1 and want[4] == 0
    # a memory-less automaton on the same ragged batch
    nfa_blob = image.blob_from_dump(oracle_lib.load_dump("nfa_abb_thompson"))
    s2 = [b"", b"abb", b"a" * 17 + b"abb", b"b" * 129 + b"abb", b"abb" + b"a" * 200] + [b"ab" * k + b"abb" for k in range(60)]
    assert list(gpu_match(capi.Image(nfa_blob), s2)) == list(oracle_lib.OracleImage(nfa_blob).match(s2))
    # exactly at the limit is matched, one byte more is flagged 2 by the device entry point and refused by the host one
    limit = 0x00ffffff
    big = torch.full((limit + 1 + 64,), ord("a"), dtype=torch.uint8, device="cuda")
    off = torch.tensor([0, limit], dtype=torch.int64, device="cuda")
    r = img.match_tensors(big, off)
    torch.cuda.synchronize()
    assert int(r[0]) == 1                                   # a^n matches
    off = torch.tensor([0, limit + 1], dtype=torch.int64, device="cuda")
    r = img.match_tensors(big, off)
    torch.cuda.synchronize()
    assert int(r[0]) == 2
    data = np.full(limit + 1, ord("a"), dtype=np.uint8)
    with pytest.raises(capi.MfaError) as e:
        img.match_host(data, np.array([0, limit + 1], dtype=np.uint64))
    assert e.value.code == capi.ERR_TOO_LONG


def test_streams_and_reuse():
    """One image used from two streams with different batches, and re-used after that: results stay per batch."""
    import torch
    blob = image.blob_from_dump(oracle_lib.load_dump("ex6_plain"))
    img1, img2 = capi.Image(blob), capi.Image(blob)
    sa = oracle_lib.load_set("pump6")
    sb = oracle_lib.load_set("rnd")
    da, oa = oracle_lib.pack(sa)
    db, ob = oracle_lib.pack(sb)
    ta = torch.zeros(len(da) + 64, dtype=torch.uint8, device="cuda"); ta[:len(da)] = torch.from_numpy(da.copy())
    tb = torch.zeros(len(db) + 64, dtype=torch.uint8, device="cuda"); tb[:len(db)] = torch.from_numpy(db.copy())
    oa_t, ob_t = torch.from_numpy(oa.astype(np.int64)).cuda(), torch.from_numpy(ob.astype(np.int64)).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ra = img1.match_tensors(ta, oa_t, stream=s1)
    rb = img2.match_tensors(tb, ob_t, stream=s2)
    torch.cuda.synchronize()
    assert np.array_equal(ra.cpu().numpy(), oracle_lib.load_bits("ex6_plain", "pump6"))
    assert np.array_equal(rb.cpu().numpy(), oracle_lib.load_bits("ex6_plain", "rnd"))
    rb2 = img1.match_tensors(tb, ob_t)
    torch.cuda.synchronize()
    assert np.array_equal(rb2.cpu().numpy(), oracle_lib.load_bits("ex6_plain", "rnd"))


def _front_end_blob(regex, tmp_path, flag="-mfa"):
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    p = subprocess.run([diploma, "-dump", flag], input=regex + "\n", capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    return image.blob_from_dump(p.stdout)


@pytest.mark.parametrize("seed", range(3))
def test_random_regexes_on_gpu(seed, tmp_path, monkeypatch):
    """Automata nobody has seen before: random regexes of the README grammar through the host front-end, matched by
    the table-driven walk (all of them) and by a freshly generated specialised kernel (the first one per seed),
    against the CPU restatement."""
    import random
    from test_frontend_fuzz import rand_regex
    rng = random.Random(9000 + seed)
    done = 0
    for _ in range(40):
        if done >= 6:
            break
        regex = rand_regex(rng, rng.randint(2, 3), [], True)
        if "{" not in regex and "&" not in regex:
            continue
        try:
            blob = _front_end_blob(regex, tmp_path)
        except (AssertionError, image.ImageError):
            continue
        strings = [("".join(rng.choice("aaabbc") for _ in range(rng.randint(0, 60)))).encode() for _k in range(200)]
        strings += [(("".join(rng.choice("ab") for _ in range(rng.randint(1, 3)))) * rng.randint(1, 400) + rng.choice(["", "a", "c"])).encode()
                    for _k in range(60)]
        want = oracle_lib.OracleImage(blob).match(strings)
        for mode in (["table", "specialised"] if done == 0 else ["table"]):
            use_engine(monkeypatch, mode)
            try:
                img = capi.Image(blob)
            except capi.MfaError as e:
                assert e.code == capi.ERR_UNSUPPORTED      # outside the kernels' structural limits: refused, not mis-matched
                break
            got = gpu_match(img, strings)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, "regex %r (%s): %d mismatches, first %r want %d" % (regex, mode, bad.size, strings[bad[0]], want[bad[0]])
        done += 1
    assert done >= 3


@pytest.mark.parametrize("ex", range(1, 11))
def test_acceleration_changes_nothing_on_the_bench_corpus(ex, monkeypatch):
    """The benchmark's own strings (pump sizes up to 64 KiB): the accelerated walk, the same kernel executing every
    step (MFA_ACCEL=0) and -- on a prefix of the batch -- the table-driven kernel must give identical answers.  Size
    independent property: three different ways through the same semantics."""
    import torch
    from mfa_amd import corpus
    n = 12000
    sizes = corpus.pump_sizes(n, 0x5EED0004 + ex, 1024, 65536)
    ws = (np.arange(n) % 2) == 0
    d_bytes, d_off = corpus.device_batch(ex, sizes, ws, torch.device("cuda", 0))
    blob = image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex))
    monkeypatch.setenv("MFA_WALK", "jit")
    img = capi.Image(blob)
    monkeypatch.setenv("MFA_ACCEL", "1")
    fast = img.match_tensors(d_bytes, d_off).clone()
    torch.cuda.synchronize()
    monkeypatch.setenv("MFA_ACCEL", "0")
    slow = img.match_tensors(d_bytes, d_off).clone()
    torch.cuda.synchronize()
    assert torch.equal(fast, slow)
    # the other engine, accelerated and executing every step (a prefix of the batch)
    monkeypatch.setenv("MFA_ACCEL", "1")
    monkeypatch.setenv("MFA_WALK", "table")
    other = capi.Image(blob).match_tensors(d_bytes, d_off).clone()
    torch.cuda.synchronize()
    assert torch.equal(other, fast)
    monkeypatch.setenv("MFA_ACCEL", "0")
    m = 600
    plain = capi.Image(blob).match_tensors(d_bytes, d_off[:m + 1].clone())
    torch.cuda.synchronize()
    assert torch.equal(plain, fast[:m])
    # and the short ones among them against the CPU restatement
    short = [k for k in range(n) if sizes[k] <= 3000][:40]
    strings = corpus.host_strings(ex, sizes[short], ws[short])
    want = oracle_lib.OracleImage(blob).match(strings)
    assert list(fast[short].cpu().numpy()) == list(want)


MANY_CELLS = ["{a*}:1{b*}:2{c*}:3{a*}:4{b*}:5&5&4&3&2&1", "{a}:1{b}:2{c}:3{a*}:4{b*}:5{c*}:6{a}:7{b}:8{c*}:9&9&8&7&6&5&4&3&2&1"]


@pytest.mark.parametrize("regex", MANY_CELLS, ids=["5cells", "9cells"])
def test_more_than_four_cells(regex, tmp_path, monkeypatch):
    """cells "1".."9" (mfa.cpp:148): automata with five and nine cells on both kernels against the CPU restatement"""
    import random
    blob = _front_end_blob(regex, tmp_path)
    assert image.blob_info(blob)["n_cells"] == (5 if regex.startswith("{a*}") else 9)
    rng = random.Random(len(regex))
    strings = [b"", b"abcab" + b"bacba"[::-1], b"aabbccaabb" + b"bbaaccbbaa"]
    for _ in range(300):
        parts = ["a" * rng.randint(0, 3), "b" * rng.randint(0, 3), "c" * rng.randint(0, 3), "a" * rng.randint(0, 3), "b" * rng.randint(0, 3)]
        s = "".join(parts) + "".join(reversed(parts)) if rng.random() < 0.5 else "".join(rng.choice("abc") for _ in range(rng.randint(0, 24)))
        strings.append(s.encode())
    strings += [b"abcaabbccabc" + b"cbccbbaacba"[::-1], b"abc" + b"a" * 40 + b"b" * 30 + b"c" * 20 + b"ab" + b"c" * 25 + b"c" * 25 + b"ba" + b"c" * 20 + b"b" * 30 + b"a" * 40 + b"cba",
                b"abc" + b"ab" + b"ba" + b"cba", b"abcaabbccabcccbaccbbaacba"]
    want = oracle_lib.OracleImage(blob).match(strings)
    for mode in ("table", "specialised"):
        use_engine(monkeypatch, mode)
        got = gpu_match(capi.Image(blob), strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s: %d mismatches, first %r want %d" % (mode, bad.size, strings[bad[0]], want[bad[0]])
    assert want.sum() > 0


# nondeterministic automata with seven cells: several ways to the same node and position with different cell contents, and reads
# of cells that were never set (mfa.cpp:148-160 recurses once per such cell)
MANY_CELLS_TIES = ["({a}:1|b)({a}:2|b)({a}:3|b)({a}:4|b)({a}:5|b)({a}:6|b)({a}:7|b)(&1|&2)(&3|&4)(&5|&6)&7",
                   "{a*}:1{a*}:2{a*}:3{a*}:4{a*}:5{a*}:6{a*}:7b&7&6&5&4&3&2&1"]


@pytest.mark.parametrize("regex", MANY_CELLS_TIES, ids=["unset7", "split7"])
def test_many_cells_tie_breaks(regex, tmp_path, monkeypatch):
    import random
    blob = _front_end_blob(regex, tmp_path)
    assert image.blob_info(blob)["n_cells"] == 7
    rng = random.Random(7 + len(regex))
    strings = [b"", b"bbbbbbb", b"bbbbbba" + b"a", b"aaaaaaa" + b"aaaa", b"abababa" + b"aaaa", b"ab", b"aab" + b"aa", b"aaab" + b"aaa"]
    for _ in range(500):
        if regex.startswith("("):
            head = "".join(rng.choice("ab") for _ in range(7))
            tail = "".join(rng.choice(["a", ""]) for _ in range(4))
            strings.append((head + tail).encode())
        else:
            n = rng.randint(0, 9)
            strings.append(("a" * n + "b" + "a" * (n if rng.random() < 0.7 else rng.randint(0, 9))).encode())
    want = oracle_lib.OracleImage(blob).match(strings)
    for mode in ("table", "specialised"):
        use_engine(monkeypatch, mode)
        got = gpu_match(capi.Image(blob), strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s: %d mismatches, first %r want %d" % (mode, bad.size, strings[bad[0]], want[bad[0]])
    assert 0 < want.sum() < len(strings)


def test_large_tabulated_automaton(tmp_path):
    """a memory-less automaton whose subset construction has hundreds of state sets (the table does not fit LDS): Thompson image
    of (a|b)*a(a|b)^8, forward walk, against the CPU restatement"""
    regex = "(a|b)*a" + "(a|b)" * 8
    blob = _front_end_blob(regex, tmp_path, "-thompson")
    img = capi.Image(blob)
    assert img.info()["dfa_states"] > 127
    rng = np.random.default_rng(99)
    strings = [bytes(rng.choice(list(b"ab"), size=int(n)).tolist()) for n in rng.integers(0, 300, size=600)] + [b"", b"a" + b"b" * 8, b"b" * 9]
    want = oracle_lib.OracleImage(blob).match(strings)
    got = gpu_match(img, strings)
    assert np.array_equal(got, want) and 0 < want.sum() < len(strings)


@pytest.mark.parametrize("k,flag", [(14, "-thompson"), (16, "-thompson"), (16, "-glushkov")])
def test_exponential_tabulated_automaton(k, flag, tmp_path):
    """(a|b)*a(a|b)^k: determinisation doubles with every k -- 32 770 state sets at k = 14 (16-bit table in L2), 131 074 at k = 16
    (32-bit table) -- walked on the GPU against the CPU restatement, which walks the NFA itself (automata.cpp:177-210)."""
    regex = "(a|b)*a" + "(a|b)" * k
    blob = _front_end_blob(regex, tmp_path, flag)
    img = capi.Image(blob)
    assert img.info()["dfa_states"] > 4096
    rng = np.random.default_rng(k)
    strings = [bytes(rng.choice(list(b"ab"), size=int(n)).tolist()) for n in rng.integers(0, 400, size=500)]
    strings += [b"", b"a" + b"b" * k, b"b" * (k + 1), b"ab" * 300 + b"a" + b"b" * k, b"ab" * 300 + b"b" + b"a" * k, b"abc" + b"a" * 30]
    want = oracle_lib.OracleImage(blob).match(strings)
    got = gpu_match(img, strings)
    assert np.array_equal(got, want) and 0 < want.sum() < len(strings)


@pytest.mark.parametrize("ex", [3, 6, 8])
def test_full_length_reversed_strings(ex):
    """BASELINE configs[4] at its real size: the reversed automata (`-reverse`, is_reversed = 1) of the nondeterministic examples on
    pump-only strings (full walk) and pump + suffix strings (early exit) of pump size 64 KiB, 40 000 and 20 000 with a damaged
    byte, against the CPU restatement."""
    from mfa_amd import corpus
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    rng = np.random.default_rng(170 + ex)
    a = (prefix + corpus.pumped_string(65536, pump)).encode()
    b = (prefix + corpus.pumped_string(65536, pump) + suffix).encode()
    c = (prefix + corpus.pumped_string(40000, pump)).encode()
    d = bytearray((prefix + corpus.pumped_string(20000, pump) + suffix).encode())
    d[int(rng.integers(100, len(d) - 100))] = ord("b") if d[5000] != ord("b") else ord("a")
    e = bytearray(a[:30000])
    e[int(rng.integers(1000, 29000))] = ord("c")
    strings = [a, b, c, bytes(d), bytes(e)]
    blob = image.blob_from_dump(oracle_lib.load_dump("ex%d_reverse" % ex))
    assert image.blob_info(blob)["reversed"] == 1
    want = oracle_lib.OracleImage(blob).match(strings)
    got = gpu_match(capi.Image(blob), strings)
    assert list(got) == list(want)


# ---- the table-driven walk: what only it can do ------------------------------------------------------------------------------------
def test_mixed_batch_in_one_call(monkeypatch):
    """mfa_match_mixed: ONE batch, ten segments, ten automata (the 10-example attack corpus at a small size), against the oracle
    and against matching every segment by itself; both engines behind the call."""
    import torch
    from mfa_amd import corpus
    layout = [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    n_per = 3000
    dev = torch.device("cuda", 0)
    parts_b, parts_o, seg, pos_b, blobs, samples = [], [], [0], 0, [], []
    for ex in layout:
        sizes = corpus.pump_sizes(n_per, 0x5EED0004 + ex, 64, 20000)
        ws = (np.arange(n_per) % 2) == 0
        b, o = corpus.device_batch(ex, sizes, ws, dev)
        nb = int(o[-1].item())
        parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
        blobs.append(image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)))
        short = [k for k in range(n_per) if sizes[k] <= 2000][:60]
        samples.append((short, corpus.host_strings(ex, sizes[short], ws[short])))
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
    results = {}
    for engine in ("table", "specialised"):
        use_engine(monkeypatch, engine)
        images = [capi.Image(b) for b in blobs]
        mx = capi.Mixed(images)
        for cuts in ("0.3,0.6,0.8,0.9", "0.17,0.55", ""):          # cuts inside segments, and one group
            monkeypatch.setenv("MFA_MIXED_CUTS", cuts)
            res = mx.match_tensors(bytes_all, off_all, seg).clone()
            torch.cuda.synchronize()
            results[(engine, cuts)] = res
        for k in range(len(layout)):                                # segment by segment through the single-automaton entry point
            alone = images[k].match_tensors(bytes_all, off_all[seg[k]:seg[k + 1] + 1])
            torch.cuda.synchronize()
            assert torch.equal(alone, results[(engine, "")][seg[k]:seg[k + 1]]), (engine, layout[k])
        mx.close()
    first = next(iter(results.values()))
    for key, r in results.items():
        assert torch.equal(r, first), key
    for k, (short, strings) in enumerate(samples):
        want = oracle_lib.OracleImage(blobs[k]).match(strings)
        got = first[seg[k]:seg[k + 1]][short].cpu().numpy()
        assert np.array_equal(got, want), layout[k]


def test_mixed_call_with_more_launches_than_slots_is_refused_before_it_starts(monkeypatch):
    """a mixed object's launch slots (24) can run out -- 26 segments whose automata alternate between one and two cells
    are 26 launches.  The call plans before it queues anything: it returns MFA_ERR_UNSUPPORTED, the result buffer is untouched, and the
    same object then matches a batch that needs few launches (the other segments empty), against the oracle."""
    import torch
    from mfa_amd import corpus
    monkeypatch.setenv("MFA_WALK", "table")
    monkeypatch.setenv("MFA_MIXED_CUTS", "")
    dev = torch.device("cuda", 0)
    layout = [1, 3] * 13
    n_per = 200
    blobs = {ex: image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)) for ex in (1, 3)}
    images = [capi.Image(blobs[ex]) for ex in layout]
    mx = capi.Mixed(images)
    parts_b, parts_o, seg, pos_b, hosts = [], [], [0], 0, []
    for ex in layout:
        sizes = corpus.pump_sizes(n_per, 0x5EED0051 + ex + len(seg), 64, 1500)
        ws = (np.arange(n_per) % 2) == 0
        b, o = corpus.device_batch(ex, sizes, ws, dev)
        nb = int(o[-1].item())
        parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
        hosts.append(corpus.host_strings(ex, sizes, ws))
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
    out = torch.full((seg[-1],), 7, dtype=torch.uint8, device=dev)
    with pytest.raises(capi.MfaError) as err:
        mx.match_tensors(bytes_all, off_all, seg, d_results=out)
    assert err.value.code == capi.ERR_UNSUPPORTED
    torch.cuda.synchronize()
    assert int((out != 7).sum().item()) == 0                      # nothing was started
    # the first four segments only: the 22 others are empty (seg_first repeats), four launches
    n4 = seg[4]
    seg4 = seg[:5] + [n4] * (len(seg) - 5)
    res = mx.match_tensors(bytes_all, off_all[:n4 + 1], seg4).clone()
    torch.cuda.synchronize()
    for k in range(4):
        want = oracle_lib.OracleImage(blobs[layout[k]]).match(hosts[k])
        assert np.array_equal(res[seg[k]:seg[k + 1]].cpu().numpy(), want), k
    mx.close()


def test_spill_budget_shrinks_the_grid_or_refuses(monkeypatch):
    """the walk's spill areas are sized per wave of the grid for the worst case and bounded by a byte budget
    (MFA_WALK_SPILL_MB, 2 GiB by default): a small budget means fewer (persistent) waves and the same answers; a budget that not even one
    workgroup fits is MFA_ERR_NOMEM, not a failed allocation somewhere inside.  The 77-node automaton: lists of 10 entries, most of them spilled."""
    import torch
    from mfa_amd import corpus
    monkeypatch.setenv("MFA_WALK", "table")
    dev = torch.device("cuda", 0)
    blob = image.blob_from_dump(oracle_lib.load_dump("ex8_reverse"))
    n = 1500
    sizes = corpus.pump_sizes(n, 0x5EED0061, 64, 1200)
    ws = (np.arange(n) % 2) == 0
    b, o = corpus.device_batch(8, sizes, ws, dev)
    want = oracle_lib.OracleImage(blob).match(corpus.host_strings(8, sizes, ws))
    for mb in ("2048", "8"):
        monkeypatch.setenv("MFA_WALK_SPILL_MB", mb)
        img = capi.Image(blob)
        got = img.match_tensors(b, o).cpu().numpy()
        assert np.array_equal(got, want), mb
        img.close()
    monkeypatch.setenv("MFA_WALK_SPILL_MB", "1")
    img = capi.Image(blob)
    with pytest.raises(capi.MfaError) as err:
        img.match_tensors(b, o)
    assert err.value.code == capi.ERR_NOMEM
    img.close()


def test_mixed_calls_on_two_streams_share_one_object(monkeypatch):
    """One mfa_mixed object, calls alternating between two caller streams with nothing but stream order between them: the object's table,
    counters and spill areas are shared, so a call must start behind the end of the one before it whichever stream that came on (the region
    launches run on the CALLER's stream).  Two batches of different content and size; every call's answers against the single-automaton entry
    point's."""
    import torch
    from mfa_amd import corpus
    monkeypatch.setenv("MFA_WALK", "table")
    monkeypatch.delenv("MFA_MIXED_CUTS", raising=False)
    dev = torch.device("cuda", 0)
    blobs = [image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)) for ex in (1, 6, 8)]
    images = [capi.Image(b) for b in blobs]
    batches = []
    for seed, n_per, hi in ((0x5EED0081, 30000, 6000), (0x5EED0082, 23000, 9000)):
        parts_b, parts_o, seg, pos_b = [], [], [0], 0
        for ex in (1, 6, 8):
            sizes = corpus.pump_sizes(n_per, seed + ex, 64, hi)
            b, o = corpus.device_batch(ex, sizes, (np.arange(n_per) % 2) == 0, dev)
            nb = int(o[-1].item())
            parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
        bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
        off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
        want = torch.cat([images[k].match_tensors(bytes_all, off_all[seg[k]:seg[k + 1] + 1]).clone() for k in range(3)])
        batches.append((bytes_all, off_all, seg, want))
    torch.cuda.synchronize()
    mx = capi.Mixed(images)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = []
    for call in range(8):
        bytes_all, off_all, seg, want = batches[call % 2]
        out = torch.full((seg[-1],), 7, dtype=torch.uint8, device=dev)
        mx.match_tensors(bytes_all, off_all, seg, d_results=out, stream=streams[(call // 2 + call) % 2])
        outs.append((out, want))
    torch.cuda.synchronize()
    for call, (out, want) in enumerate(outs):
        assert torch.equal(out, want), call
    mx.close()


def test_mixed_batch_default_grouping_reads_the_size_once(monkeypatch):
    """Without MFA_MIXED_CUTS the call chooses its groups from the batch's bytes, which it reads back on the first call with a string
    count it has not met (waiting for the caller's stream) and remembers: first and second call, on a side stream with work pending, must
    agree with the single-automaton entry point."""
    import torch
    from mfa_amd import corpus
    monkeypatch.delenv("MFA_MIXED_CUTS", raising=False)
    dev = torch.device("cuda", 0)
    n_per = 40000                                             # 80 000 strings: above the 65 536 below which a batch is never cut
    parts_b, parts_o, seg, pos_b, blobs = [], [], [0], 0, []
    for ex in (1, 6):
        sizes = corpus.pump_sizes(n_per, 0x5EED0031 + ex, 64, 3000)
        b, o = corpus.device_batch(ex, sizes, (np.arange(n_per) % 2) == 0, dev)
        nb = int(o[-1].item())
        parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
        blobs.append(image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)))
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
    images = [capi.Image(b) for b in blobs]
    want = torch.cat([images[k].match_tensors(bytes_all, off_all[seg[k]:seg[k + 1] + 1]).clone() for k in range(2)])
    torch.cuda.synchronize()
    mx = capi.Mixed(images)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        busy = torch.zeros(1 << 26, dtype=torch.float32, device=dev)
        for _ in range(20):
            busy += 1.0                                        # work the first call has to wait for
        first = mx.match_tensors(bytes_all, off_all, seg, stream=side).clone()
        second = mx.match_tensors(bytes_all, off_all, seg, stream=side).clone()
    side.synchronize()
    assert torch.equal(first, want) and torch.equal(second, want)
    mx.close()


def test_mixed_batch_corner_cases(tmp_path, monkeypatch):
    """mfa_match_mixed beyond the headline shape: reversed automata in one launch (a launch scans in one direction), an automaton
    with 7 cells among automata with 1-2 (every table of the launch is then read with 3-word edges), an empty segment, and a mix
    of scan directions, which the table engine refuses and the per-segment engine matches."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)

    def batch_of(lists):
        strings = [s for l in lists for s in l]
        seg = [0]
        for l in lists:
            seg.append(seg[-1] + len(l))
        data, off = oracle_lib.pack(strings if strings else [b""])
        d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
        d_bytes[:len(data)] = torch.from_numpy(data.copy())
        return d_bytes, torch.from_numpy(off.astype(np.int64)).to(dev), seg, strings

    def texts(n):
        out = []
        for _ in range(n):
            u = bytes(rng.choice(list(b"ab"), size=int(rng.integers(1, 6))).tolist())
            out.append(u * int(rng.integers(1, 500)) + bytes(rng.choice(list(b"abc"), size=int(rng.integers(0, 3))).tolist()))
        return out

    cases = {
        "reversed": [image.blob_from_dump(oracle_lib.load_dump(n)) for n in ("ex3_reverse", "ex6_reverse", "ex8_reverse", "ex2_reverse")],
        "wide": [image.blob_from_dump(oracle_lib.load_dump("ex1_plain")), _front_end_blob(MANY_CELLS_TIES[1], tmp_path), image.blob_from_dump(oracle_lib.load_dump("ex5_plain"))],
        "mixed directions": [image.blob_from_dump(oracle_lib.load_dump("ex1_plain")), image.blob_from_dump(oracle_lib.load_dump("ex3_reverse"))],
    }
    for name, blobs in cases.items():
        lists = [texts(int(rng.integers(150, 400))) for _ in blobs]
        if name == "reversed":
            lists[1] = []                                           # an empty segment
        d_bytes, d_off, seg, strings = batch_of(lists)
        want = np.concatenate([oracle_lib.OracleImage(b).match(l) if l else np.zeros(0, dtype=np.uint8) for b, l in zip(blobs, lists)])
        for engine in ("table", "specialised"):
            use_engine(monkeypatch, engine)
            images = [capi.Image(b) for b in blobs]
            mx = capi.Mixed(images)
            got = mx.match_tensors(d_bytes, d_off, seg)
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), want), (name, engine)
            mx.close()


def test_more_than_128_nodes(tmp_path, monkeypatch):
    """beyond the generated kernels' 128 nodes: a memory automaton with several hundred nodes (a long literal chain around a
    back-reference) walked by the table-driven kernel, against the CPU restatement"""
    import random
    rng = random.Random(128)
    chain = "".join(rng.choice("ab") for _ in range(300))
    regex = "{a*}:1" + chain + "&1" + "(a|b)*"
    blob = _front_end_blob(regex, tmp_path)
    assert image.blob_info(blob)["n_nodes"] > 128
    strings = []
    for n in (0, 1, 5, 70, 400):
        strings += [("a" * n + chain + "a" * n).encode(), ("a" * n + chain + "a" * n + "abba").encode(), ("a" * n + chain + "a" * (n + 1) + "b").encode(),
                    ("a" * n + chain[:-1] + "a" * n).encode()]
    want = oracle_lib.OracleImage(blob).match(strings)
    img = capi.Image(blob)
    got = gpu_match(img, strings)
    assert img.info()["last_kernel"] == capi.KERNEL_WALK
    assert list(got) == list(want) and 0 < want.sum() < len(strings)


def _images_of(name_filter):
    return [a["name"] for a in MANIFEST["automata"] if name_filter(a["name"])]


@pytest.mark.parametrize("name", _images_of(lambda n: n.startswith("ex") and n.split("_")[1] in ("plain", "bnf", "reverse")))
def test_long_strings_every_image(name, monkeypatch):
    """Every image of the ten examples (plain, -bnf, -reverse) on at least 100 strings of 32-64 KiB: pumped with and without the
    suffix, with one to three damaged bytes at random offsets, and cut-and-spliced halves -- where cell reads jump the farthest
    (mfa.cpp:177-191).  Against the CPU restatement, on both engines."""
    from mfa_amd import corpus
    ex = int("".join(c for c in name.split("_")[0] if c.isdigit()))
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    rng = np.random.default_rng(1000 + ex)
    strings = []
    for k in range(26):
        n = int(rng.integers(32768, 65536))
        base = prefix + corpus.pumped_string(n, pump)
        for ws in (False, True):
            s = (base + (suffix if ws else "")).encode()
            strings.append(s)
            d = bytearray(s)
            for _ in range(int(rng.integers(1, 4))):
                d[int(rng.integers(0, len(d)))] = int(rng.choice(list(b"abc")))
            strings.append(bytes(d))
        h = len(base) // 2
        cut = int(rng.integers(1, h))
        strings.append((base[:h] + base[cut:]).encode())
    assert len(strings) >= 100 and min(len(s) for s in strings) >= 32768
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    want = oracle_lib.OracleImage(blob).match(strings)
    for engine in ("table", "specialised"):
        use_engine(monkeypatch, engine)
        got = gpu_match(capi.Image(blob), strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s %s: %d mismatches, first at string %d (len %d) want %d" % (name, engine, bad.size, bad[0], len(strings[bad[0]]), want[bad[0]])


def test_config3_noise_strings_both_engines_and_mixed(monkeypatch):
    """BASELINE configs[2]'s fourth kind of string: example 1 on 64 KiB strings of i.i.d. bytes
    {a: 0.99, b: 0.01}, seed 0x5EED0003 -- hundreds of medium runs per string, more periodic stretches than a region-table row holds --
    through both walk engines and through mfa_match_mixed, against the CPU restatement (mfa.cpp:177-191: every cell read compares a
    run of a's with the run at the scan position).  A few strings with 0, 1 and 2 foreign bytes are mixed in so that both answers occur."""
    import torch
    n, length = 288, 65536
    rng = np.random.Generator(np.random.Philox(0x5EED0003))
    rows = np.where(rng.random((n, length)) < 0.01, ord("b"), ord("a")).astype(np.uint8)
    rows[256:264] = ord("a")                                        # a^L: accepted
    rows[264:272] = ord("a"); rows[264:272, -1] = ord("b")          # a^(L-1) b
    rows[272:288] = ord("a")
    for k in range(272, 288):
        rows[k, rng.integers(0, length, size=2)] = ord("b")
    strings = [r.tobytes() for r in rows]
    blob = image.blob_from_dump(oracle_lib.load_dump("ex1_plain"))
    want = np.asarray(oracle_lib.OracleImage(blob).match(strings))
    assert want[256:264].all() and not want[:256].any()
    for engine in ("table", "specialised"):
        use_engine(monkeypatch, engine)
        img = capi.Image(blob)
        got = gpu_match(img, strings)
        assert np.array_equal(got, want), (engine, np.nonzero(got != want)[0][:5])
        data, off = oracle_lib.pack(strings)
        d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
        d_bytes[:len(data)] = torch.from_numpy(data.copy())
        d_off = torch.from_numpy(off.astype(np.int64)).cuda()
        mx = capi.Mixed([img])
        for cuts in ("", "0.25,0.5"):
            monkeypatch.setenv("MFA_MIXED_CUTS", cuts)
            res = mx.match_tensors(d_bytes, d_off, [0, n]).clone()
            torch.cuda.synchronize()
            assert np.array_equal(res.cpu().numpy(), want), (engine, "mixed", cuts)
        monkeypatch.delenv("MFA_MIXED_CUTS")
        tab = capi.region_scan(d_bytes, d_off)
        assert int(((tab[:256, 0] & capi.REGION_OVERFLOW) != 0).sum().item()) > 200      # the noise strings overflow their table rows
        mx.close(); img.close()


@pytest.mark.parametrize("name", ["ex1_plain", "ex6_plain", "ex9_plain", "ex8_reverse", "ex2_bnf"])
def test_lean_walk_of_text_without_stretches(name, monkeypatch):
    """The table engine hands strings without periodic stretches (an empty region-table row, 256 bytes and more) to walk_lean_kernel through
    a queue (walk.hip): a batch that mixes such text with attack strings and short strings, against the CPU restatement, with the lean
    kernel and with MFA_WALK_LEAN=0, through the single-automaton call and through mfa_match_mixed."""
    import random
    import torch
    from mfa_amd import corpus
    rng = random.Random(3 + len(name))
    strings = []
    for _ in range(300):
        strings.append(b"".join((b"a" * rng.randint(1, 20) + b"b") for _ in range(rng.randint(5, 400))))
        strings.append(bytes(rng.choice(b"ab") for _ in range(rng.randint(200, 3000))))
    ex = int("".join(c for c in name.split("_")[0] if c.isdigit()))
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    for n in (300, 2000, 9000):
        strings.append((prefix + corpus.pumped_string(n, pump) + suffix).encode())
        strings.append((prefix + corpus.pumped_string(n, pump)).encode())
    strings += [b"", b"a", b"ab" * 100, b"b" * 255]
    rng.shuffle(strings)
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    want = np.asarray(oracle_lib.OracleImage(blob).match(strings))
    monkeypatch.setenv("MFA_WALK", "table")
    for lean in ("1", "0"):
        monkeypatch.setenv("MFA_WALK_LEAN", lean)
        img = capi.Image(blob)
        got = gpu_match(img, strings)
        assert np.array_equal(got, want), (name, lean, np.nonzero(got != want)[0][:5])
        data, off = oracle_lib.pack(strings)
        d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
        d_bytes[:len(data)] = torch.from_numpy(data.copy())
        d_off = torch.from_numpy(off.astype(np.int64)).cuda()
        mx = capi.Mixed([img])
        res = mx.match_tensors(d_bytes, d_off, [0, len(strings)]).clone()
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy(), want), (name, lean, "mixed")
        mx.close(); img.close()
