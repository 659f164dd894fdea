"""Region pre-pass (re2-modification_amd/csrc/regions.hip) against a numpy restatement of its definition, and the
launch-context pool behind the C-ABI's re-entrancy promise (include/mfa_hip.h).

The table is an aid of the walk kernels, so what is checked is what they rely on: every entry is a true
periodic region, q = 1 entries are exactly the maximal runs, long runs are all there."""
import os
import threading

import numpy as np
import pytest

import oracle_lib
from mfa_amd import capi, corpus, image

pytestmark = pytest.mark.gpu


def true_regions(s, q):
    """maximal q-periodic regions [lo, hi) of the byte string s (numpy uint8): zero runs of s[j] != s[j+q]"""
    n = len(s)
    if n <= q:
        return []
    d = np.concatenate(([1], (s[:-q] != s[q:]).astype(np.int8), [1]))
    edges = np.flatnonzero(np.diff(d))
    return [(int(a), int(b) + q) for a, b in zip(edges[0::2], edges[1::2])]


def scan(strings, pad=0):
    """region tables of `strings`; `pad` shifts the whole batch so that strings start at every alignment"""
    import torch
    data, off = oracle_lib.pack(strings)
    d_bytes = torch.zeros(pad + len(data) + 64, dtype=torch.uint8, device="cuda")
    d_bytes[pad:pad + len(data)] = torch.from_numpy(data.copy())
    d_off = torch.from_numpy(off.astype(np.int64) + pad).cuda()
    tab = capi.region_scan(d_bytes, d_off)
    torch.cuda.synchronize()
    return tab.cpu().numpy().astype(np.uint64)


def decode(row):
    cnt = int(row[0]) & 0xff
    assert cnt <= capi.REGION_MAX
    out = []
    for w in row[1:1 + cnt]:
        w = int(w)
        out.append((w & 0xffffff, (w >> 24) & 0xffffff, (w >> 48) & 15))
    return out, bool(int(row[0]) & capi.REGION_OVERFLOW)


def check_tables(strings, tabs):
    n_entries = 0
    for s, row in zip(strings, tabs):
        a = np.frombuffer(s, dtype=np.uint8)
        entries, overflow = decode(row)
        n_entries += len(entries)
        for lo, hi, q in entries:
            assert 1 <= q <= 8 and lo < hi <= len(a) and hi - lo >= capi.REGION_MIN_LEN, (lo, hi, q, len(a))
            assert np.array_equal(a[lo:hi - q], a[lo + q:hi]), ("not periodic", lo, hi, q, s[:80])
            if q == 1:                                  # maximal at both ends
                assert lo == 0 or a[lo - 1] != a[lo], ("run not maximal at lo", lo, hi, s[:80])
                assert hi == len(a) or a[hi] != a[hi - 1], ("run not maximal at hi", lo, hi, s[:80])
        if overflow:
            continue
        have = set(entries)
        for lo, hi in true_regions(a, 1):
            if hi - lo >= 128:
                assert (lo, hi, 1) in have, ("long run missing", lo, hi, entries[:6], s[:80])
        # longer periods: what a divisor's region covers may be dropped or cut; the rest must be there, up to short edges
        for q in range(2, 9):
            for lo, hi in true_regions(a, q):
                if hi - lo < 512:
                    continue
                cov = np.zeros(hi - lo, dtype=bool)
                for l2, h2, q2 in entries:
                    if q % q2 == 0:
                        cov[max(l2, lo) - lo:max(min(h2, hi), lo) - lo] = True
                gaps = np.flatnonzero(np.diff(np.concatenate(([1], cov.astype(np.int8), [1]))))
                worst = max([int(b - a2) for a2, b in zip(gaps[0::2], gaps[1::2])], default=0)
                assert worst <= 160, ("long stretch of a region uncovered", q, lo, hi, worst, entries[:8], s[:80])
    return n_entries


def _fuzz_strings(rng, count, max_len):
    out = []
    for k in range(count):
        kind = k % 5
        target = int(rng.integers(0, max_len))
        s = bytearray()
        if kind == 0:                                   # concatenated periodic stretches
            while len(s) < target:
                q = int(rng.integers(1, 9))
                word = bytes(rng.choice(list(b"abc"), size=q).tolist())
                s += word * int(rng.integers(1, max(2, target // q // int(rng.integers(1, 5)) + 1)))
                if rng.random() < 0.4:
                    s += bytes([int(rng.choice(list(b"abcd")))])
        elif kind == 1:                                 # random text
            s += bytes(rng.choice(list(b"ab"), size=target).tolist())
        elif kind == 2:                                 # one byte repeated, a little damage
            s += b"a" * target
            for _ in range(int(rng.integers(0, 4))):
                if s:
                    s[int(rng.integers(0, len(s)))] = ord("b")
        elif kind == 3:                                 # runs of random lengths around the table's threshold
            while len(s) < target:
                s += bytes([int(rng.choice(list(b"ab")))]) * int(rng.integers(1, 200))
        else:                                           # attack strings of the corpus
            ex = int(rng.integers(1, 11))
            regex, pump, suffix, prefix = corpus.EXAMPLES[ex]
            s += (prefix + corpus.pumped_string(max(target, 8), pump) + (suffix if k % 2 else "")).encode()
        out.append(bytes(s[:max_len]))
    return out


def test_long_region_behind_many_short_ones(monkeypatch):
    """More candidates than the pass can hold (text made of a hundred medium runs) IN FRONT of a long periodic stretch: the pass
    reads on and keeps the longest candidates, so the stretch is in the table (overflow flag set) and the walk jumps over it --
    a prefix an attacker can build must not switch the acceleration off."""
    import time
    import torch
    rng = np.random.default_rng(64)
    strings = []
    for q_word in (b"a", b"ab", b"aab", b"abbabba"):
        head = b"".join(bytes([97 + (k % 2)]) * int(rng.integers(70, 130)) for k in range(100))      # ~100 runs of 70..130 bytes
        strings.append(head + q_word * (60000 // len(q_word)) + b"c")
    tabs = scan(strings)
    check_tables(strings, tabs)
    for s, row in zip(strings, tabs):
        entries, overflow = decode(row)
        assert overflow
        longest = max(hi - lo for lo, hi, q in entries)
        assert longest >= 59000, (longest, entries)
    # and the walk uses it: example 1's automaton on a^n behind such a prefix must not take time proportional to n^2
    blob = image.blob_from_dump(oracle_lib.load_dump("ex1_plain"))
    batch = [strings[0]] * 256
    data, off = oracle_lib.pack(batch)
    d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    d_bytes[:len(data)] = torch.from_numpy(data.copy())
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    for engine in ("table", "jit"):
        monkeypatch.setenv("MFA_WALK", engine)
        img = capi.Image(blob)
        img.match_tensors(d_bytes, d_off)
        got = img.match_tensors(d_bytes, d_off)
        torch.cuda.synchronize()
        assert img.last_kernel_ms(0) < 50.0, (engine, img.last_kernel_ms(0))      # every step executed: seconds
        assert list(got.cpu().numpy()[:2]) == list(oracle_lib.OracleImage(blob).match(batch[:2]))


def test_counted_waits_against_full_waits(monkeypatch):
    """The row loop of the region pass waits with counted `s_waitcnt vmcnt(N)` that rely on the order of its requests; a build
    that waits for everything (MFA_REGION_SAFE_WAITS=1) must produce the same tables on a large fuzz batch."""
    rng = np.random.default_rng(777)
    strings = _fuzz_strings(rng, 600, 9000) + _fuzz_strings(rng, 60, 70000)
    fast = scan(strings, 3)
    monkeypatch.setenv("MFA_REGION_SAFE_WAITS", "1")
    safe = scan(strings, 3)
    assert np.array_equal(fast[:, 0], safe[:, 0])
    for k in range(len(strings)):
        cnt = int(fast[k, 0]) & 0xff
        assert np.array_equal(fast[k, :1 + cnt], safe[k, :1 + cnt]), k


def test_rotated_assignment_changes_nothing(monkeypatch):
    """Which wave scans which string is rotated within blocks of 64 (so that a cost pattern with a small period in the batch does not fall to
    the same XCDs); every string must still be scanned exactly once, with the same table as when wave w takes string w."""
    rng = np.random.default_rng(778)
    strings = _fuzz_strings(rng, 64 * 9 + 17, 5000)          # nine whole blocks of 64 and a ragged one
    rotated = scan(strings, 1)
    monkeypatch.setenv("MFA_REGION_ROTATE", "0")
    plain = scan(strings, 1)
    assert np.array_equal(rotated[:, 0], plain[:, 0])
    for k in range(len(strings)):
        cnt = int(plain[k, 0]) & 0xff
        assert np.array_equal(rotated[k, :1 + cnt], plain[k, :1 + cnt]), k


@pytest.mark.parametrize("pad", [0, 5, 15])
def test_region_tables(pad):
    rng = np.random.default_rng(4242 + pad)
    strings = [b"", b"a", b"ab", b"a" * 63, b"a" * 64, b"a" * 65, b"a" * 200, b"ab" * 100, b"abcabcab" * 40, b"x" + b"a" * 300 + b"y" + b"a" * 300]
    strings += _fuzz_strings(rng, 400, 5000)
    strings += _fuzz_strings(rng, 40, 70000)
    tabs = scan(strings, pad)
    assert check_tables(strings, tabs) > 300


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_region_tables_every_rotation_depth(depth, monkeypatch):
    """the region kernel's development variants (rows per wave in rotation, MFA_REGION_DEPTH) give the same tables"""
    rng = np.random.default_rng(777)
    strings = [b"", b"a" * 64, b"ab" * 600, b"x" + b"a" * 3000 + b"y" + b"abc" * 700] + _fuzz_strings(rng, 200, 9000)
    monkeypatch.delenv("MFA_REGION_DEPTH", raising=False)
    want = scan(strings, 3)
    monkeypatch.setenv("MFA_REGION_DEPTH", str(depth))
    got = scan(strings, 3)
    assert np.array_equal(got, want)
    check_tables(strings, got)


def test_overflowing_table_keeps_the_first_regions():
    """A string with hundreds of medium runs: the table keeps the longest stretches AND the first ones -- a walk that dies at the
    first foreign byte (BASELINE configs[2]'s noise strings under example 1) must not step through the first run byte by byte."""
    rng = np.random.default_rng(65)
    strings = [b"".join(bytes([97 + (k % 2)]) * int(rng.integers(100, 131)) for k in range(300)),
               bytes(np.where(rng.random(65536) < 0.01, 98, 97).astype(np.uint8).tobytes())]
    tabs = scan(strings)
    check_tables(strings, tabs)
    for s, row in zip(strings, tabs):
        entries, overflow = decode(row)
        assert overflow and len(entries) == 15
        # the first run of at least 128 equal bytes is a candidate whatever its alignment (six whole clean blocks): it, or an earlier one, is there
        k, first = 0, None
        while k < len(s) and first is None:
            j = k
            while j < len(s) and s[j] == s[k]:
                j += 1
            if j - k >= 128:
                first = (k, j)
            k = j
        assert first is not None
        assert any(lo <= first[0] and q == 1 for lo, hi, q in entries), (first, sorted(entries)[:4])
        assert any(lo == first[0] and hi == first[1] and q == 1 for lo, hi, q in entries) or min(lo for lo, hi, q in entries) < first[0]


def test_region_table_overflow_keeps_true_regions():
    """more long regions than a table holds: flagged, and what is kept is still true"""
    s = b"".join(bytes([97 + (k % 3)]) * 150 for k in range(60))
    t = b"".join((b"ab" if k % 2 else b"cd") * 100 for k in range(40))
    tabs = scan([s, t, b"a" * 1000])
    e0, o0 = decode(tabs[0])
    e1, o1 = decode(tabs[1])
    e2, o2 = decode(tabs[2])
    assert o0 and o1 and not o2 and len(e0) >= 8 and len(e1) >= 8 and e2 == [(0, 1000, 1)]
    check_tables([s, t, b"a" * 1000], tabs)


def _big_batch(ex, n, seed, lo=512, hi=8192):
    import torch
    sizes = corpus.pump_sizes(n, seed, lo, hi)
    ws = (np.arange(n) % 2) == 0
    d_bytes, d_off = corpus.device_batch(ex, sizes, ws, torch.device("cuda", 0))
    return d_bytes, d_off, sizes, ws


def test_one_image_two_streams_back_to_back():
    """ONE image, two streams, two large different batches launched back to back: every launch has its own ticket
    counter, scratch, region table and events, so the overlapping kernels cannot take each other's strings."""
    import torch
    blob = image.blob_from_dump(oracle_lib.load_dump("ex6_plain"))
    img = capi.Image(blob)
    n = 60000
    ba, oa, sza, wsa = _big_batch(6, n, 11)
    bb, ob, szb, wsb = _big_batch(6, n, 22)
    want_a = img.match_tensors(ba, oa).clone()
    want_b = img.match_tensors(bb, ob).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        ra = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        rb = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        img.match_tensors(ba, oa, ra, stream=s1)
        img.match_tensors(bb, ob, rb, stream=s2)
        img.match_tensors(bb, ob, rb, stream=s2)            # the same stream again: its context is reused in stream order
        torch.cuda.synchronize()
        assert torch.equal(ra, want_a) and torch.equal(rb, want_b)
    # anchor: the short strings of batch a against the CPU restatement
    short = [k for k in range(n) if sza[k] <= 1500][:60]
    strings = corpus.host_strings(6, sza[short], wsa[short])
    assert list(want_a[short].cpu().numpy()) == list(oracle_lib.OracleImage(blob).match(strings))


def test_one_image_two_host_threads():
    """the same from two host threads, each with its own stream"""
    import torch
    blob = image.blob_from_dump(oracle_lib.load_dump("ex3_plain"))
    img = capi.Image(blob)
    n = 40000
    batches = [_big_batch(3, n, 100 + t) for t in range(2)]
    want = []
    for b, o, _, _ in batches:
        want.append(img.match_tensors(b, o).clone())
    torch.cuda.synchronize()
    got = [None, None]
    errs = []

    def work(t):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            b, o, _, _ = batches[t]
            for _ in range(4):
                r = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
                img.match_tensors(b, o, r, stream=st)
                st.synchronize()
                got[t] = r
                if not torch.equal(r, want[t]):
                    errs.append("thread %d: results differ" % t)
        except Exception as e:                                # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    sz, ws = batches[0][2], batches[0][3]
    short = [k for k in range(n) if sz[k] <= 1500][:60]
    strings = corpus.host_strings(3, sz[short], ws[short])
    assert list(want[0][short].cpu().numpy()) == list(oracle_lib.OracleImage(blob).match(strings))


def test_shared_region_table_for_several_automata():
    """mfa_match_batch_regions: one region pass, several automata over the same strings (and a sub-range of them)"""
    import torch
    n = 3000
    d_bytes, d_off, sizes, ws = _big_batch(2, n, 5, 200, 3000)
    tab = capi.region_scan(d_bytes, d_off)
    for name in ("ex2_plain", "ex2_reverse", "ex9_plain"):
        blob = image.blob_from_dump(oracle_lib.load_dump(name))
        img = capi.Image(blob)
        own = img.match_tensors(d_bytes, d_off).clone()
        shared = img.match_tensors_regions(d_bytes, d_off, tab).clone()
        none = img.match_tensors_regions(d_bytes, d_off, None).clone()
        part = img.match_tensors_regions(d_bytes, d_off[1000:2001].clone(), tab[1000:2000]).clone()
        torch.cuda.synchronize()
        assert torch.equal(own, shared) and torch.equal(own, none) and torch.equal(own[1000:2000], part), name
        short = [k for k in range(n) if sizes[k] <= 900][:40]
        strings = corpus.host_strings(2, sizes[short], ws[short])
        assert list(own[short].cpu().numpy()) == list(oracle_lib.OracleImage(blob).match(strings)), name


def test_bench_line_smoke():
    """bench.py end to end at a small size: one JSON line with the contract's fields, the roofline and the cpu_baseline objects"""
    import json
    import subprocess
    import sys
    root = oracle_lib.ROOT
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--strings-per-example", "3000", "--steps", "2", "--warmup", "1",
                        "--no-secondary"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "ranks_seen"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["dtype"] == "u8"
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["peak"] == 8000.0 and r["region_scan_kernel"]["launches_per_step"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["parity"] is True and c["cores"] >= 1
    # parity at scale: a seeded 1 % of every example's strings, no length cap, against the CPU restatement
    ps = d["parity_sample"]
    assert ps["strings"] >= 10 * 30 and ps["mismatches"] == 0 and ps["max_len"] > 20000
    assert c["parity_restatement"]["strings"] == ps["strings"]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_rehearsal(scaling):
    """The N > 1 path of bench.py in every GPU test run: two ranks on this one GPU over gloo (MFA_BENCH_REHEARSE=1; real runs
    use one GPU per rank and RCCL).  Both ranks must show up in the line, every rank's bitmap must reach rank 0 (bench.py itself
    checks rank 0's segment of the gathered vector against its local results), and under strong scaling the two ranks' strings
    and bytes must add up to the job's ONE batch."""
    import json
    import subprocess
    import sys
    root = oracle_lib.ROOT
    env = dict(os.environ)
    env["MFA_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--strings-per-example", "3000", "--steps", "2", "--warmup", "1",
                        "--no-secondary", "--no-cpu-baseline", "--scaling", scaling], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == scaling
    assert len(d["bytes_by_rank"]) == 2 and min(d["bytes_by_rank"]) > 0 and len(d["strings_by_rank"]) == 2
    if scaling == "strong":
        assert sum(d["strings_by_rank"]) == 10 * 3000
        lo, hi = sorted(d["bytes_by_rank"])
        assert hi - lo <= 70000                                   # the cut is balanced by bytes (to within one string)
    else:
        assert d["strings_by_rank"] == [30000, 30000]



def test_bench_single_rank_rccl():
    """RCCL executed once before an 8-GPU run does: bench.py as ONE rank that still goes through torch.distributed with the
    `nccl` backend (MFA_BENCH_FORCE_DIST=1): communicator initialised on the device, all_gather / gather / all_reduce / barrier on
    device tensors, the gathered bitmap compared with the local results by bench.py itself.  Two `nccl` ranks cannot share the one
    GPU of this box: this is what a 1-GPU lease can prove of sharding.gather_results and bench.py's N > 1 branch."""
    import json
    import subprocess
    import sys
    root = oracle_lib.ROOT
    env = dict(os.environ)
    env.update({"MFA_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1"})
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--strings-per-example", "3000", "--steps", "2", "--warmup", "1",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["backend"] == "nccl"
    assert d["strings_by_rank"] == [30000] and d["bytes_by_rank"][0] > 0
    assert "nccl" in d["config"]["exchange"] and d["parity_sample"]["mismatches"] == 0
    assert "shard needs about" in p.stderr                        # the HBM preflight of multi-rank runs


def test_gated_walks_see_fresh_tables(monkeypatch):
    """The gate of mfa_match_mixed (ONE region launch, the walks of a group released by a counter the region kernel raises; walk_launch.hip,
    regions.hip: gate_signal): the walk kernels start while the region kernel is still running and read table rows that other CUs -- other
    XCDs -- have just written.  Two batches of the SAME shape and DIFFERENT content go through one mixed object in turn (one table buffer,
    rewritten by every call): a walk that read a stale row -- the other batch's regions -- would walk over bytes that do not repeat and
    answer wrongly.  Every call is compared with the ungated schedule (events behind per-group region launches) and a sample with the oracle."""
    import torch
    from mfa_amd import corpus
    dev = torch.device("cuda", 0)
    layout = [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    n_per = 9000
    blobs = [image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)) for ex in layout]
    images = [capi.Image(b) for b in blobs]

    def batch(variant):
        parts_b, parts_o, seg, pos_b, samples = [], [], [0], 0, []
        for ex in layout:
            sizes = corpus.pump_sizes(n_per, 0x5EED0100 + ex, 200, 9000)          # the same lengths in both variants ...
            ws = ((np.arange(n_per) + variant) % 2) == 0                            # ... but who has the suffix alternates: other answers
            if variant:                                                            # ... and other strings: the pump sizes of the neighbours
                sizes = np.roll(sizes, 1)
            b, o = corpus.device_batch(ex, sizes, ws, dev)
            nb = int(o[-1].item())
            parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
            short = [k for k in range(n_per) if sizes[k] <= 1500][:25]
            samples.append((short, corpus.host_strings(ex, sizes[short], ws[short])))
        return (torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)]),
                torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)]), seg, samples, pos_b)

    A, B = batch(0), batch(1)
    monkeypatch.setenv("MFA_WALK", "table")
    monkeypatch.setenv("MFA_MIXED_CUTS", "0.15,0.3,0.45,0.6,0.75,0.87,0.95")
    mx = capi.Mixed(images)
    ref = {}
    monkeypatch.setenv("MFA_MIXED_GATE", "0")
    for tag, (bts, off, seg, samples, total) in (("A", A), ("B", B)):
        ref[tag] = mx.match_tensors(bts, off, seg).clone()
        torch.cuda.synchronize()
        assert not mx.last_launches()["gated"] and mx.last_launches()["region_launches"] == 8
        for k, (short, strings) in enumerate(samples):
            want = oracle_lib.OracleImage(blobs[k]).match(strings)
            assert np.array_equal(ref[tag][seg[k]:seg[k + 1]][short].cpu().numpy(), want), (tag, layout[k])
    assert not torch.equal(ref["A"], ref["B"])
    monkeypatch.setenv("MFA_MIXED_GATE", "1")
    res = torch.empty_like(ref["A"])
    for r in range(40):
        tag, (bts, off, seg, samples, total) = (("A", A), ("B", B))[r % 2]
        res.fill_(7)
        mx.match_tensors(bts, off, seg, res, total_bytes=total)
        torch.cuda.synchronize()
        la = mx.last_launches()
        assert la["gated"] and la["region_launches"] == 1 and la["groups"] == 8
        bad = torch.nonzero(res != ref[tag]).flatten()
        assert bad.numel() == 0, "round %d (%s): %d answers differ from the ungated schedule, first at string %d" % (r, tag, bad.numel(), int(bad[0]))
    # the tables themselves: the gated launch writes what per-group launches write
    monkeypatch.setenv("MFA_MIXED_GATE", "0")
    mx.close()


@pytest.mark.parametrize("name,bound", [("ex8_reverse", 125.0), ("ex8_bnf", 60.0)])
def test_executed_steps_of_the_77_node_automata(name, bound):
    """BASELINE configs[4]'s worst line -- example 8 `-reverse` (77 nodes, reversed scan) on pump-only strings -- and the same automaton's
    forward twin (`-bnf`): the walk must get through a string in a bounded number of EXECUTED steps whatever its length (1-64 KiB here:
    1 900 MB of input, ~95 / ~40 steps per string), not in wall time: counted by the MFA_WALK_STATS build of the kernel (lane steps /
    strings).  Round 3's probe control took 156 steps per string on the reversed automaton (periods tried in turn, a period that had
    failed once kept for the rest of the string); the bound fails if the probes stop finding the list's own period."""
    import re
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(oracle_lib.ROOT, "tools", "rev8_run.py"), "25000", "1", "1", name], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    m = re.search(r"(\d+) strings, wave-iterations (\d+) \(dual (\d+)\), lane steps (\d+), skipped (\d+)", p.stderr)
    assert m, p.stderr[-1500:]
    strings, iters, dual, steps, skipped = (int(x) for x in m.groups())
    assert strings == 25000 and "(25000 accepted)" in p.stdout
    assert steps / strings <= bound, "%.1f executed steps per string" % (steps / strings)
    assert skipped > 100 * steps                                      # nearly every character is jumped over


def test_bench_under_torchrun_rehearsal():
    """The driver's multi-GPU launch line, word for word -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with N = 2 ranks rehearsed on this one GPU over gloo
    (MFA_BENCH_REHEARSE=1): bench.py must take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher, print ONE line on rank 0
    and leave no rank behind."""
    import json
    import socket
    import subprocess
    import sys
    root = oracle_lib.ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env["MFA_BENCH_REHEARSE"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--strings-per-example", "3000", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["strings_by_rank"] == [30000, 30000]
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
