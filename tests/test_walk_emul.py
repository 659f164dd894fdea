"""The walk kernel's own source (csrc/walk_core.h + csrc/walk_tables.cpp) compiled for the host as a wave of one lane
(tests/emul/), against the oracle: the table builder, the list step, the jump logic, spills (tiny list capacity) and mixed
batches are checked here without a GPU.  The kernel proper is checked by the -m gpu tests."""
import glob
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib
from mfa_amd import corpus, image

EMUL_DIR = os.path.join(oracle_lib.ROOT, "tests", "emul")


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "walk_emul")
    subprocess.check_call([os.path.join(EMUL_DIR, "build.sh"), exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return exe


@pytest.fixture(scope="module")
def emul_map(tmp_path_factory):
    """the same source as the kernel for long lists compiles it: WALK_NODE_MAP=1 (a node's entry found through a per-lane map, walk_core.h: insert)"""
    exe = str(tmp_path_factory.mktemp("emul_map") / "walk_emul")
    subprocess.check_call([os.path.join(EMUL_DIR, "build.sh"), exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, EMUL_FLAGS="-DWALK_NODE_MAP=1"))
    return exe


def mfa_names():
    names = []
    for p in sorted(glob.glob(os.path.join(oracle_lib.GOLDEN, "images", "*.dump"))):
        name = os.path.basename(p)[:-5]
        if image.blob_info(image.blob_from_dump(oracle_lib.load_dump(name)))["kind"] == image.KIND_MFA:
            names.append(name)
    return names


def example_of(name):
    if not name.startswith("ex"):
        return None
    ex = int("".join(ch for ch in name.split("_")[0] if ch.isdigit()))
    return ex if ex in corpus.ALL_EXAMPLES else None


def long_strings(name, rng):
    out = []
    ex = example_of(name)
    if ex is not None:
        regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
        for n in (70, 200, 700, 2500, 6000):
            for ws in (False, True):
                s = (prefix + corpus.pumped_string(n, pump) + (suffix if ws else "")).encode()
                out.append(s)
                b = bytearray(s)
                b[rng.randrange(len(b))] = ord(rng.choice("abc"))
                out.append(bytes(b))
                out.append(s[:len(s) // 2] + s[len(s) // 3:])
    for _ in range(6):
        u = "".join(rng.choice("ab") for _ in range(rng.randint(1, 7)))
        out.append((u * rng.randint(20, 400)).encode() + rng.choice([b"", b"b", b"c", b"ab"]))
    return out


def run_emul(exe, args, strings):
    p = subprocess.run([exe] + [str(a) for a in args], input=b"".join(s + b"\n" for s in strings), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    return np.array([int(x) for x in p.stdout.split()], dtype=np.uint8)


@pytest.mark.parametrize("name", mfa_names())
def test_walk_source_against_oracle(emul, name, tmp_path):
    rng = random.Random(hash(name) & 0xffff)
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    path = tmp_path / "a.blob"
    path.write_bytes(blob)
    strings = [s for s in oracle_lib.load_set("abc7")[::7] + oracle_lib.load_set("rnd")[:300] + oracle_lib.load_set("odd") if s and b"\n" not in s]
    strings += long_strings(name, rng)
    want = oracle_lib.OracleImage(blob).match(strings)
    # (list capacity, regions): capacity 2-3 makes most lists spill; with regions the lane jumps over periodic stretches
    for cap, accel in ((8, 0), (2, 0), (8, 1), (3, 1)):
        got = run_emul(emul, [path, cap, accel], strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s capacity %d regions %d: %d mismatches, first %r want %d got %d" % (
            name, cap, accel, bad.size, strings[bad[0]][:60], want[bad[0]], got[bad[0]])


@pytest.mark.parametrize("name", mfa_names())
def test_walk_source_with_node_map(emul_map, name, tmp_path):
    """the long-list kernel's form of the insertion (WALK_NODE_MAP) on every memory automaton, spilling lists and re-executed steps included"""
    rng = random.Random(hash(name) & 0xfff)
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    path = tmp_path / "a.blob"
    path.write_bytes(blob)
    strings = [s for s in oracle_lib.load_set("abc7")[::11] + oracle_lib.load_set("rnd")[:200] + oracle_lib.load_set("odd") if s and b"\n" not in s]
    strings += long_strings(name, rng) + text_without_stretches(rng, 6)
    want = oracle_lib.OracleImage(blob).match(strings)
    for cap, accel in ((2, 0), (3, 1)):
        got = run_emul(emul_map, [path, cap, accel], strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s capacity %d regions %d: %d mismatches, first %r want %d got %d" % (
            name, cap, accel, bad.size, strings[bad[0]][:60], want[bad[0]], got[bad[0]])


def test_mixed_segments(emul, tmp_path):
    """several automata, with different cell counts, walked by one launch: segment k of the batch belongs to automaton k"""
    names = ["ex1_plain", "ex3_plain", "ex2_plain", "ex5_plain", "ex9_plain"]
    strings, seg, want, paths = [], [], [], []
    for k, nm in enumerate(names):
        blob = image.blob_from_dump(oracle_lib.load_dump(nm))
        p = tmp_path / ("m%d.blob" % k)
        p.write_bytes(blob)
        paths.append(p)
        ss = [s for s in oracle_lib.load_set("rnd")[:200] if s] + corpus.host_strings(example_of(nm), np.array([300, 4000, 9000]), [True, False, True])
        seg.append(len(strings))
        strings += ss
        want += list(oracle_lib.OracleImage(blob).match(ss))
    args = [paths[0], 3, 1]
    for k in range(1, len(names)):
        args += [paths[k], seg[k]]
    got = run_emul(emul, args, strings)
    assert np.array_equal(got, np.array(want, dtype=np.uint8))


def text_without_stretches(rng, n):
    """strings that hold no periodic stretch of 64 bytes: short runs of a's cut by b's, and coin tosses"""
    out = []
    for _ in range(n):
        out.append(b"".join((b"a" * rng.randint(1, 20) + b"b") for _ in range(rng.randint(5, 120))))
        out.append(bytes(rng.choice(b"ab") for _ in range(rng.randint(200, 900))))
    return out


@pytest.mark.parametrize("name", ["ex1_plain", "ex6_plain", "ex9_plain", "ex3_reverse", "ex8_reverse", "ex5_bnf"])
def test_lean_pass(emul, name, tmp_path):
    """Strings without a periodic stretch (an empty region-table row) are handed to the lean walk -- the plain step only, walk_wave_lean --
    through a queue; strings with stretches stay with the first pass.  Both kinds in one batch, several automata's worth of segments."""
    rng = random.Random(11 + len(name))
    blob = image.blob_from_dump(oracle_lib.load_dump(name))
    path = tmp_path / "a.blob"
    path.write_bytes(blob)
    strings = text_without_stretches(rng, 30) + [b"a" * 700 + b"b", b"ab" * 300, b"a" * 255, b"b"]
    rng.shuffle(strings)
    want = oracle_lib.OracleImage(blob).match(strings)
    p = subprocess.run([emul, str(path), "3", "1"], input=b"".join(s + b"\n" for s in strings), capture_output=True)
    assert p.returncode == 0
    assert [int(x) for x in p.stdout.split()] == list(want)
    handed_on = int(p.stderr.decode().split(" of the strings walked by the lean pass")[0].split()[-1])
    assert 40 <= handed_on <= 60                                    # the texts of 256 bytes and more; not the periodic ones, not the short ones
