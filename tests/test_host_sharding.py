"""The C++ host mirror spreads a packed batch over the HIP devices of the node (host/automata_host.cpp: Automata::match_packed,
north_star: "the string batch shards trivially across the 8 GPUs of one node"): the partition rule on the CPU, the sharded call on
the GPU box (one device there: several shards on it)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from mfa_amd import sharding

HOST = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host")


def test_partition_rule_is_the_python_one():
    """diploma_partition_by_bytes == mfa_amd.sharding.partition_by_bytes (what bench.py's strong scaling uses) on ragged batches:
    empty strings, a few giants, more parts than strings."""
    lib = ctypes.CDLL(os.path.join(HOST, "libdiploma_host.so"))
    fn = lib.diploma_partition_by_bytes
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(0, 60))
        lens = rng.integers(0, 50, size=n)
        if n and trial % 3 == 0:
            lens[rng.integers(0, n)] = 100000
        if trial % 7 == 0:
            lens[:] = 0
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        off += np.uint64(rng.integers(0, 1000))                     # a sub-batch does not start at 0
        for parts in (1, 2, 3, 8, 64):
            cuts = np.zeros(parts + 1, dtype=np.uint64)
            assert fn(off.ctypes.data, n, parts, cuts.ctypes.data) == 0
            want = sharding.partition_by_bytes(off.astype(np.int64), parts)
            assert list(cuts) == list(want), (trial, parts, lens.tolist())
            assert cuts[0] == 0 and cuts[-1] == n and all(cuts[k] <= cuts[k + 1] for k in range(parts))


@pytest.mark.gpu
def test_sharded_match_file(tmp_path):
    """`diploma -match-file mfa` on a 6 MB file: one device call, and the same batch cut into three shards matched by three host threads
    (DIPLOMA_FORCE_SHARDS=3; on a node with several GPUs the shards go to different devices): identical answers, checked against the
    oracle on a sample."""
    from mfa_amd import corpus, image
    ex = 6
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    sizes = corpus.pump_sizes(3000, 0x5EED0200, 50, 6000)
    ws = (np.arange(3000) % 2) == 0
    strings = corpus.host_strings(ex, sizes, ws)
    path = tmp_path / "in.txt"
    path.write_bytes(b"".join(s + b"\n" for s in strings))
    diploma = os.path.join(HOST, "diploma")
    outs = []
    for env_extra in ({"DIPLOMA_DEVICES": "1"}, {"DIPLOMA_FORCE_SHARDS": "3"}, {}):
        env = dict(os.environ)
        env.update(env_extra)
        p = subprocess.run([diploma, "-match-file", "mfa", str(path)], input=regex.encode() + b"\n", capture_output=True, cwd=tmp_path, env=env, timeout=300)
        assert p.returncode == 0, p.stderr
        outs.append([int(x) for x in p.stdout.split()[1:]])
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == 3000
    blob = image.blob_from_dump(oracle_lib.load_dump("ex6_plain"))
    idx = [k for k in range(3000) if sizes[k] < 1500][:80]
    assert [outs[1][k] for k in idx] == list(oracle_lib.OracleImage(blob).match([strings[k] for k in idx]))
