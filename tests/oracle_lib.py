"""ctypes front of oracle/liboracle.so (the CPU restatement) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("steps", "evaluations", "cell_reads", "compare_bytes", "max_states", "variables")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "mfa_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "restatement"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        L.mfa_oracle_image_load.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        L.mfa_oracle_image_free.argtypes = [ctypes.c_void_p]
        L.mfa_oracle_match_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.POINTER(Stats)]
        _lib = L
    return _lib


def pack(strings):
    """list of bytes -> (uint8 array, uint64 offsets[n+1])"""
    lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=len(strings))
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    data = np.frombuffer(b"".join(strings), dtype=np.uint8) if off[-1] else np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(data), off


class OracleImage:
    def __init__(self, blob):
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(blob, len(blob))
        rc = lib().mfa_oracle_image_load(buf, len(blob), ctypes.byref(self._h))
        if rc:
            raise ValueError("oracle image load failed: %d" % rc)

    def match_packed(self, data, off, stats=None):
        n = len(off) - 1
        res = np.zeros(max(n, 1), dtype=np.uint8)
        rc = lib().mfa_oracle_match_batch(self._h, data.ctypes.data, off.ctypes.data, n, res.ctypes.data,
                                          ctypes.byref(stats) if stats is not None else None)
        if rc:
            raise RuntimeError("oracle match failed: %d" % rc)
        return res[:n]

    def match(self, strings, stats=None):
        data, off = pack(strings)
        return self.match_packed(data, off, stats)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mfa_oracle_image_free(self._h)
            self._h = None


def load_set(name):
    with open(os.path.join(GOLDEN, "strings", name + ".txt"), "rb") as f:
        body = f.read()
    lines = body.split(b"\n")
    assert lines[-1] == b""
    return lines[:-1]


def load_bits(auto, sset):
    with open(os.path.join(GOLDEN, "results", "%s.%s.bits" % (auto, sset))) as f:
        return np.frombuffer(f.read().strip().encode(), dtype=np.uint8) - ord("0")


def load_dump(auto):
    with open(os.path.join(GOLDEN, "images", auto + ".dump")) as f:
        return f.read()
