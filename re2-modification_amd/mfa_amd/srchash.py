"""A stamp of the kernel sources: what a measurement kept under profiles/ was taken with.

bench.py quotes `roofline.traffic` from a file of PMC counters collected in a separate run (profiles/*_traffic.json); the file
carries the stamp of the sources it was collected with, and bench.py quotes it only while the sources still have that stamp
(the GPU box receives the repository without .git, so the stamp is a hash of the files, not a commit)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


def kernel_source_hash():
    """sha256 over the device and launch sources of libmfa_hip.so (csrc/*.hip, *.h, *.cpp, Makefile; names and contents, sorted),
    first 16 hex digits"""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
            h.update(b"\0")
    return h.hexdigest()[:16]
