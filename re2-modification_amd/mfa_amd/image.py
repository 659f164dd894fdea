"""Automaton images: text dump <-> binary blob (include/mfa_image_format.h).

The text form is what oracle/ref_harness.cpp prints for a reference automaton
(`dump`), and what the host front-end prints for its own (`diploma -dump`): one
`node <list index> <order rank> <n_edges>` line per node followed by its
`edge <label-hex|-> <target list index> [o<cell>|c<cell> ...]` lines.  The blob
numbers nodes by ORDER RANK (reference: pointer order of std::set<MemoryState>,
automata.h:12-13; allocation order under the canonical model).
"""
import struct

MAGIC = 0x4941464D
VERSION = 1
KIND_NFA, KIND_MFA = 0, 1
EDGE_EPS = 1
ACT_OPEN, ACT_CLOSE = 1, 2


class ImageError(ValueError):
    pass


def parse_dump(text):
    """Parse a text dump into a dict; nodes stay in list order, ranks recorded."""
    img = {"kind": None, "reversed": 0, "start": 0, "finish": 0, "nodes": []}
    for line in text.splitlines():
        f = line.split()
        if not f:
            continue
        if f[0] == "kind":
            img["kind"] = {"nfa": KIND_NFA, "mfa": KIND_MFA}[f[1]]
        elif f[0] == "reversed":
            img["reversed"] = int(f[1])
        elif f[0] == "nodes":
            img["n_nodes"] = int(f[1])
        elif f[0] == "start":
            img["start"] = int(f[1])
        elif f[0] == "finish":
            img["finish"] = int(f[1])
        elif f[0] == "node":
            img["nodes"].append({"rank": int(f[2]), "edges": []})
        elif f[0] == "edge":
            label = None if f[1] == "-" else bytes.fromhex(f[1])
            acts = {}
            for a in f[3:]:
                acts[int(a[1:])] = ACT_OPEN if a[0] == "o" else ACT_CLOSE
            img["nodes"][-1]["edges"].append((label, int(f[2]), acts))
        else:
            raise ImageError("bad dump line: %r" % line)
    if img["kind"] is None or len(img["nodes"]) != img.get("n_nodes"):
        raise ImageError("truncated dump")
    return img


def to_blob(img):
    """Serialise a parsed dump; nodes renumbered by rank."""
    n = len(img["nodes"])
    ranks = [nd["rank"] for nd in img["nodes"]]
    if sorted(ranks) != list(range(n)):
        raise ImageError("ranks are not a permutation")
    by_rank = [None] * n
    for idx, nd in enumerate(img["nodes"]):
        by_rank[nd["rank"]] = idx
    edge_begin, edges, n_cells = [0], [], 0
    for r in range(n):
        for label, to, acts in img["nodes"][by_rank[r]]["edges"]:
            flags, lab, actions = 0, 0, 0
            if label is None:
                flags |= EDGE_EPS
            else:
                if len(label) != 1:
                    raise ImageError("multi-byte edge label %r" % (label,))
                lab = label[0]
                if 0x31 <= lab <= 0x39:
                    n_cells = max(n_cells, lab - 0x30)
            for cell, act in acts.items():
                if not 1 <= cell <= 9:
                    raise ImageError("bad cell %d" % cell)
                actions |= act << (2 * cell)
                n_cells = max(n_cells, cell)
            edges.append(struct.pack("<BBHI", lab, flags, ranks[to], actions))
        edge_begin.append(len(edges))
    hdr = struct.pack("<10I", MAGIC, VERSION, img["kind"], img["reversed"], n, len(edges),
                      ranks[img["start"]], ranks[img["finish"]], n_cells, 0)
    return hdr + struct.pack("<%dI" % (n + 1), *edge_begin) + b"".join(edges)


def blob_from_dump(text):
    return to_blob(parse_dump(text))


def blob_info(blob):
    f = struct.unpack_from("<10I", blob, 0)
    return {"kind": f[2], "reversed": f[3], "n_nodes": f[4], "n_edges": f[5], "start": f[6], "finish": f[7],
            "n_cells": f[8]}
