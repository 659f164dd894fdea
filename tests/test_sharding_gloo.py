"""The N>1 path on CPUs: world_size-2 gloo processes, byte-balanced partition, bitmap gather, order
restored on rank 0.  The per-shard 0/1 answers come from the oracle here (there is no GPU in this
test); on the GPU box bench.py runs the same partition/gather code around the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from mfa_amd import image, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    strings = oracle_lib.load_set("rnd") + oracle_lib.load_set("abc7")[:500] + [b"a" * 3000, b"a" * 2999 + b"b"]
    data, off = oracle_lib.pack(strings)
    cuts = sharding.partition_by_bytes(off, world)
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    img = oracle_lib.OracleImage(image.blob_from_dump(oracle_lib.load_dump(name)))
    local = torch.from_numpy(img.match(strings[lo:hi]).copy())
    counts = [int(cuts[r + 1] - cuts[r]) for r in range(world)]
    full = sharding.gather_results(local, counts, dist, rank, world)
    if rank == 0:
        want = img.match(strings)
        ok = full is not None and np.array_equal(full.numpy(), want)
        # the partition is balanced by bytes: no rank holds more than its share plus one string
        share = [int(off[cuts[r + 1]] - off[cuts[r]]) for r in range(world)]
        balanced = max(share) - min(share) <= max(len(s) for s in strings)
        with open(out_path, "w") as f:
            f.write("ok" if (ok and balanced) else "bad ok=%s balanced=%s share=%s" % (ok, balanced, share))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ex1_plain", "nfa_abb_thompson"])
def test_two_rank_gather(name, tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_partition_edges():
    off = np.array([0, 0, 10, 10, 30, 100, 100], dtype=np.int64)
    for world in (1, 2, 3, 8):
        cuts = sharding.partition_by_bytes(off, world)
        assert cuts[0] == 0 and cuts[-1] == 6 and np.all(np.diff(cuts) >= 0)
    assert list(sharding.partition_by_bytes(np.array([0], dtype=np.int64), 4)) == [0, 0, 0, 0, 0]


def test_bitmap_roundtrip():
    for n in (0, 1, 7, 8, 9, 1001):
        r = (torch.arange(n) % 3 == 0).to(torch.uint8)
        assert torch.equal(sharding.unpack_bitmap(sharding.pack_bitmap(r), n), r)


def test_bitmap_ignores_unmatched_code():
    """result code 2 (string longer than the device limit) is not an accept and must not leak into a neighbour's bit"""
    r = torch.tensor([1, 2, 0, 1, 2, 2, 1, 0, 2, 1], dtype=torch.uint8)
    bm = sharding.pack_bitmap(r)
    assert torch.equal(sharding.unpack_bitmap(bm, 10), (r == 1).to(torch.uint8))
    assert sharding.count_unmatched(r) == 4


@pytest.mark.gpu
def test_result_bitmap_kernel_against_the_tensor_form(monkeypatch):
    """mfa_pack_result_bitmap (what pack_bitmap uses for device tensors) against the tensor-library form of the same definition: ragged
    counts, a result vector that does not start at an 8-byte boundary, codes 0 / 1 / 2, and the round trip."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(0x5EED0071)
    base = torch.randint(0, 3, (100003,), dtype=torch.uint8, device=dev, generator=g)
    for lo, n in ((0, 1), (0, 7), (0, 8), (0, 9), (0, 4096), (3, 65537), (5, 99991), (0, 100003), (1, 63)):
        r = base[lo:lo + n]
        monkeypatch.setenv("MFA_TORCH_PACK", "1")
        want = sharding.pack_bitmap(r)
        monkeypatch.delenv("MFA_TORCH_PACK")
        got = sharding.pack_bitmap(r)
        torch.cuda.synchronize()
        assert got.dtype == torch.uint8 and got.numel() == (n + 7) // 8
        assert torch.equal(got, want), (lo, n)
        assert torch.equal(sharding.unpack_bitmap(got, n), (r == 1).to(torch.uint8)), (lo, n)
