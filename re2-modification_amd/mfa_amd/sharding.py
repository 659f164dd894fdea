"""Multi-GPU data parallelism of the match path: one process per GPU, strings are independent.

The batch is cut into `world` contiguous ranges balanced by BYTES (per-string cost is proportional to
length for forward attack strings), every rank matches its own range with the replicated automaton
image, and the only exchange is a gather of the per-rank result bitmaps to rank 0 (RCCL over xGMI on
GPUs -- torch.distributed backend "nccl" -- or gloo on CPUs in the tests).  No collective runs inside
the match.  Order is restored from the partition: rank r's bits are strings cuts[r] .. cuts[r+1]-1.
"""
import os

import numpy as np


def partition_by_bytes(offsets, world):
    """offsets: int array (n+1).  Returns cuts (world+1): rank r owns strings cuts[r]:cuts[r+1]."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    total = int(offsets[-1] - offsets[0])
    cuts = [0]
    for r in range(1, world):
        target = offsets[0] + (total * r) // world
        k = int(np.searchsorted(offsets, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return np.asarray(cuts, dtype=np.int64)


def pack_bitmap(results):
    """torch uint8 result tensor (n; 1 = accepted, 0 = rejected, 2 = not matched: longer than the device limit) -> uint8 bitmap
    (ceil(n/8)), bit k%8 of byte k//8 = string k accepted.  Only the value 1 sets a bit; count_unmatched() reports the 2s."""
    import torch
    n = results.numel()
    if results.is_cuda and results.is_contiguous() and n and os.environ.get("MFA_TORCH_PACK", "") != "1":
        from . import capi                                   # one kernel of the library instead of five of torch's (behind every step of a batch)
        return capi.pack_result_bitmap(results)
    r = (results == 1).to(torch.uint8)
    pad = (-n) % 8
    if pad:
        r = torch.cat([r, r.new_zeros(pad)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=results.device)
    return (r.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8)


def count_unmatched(results):
    """strings the device refused (result code 2: longer than MFA_MAX_STRING_BYTES)"""
    return int((results > 1).sum().item())


def unpack_bitmap(bitmap, n):
    import torch
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=bitmap.device)
    bits = (bitmap.view(-1, 1) & w) != 0
    return bits.reshape(-1)[:n].to(torch.uint8)


def gather_results(local_results, counts, dist, rank, world, dst=0, comm_device=None):
    """Gather every rank's 0/1 results to `dst` as bitmaps; returns the full result vector there (None
    elsewhere).  counts[r] = number of strings of rank r (known from the partition).  comm_device: where the
    collective's buffers live (the GPU for RCCL; "cpu" for gloo)."""
    import torch
    max_bytes = (max(int(c) for c in counts) + 7) // 8
    bm = pack_bitmap(local_results)
    if comm_device is not None:
        bm = bm.to(comm_device)
    if bm.numel() < max_bytes:                      # gather needs equal sizes
        bm = torch.cat([bm, bm.new_zeros(max_bytes - bm.numel())])
    if world == 1 and dist is None:
        return unpack_bitmap(bm, int(counts[0]))
    bufs = [torch.empty_like(bm) for _ in range(world)] if rank == dst else None
    dist.gather(bm, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([unpack_bitmap(bufs[r], int(counts[r])) for r in range(world)])
