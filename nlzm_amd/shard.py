"""Multi-GPU sharding of the compress path.

One NLZM stream is serial end to end (dictionary, finder state, model and rep set flow
through the whole file, NLZM.cpp:1729-1770), so the path shards only as independent
blocks: rank i of k compresses bytes [i*ceil(N/k), min(N,(i+1)*ceil(N/k))) as its own
stream (own header, window, model, terminator).  There is no data-path collective; the
only exchange is the final gather of the k streams onto rank 0 (RCCL over xGMI: one
all_gather of the lengths, then point-to-point sends).  Streams are self-delimiting
(frame headers carry sizes, NLZM.cpp:645-663; terminator num_ops == 0, :646-648), so the
gathered artifact is their plain concatenation and `split_streams` recovers them.
"""
from __future__ import annotations

import numpy as np


def block_range(n: int, k: int, i: int) -> tuple[int, int]:
    per = -(-n // k) if k > 0 else n
    lo = min(n, i * per)
    return lo, min(n, lo + per)


def gather_streams(local, rank: int, world: int, device=None):
    """Gather every rank's stream (1-D uint8 torch tensor) onto rank 0.

    Returns (list of uint8 tensors on rank 0 | None elsewhere, list of lengths).
    Works with the nccl (= RCCL) backend on GPU tensors and with gloo on CPU tensors.
    """
    import torch
    import torch.distributed as dist

    dev = local.device if device is None else device
    ln = torch.tensor([local.numel()], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, ln)
    lens = [int(x.item()) for x in lens]
    if world == 1:
        return [local], lens
    if rank == 0:
        parts = [local] + [torch.empty(lens[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
        reqs = [dist.irecv(parts[r], src=r) for r in range(1, world) if lens[r]]
        for q in reqs:
            q.wait()
        return parts, lens
    if local.numel():
        dist.send(local, dst=0)
    return None, lens


def split_streams(blob: bytes) -> list[bytes]:
    """Split a concatenation of NLZM streams at their terminators."""
    out = []
    pos = 0
    n = len(blob)
    while pos < n:
        start = pos
        if pos + 8 > n:
            raise ValueError("truncated stream header")
        pos += 4
        while True:
            if pos + 4 > n:
                raise ValueError("truncated frame")
            ops = int.from_bytes(blob[pos:pos + 4], "big")
            if ops == 0:
                pos += 4
                break
            if pos + 12 > n:
                raise ValueError("truncated frame header")
            nb = int.from_bytes(blob[pos + 4:pos + 8], "big")
            nr = int.from_bytes(blob[pos + 8:pos + 12], "big")
            # a frame holds its 12 header bytes and 4 bit-flush bytes, then the four rANS states (NLZM.cpp:591-628)
            if nb < 12 or nr < 16 or pos + nb + nr > n:
                raise ValueError(f"malformed frame at byte {pos}: sizes {nb}, {nr}")
            pos += nb + nr
        out.append(blob[start:pos])
    return out


def concat(parts) -> bytes:
    return b"".join(bytes(np.asarray(p.cpu() if hasattr(p, "cpu") else p, dtype=np.uint8)) for p in parts)
