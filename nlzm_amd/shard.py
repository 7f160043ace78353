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


# ---- what bench.py does with more than one rank (one process per GPU), kept here so that the CPU suite can drive it at world size 2 over gloo ----------

def golden_for_rank(golden_dir: str, world: int, rank: int, n: int, name_1gpu: str = "text_1g_w28"):
    """The reference's own stream for what rank `rank` of `world` compresses in bench.py: at N = 1 the 1e9-byte stand-in (full.json), at N > 1 that rank's
    block (gpus.json: one reference run per block, NLZM.cpp:1711 with the auto-shrink of :1716-1718).  None when there is no record of `n` bytes."""
    import json
    import os

    if world == 1:
        path = os.path.join(golden_dir, "full.json")
        gold = {c["name"]: c for c in json.load(open(path))["cases"]}.get(name_1gpu) if os.path.exists(path) else None
    else:
        path = os.path.join(golden_dir, "gpus.json")
        gold = {(r["world"], r["rank"]): r for r in json.load(open(path))["ranks"]}.get((world, rank)) if os.path.exists(path) else None
    return gold if gold and gold["size"] == n else None


def stream_matches(gold, stream_len: int, stream_sha256: str):
    """True / False against a golden record, None without one."""
    if not gold:
        return None
    return bool(gold["stream_size"] == stream_len and gold["stream_sha256"] == stream_sha256)


def aggregate_ranks(rank: int, world: int, bytes_timed: int, seconds: float, bit_exact, device=None):
    """Every rank's (input bytes of the timed steps, time) and its verdict against the reference's stream -> whole-job figures, the same on every rank:
    total bytes, the slowest rank's time (value = total / that), all(bit_exact) over the ranks that had a reference stream to compare with, and how many
    had.  One all_gather of two float64 and one all_gather_object (nccl = RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return float(bytes_timed), float(seconds), bit_exact, (1 if bit_exact is not None else 0)
    mine = torch.tensor([float(bytes_timed), float(seconds)], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    total_in = sum(float(v[0]) for v in allv)
    tmax = max(float(v[1]) for v in allv)
    flags = [None] * world
    dist.all_gather_object(flags, bit_exact)
    checked = [f for f in flags if f is not None]
    return total_in, tmax, (all(checked) if checked else None), len(checked)


def aggregate_blocks(rank: int, world: int, blocks: dict):
    """The per-GPU block-mode legs of bench.py -> one record on rank 0 (None elsewhere): bytes of all ranks over the slowest rank's time; any rank's
    error fails the record."""
    import torch.distributed as dist

    if world == 1:
        return blocks
    allb = [None] * world
    dist.all_gather_object(allb, blocks)
    if rank != 0:
        return None
    errs = [b["error"] for b in allb if "error" in b]
    agg = dict(allb[0])
    if errs:
        agg["error"] = "; ".join(errs)
        return agg
    agg["value"] = round(sum(b["bytes_timed"] for b in allb) / 1e6 / max(b["seconds"] for b in allb), 4)
    agg["per_gpu"] = [b["value"] for b in allb]
    agg["bytes_timed"] = sum(b["bytes_timed"] for b in allb)
    agg["workload"] = f"every one of the {world} GPUs: " + agg["workload"]
    return agg


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
