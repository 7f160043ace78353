"""CPU suite: the N>1 path (block partition + final gather) with world_size 2 over gloo.
The per-block compressor is a stand-in here (the oracle); on GPUs bench.py runs the HIP
path per rank and the same gather over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nlzm_amd import corpus, shard
from tests import oracle_py


def test_block_ranges_cover_input():
    for n in (0, 1, 7, 1000, 1_000_000_000):
        for k in (1, 2, 4, 8):
            r = [shard.block_range(n, k, i) for i in range(k)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(k - 1))
    assert shard.block_range(1_000_000_000, 8, 3) == (375_000_000, 500_000_000)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = corpus.syn_text(400_001)
    lo, hi = shard.block_range(data.size, world, rank)
    stream = oracle_py.compress(data[lo:hi], 18)
    local = torch.frombuffer(bytearray(stream), dtype=torch.uint8)
    dist.barrier()
    parts, lens = shard.gather_streams(local, rank, world)
    if rank == 0:
        q.put((shard.concat(parts), lens))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_round_trip():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    blob, lens = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = corpus.syn_text(400_001)
    streams = shard.split_streams(blob)
    assert [len(s) for s in streams] == lens and len(streams) == 2
    # parity oracle for a k-way run = the reference run separately on each block
    back = b"".join(oracle_py.decompress(s) for s in streams)
    assert back == data.tobytes()
    for i, s in enumerate(streams):
        lo, hi = shard.block_range(data.size, 2, i)
        assert s == oracle_py.compress(data[lo:hi], 18)
