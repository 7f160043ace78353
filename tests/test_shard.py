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


# ---- bench.py's N > 1 logic (nlzm_amd/shard.py: golden_for_rank, stream_matches, aggregate_ranks, aggregate_blocks) at world size 2 -------------------

def test_golden_lookup_per_rank():
    """bench.py --gpus N compares every rank's stream with the reference's run on that rank's block (tests/golden/gpus.json) and, at N = 1, the whole
    1e9-byte stream (full.json): the lookup must find exactly those records and nothing for other sizes."""
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for world in (2, 4, 8):
        for rank in range(world):
            lo, hi = shard.block_range(1_000_000_000, world, rank)
            g = shard.golden_for_rank(gd, world, rank, hi - lo)
            assert g is not None and (g["world"], g["rank"], g["size"]) == (world, rank, hi - lo) and len(g["stream_sha256"]) == 64
            assert shard.golden_for_rank(gd, world, rank, hi - lo - 1) is None
            assert shard.stream_matches(g, g["stream_size"], g["stream_sha256"]) is True
            assert shard.stream_matches(g, g["stream_size"] + 1, g["stream_sha256"]) is False
    g1 = shard.golden_for_rank(gd, 1, 0, 1_000_000_000)
    assert g1 is not None and g1["name"] == "text_1g_w28"
    assert shard.golden_for_rank(gd, 3, 0, 333_333_334) is None and shard.stream_matches(None, 1, "x") is None


def _bench_worker(rank, world, port, q, spoil):
    import hashlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = corpus.syn_text(300_001, corpus.SEED + 3)
    lo, hi = shard.block_range(data.size, world, rank)
    stream = oracle_py.compress(data[lo:hi], 18)                       # (the oracle stands in for the per-rank compressor)
    gold = {"size": hi - lo, "stream_size": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest()}
    got = stream if not (spoil and rank == 1) else stream[:-5] + b"\x01" + stream[-4:]
    verdict = shard.stream_matches(gold, len(got), hashlib.sha256(got).hexdigest())
    dist.barrier()
    parts, lens = shard.gather_streams(torch.frombuffer(bytearray(got), dtype=torch.uint8), rank, world)     # (inside bench.py's timed region)
    total_in, tmax, bit_exact, checked = shard.aggregate_ranks(rank, world, hi - lo, 1.0 + rank, verdict)
    blocks = {"value": 10.0 * (rank + 1), "bytes_timed": 1000 * (rank + 1), "seconds": 2.0 + rank, "workload": "w", "bit_exact": True}
    if spoil and rank == 1:
        blocks["error"] = "block 3 differs"
    agg = shard.aggregate_blocks(rank, world, blocks)
    q.put((rank, total_in, tmax, bit_exact, checked, agg, lens))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spoil", [False, True])
def test_bench_rank_aggregation_world2(spoil):
    """What `bench.py --gpus 2` does after the timed steps, with gloo and the oracle as the compressor: every rank's bytes and time -> total bytes over
    the slowest rank's time, every rank's verdict against its golden record -> one bit_exact (False as soon as ONE rank's stream differs), the per-GPU
    block legs -> one record on rank 0 (an error of any rank fails it)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q, spoil)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 300_001
    for rank, total_in, tmax, bit_exact, checked, agg, lens in res:
        assert total_in == float(n) and tmax == 2.0 and checked == 2
        assert bit_exact is (not spoil)
        assert len(lens) == 2 and sum(lens) > 0
        if rank == 0:
            if spoil:
                assert "block 3 differs" in agg["error"]
            else:
                assert agg["bytes_timed"] == 3000 and agg["value"] == round(3000 / 1e6 / 3.0, 4) and agg["per_gpu"] == [10.0, 20.0]
                assert agg["workload"].startswith("every one of the 2 GPUs")
        else:
            assert agg is None
