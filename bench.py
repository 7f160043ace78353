#!/usr/bin/env python3
"""bench.py -- compress MB/s of the MI355X NLZM path on the enwik9 configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-chunks B] [--full]

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): a 1,000,000,000-byte
stream at -window:28, synthetic stand-in for enwik9 (nlzm_amd/corpus.syn_text; the real file is used when
$NLZM_CORPUS_DIR holds it).  A *step* is one pass of the hot path over one batch of B consecutive chunks of
that stream (chunk = 122,368 input bytes = one frame, NLZM.cpp:1724): pre-pass kernels, the persistent
match-find/parse/emit launch, rANS-coding of the batch's frames and their gather into the output stream.
Steps are consecutive batches of the SAME stream (window, model and finder state carry over), so
--steps ceil(8173/B) --warmup 0 (or --full) compresses the whole file; the default K/W finish in minutes.
The input is resident in HBM before the timed region starts.

With N > 1 (one process per GPU, torchrun) the input is split into N independent blocks, one stream per
GPU; there is no data-path collective, only the final gather of the streams over RCCL,
which is inside the timed region.  value = input bytes all ranks consumed in the timed steps / max-rank time.

The JSON line also carries:
  roofline      dominant kernel (pipeline2_kernel) against the HBM roof: algorithmic bytes per launch from
                the device's own operation counters / the kernel's mean launch time measured
                with HIP events on the library's stream; peak_measured = a device copy on this GPU; b_min = the bytes
                that must move at least (input + output)
  cpu_baseline  the reference itself (oracle/_ref/nlzm_ref, built from /root/reference at build time) or, if
                that binary is absent, the oracle port, pinned to one core (taskset), on the bytes
                [0, (warmup + steps) * B * 122,368) of the same stream as a file of its own; gpu_same_bytes is
                the GPU's rate over exactly those bytes (warm-up launches included)
  --full        the whole stream; its SHA-256 is compared with the reference's (tests/golden/full.json)
  blocks        (N = 1) the same stream split into --block-streams independent blocks, all in flight on the one
                GPU: the path's only shard axis used inside a GPU.  A second measurement next to
                `value`, never part of it (the streams differ from the single-stream output).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

# Block mode runs more than one persistent launch at a time; the HIP runtime maps streams onto 4 hardware queues by
# default, which would serialise them (must be set before the runtime starts, i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nlzm_amd  # noqa: E402
from nlzm_amd import corpus, shard  # noqa: E402

STREAM_BYTES = 1_000_000_000
WINDOW = 28
CHUNK = 122_368
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s HBM3E


def algorithmic_bytes(st: dict) -> int:
    """bytes a bit-exact implementation must move, from the operation counters."""
    return (st["in_bytes_step"] + st["out_bytes_step"] + 16 * st["bt_calls"] + 12 * st["bt_tests"] + 2 * st["cmp_bytes"]
            + 8 * st["ht_rows"] + 4 * st["rk_probes"] + 4 * st["rk_inserts"])


def cpu_baseline(sample: np.ndarray) -> dict:
    """Time the reference's single-threaded CPU path on a bounded sample (rank 0, N=1 only), pinned to one core."""
    import shutil
    ref = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
    geo = nlzm_amd.geometry(int(sample.size), WINDOW)
    desc = (f"bytes [0, {sample.size}) of the workload as a file of its own, -window:{WINDOW} "
            f"(the reference shrinks it to {geo['hist_bits']} for a file of this size, NLZM.cpp:1716-1718)")
    pin = []
    if shutil.which("taskset"):
        cores = sorted(os.sched_getaffinity(0))
        pin = ["taskset", "-c", str(cores[len(cores) // 2])]
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
            sample.tofile(inp)
            t0 = time.perf_counter()
            subprocess.run(pin + [ref, f"-window:{WINDOW}", "c", inp, out], check=True, capture_output=True)
            dt = time.perf_counter() - t0
        kind = "reference"
    else:
        from tests import oracle_py       # the oracle is only the timed CPU leg here, never the product
        t0 = time.perf_counter()
        oracle_py.compress(sample, WINDOW)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": round(sample.size / 1e6 / dt, 4), "unit": "MB/s", "cores": 1, "kind": kind, "sample": desc,
            "seconds": round(dt, 2), "pinned": " ".join(pin) if pin else None, "host_cores_present": os.cpu_count()}


def copy_peak_gbs(torch, dev) -> float:
    """HBM bandwidth a plain device copy reaches on this GPU (read + write bytes / time), next to the nominal 8 TB/s."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9


def blocks_leg(lib, torch, dev, k: int, B: int, steps: int, warmup: int) -> dict:
    """The path's shard axis on ONE GPU: the stream split into k independent blocks, every block's
    NLZM stream in flight at once (one master CU + its worker CUs each, shared persistent launches).  Reported next
    to the headline value, never instead of it: the bytes differ from the single-stream output (each block equals
    the reference run on that block, tests/test_gpu_parity.py::test_blocks_in_flight_on_one_gpu)."""
    per = -(-STREAM_BYTES // k)
    need = min(per, (steps + warmup) * B * CHUNK + (1 << 20))
    d_in = torch.zeros(STREAM_BYTES + 4096, dtype=torch.uint8, device=dev)
    for i in range(k):      # only the prefix of each block is read by the timed rounds
        lo = min(STREAM_BYTES, i * per)
        m = min(need, STREAM_BYTES - lo)
        d_in[lo:lo + m].copy_(torch.from_numpy(corpus.syn_text(m, corpus.SEED + 100 + i)))
    torch.cuda.synchronize()
    rc = lib.nlzm_hip_blocks_begin(d_in.data_ptr(), STREAM_BYTES, k, WINDOW)
    if rc:
        return {"streams": k, "error": lib.nlzm_hip_last_error().decode()}
    done, fin, ms = C.c_uint64(0), C.c_int(0), C.c_double(0)
    try:
        for _ in range(warmup):
            if lib.nlzm_hip_blocks_step(B, C.byref(done), C.byref(fin), C.byref(ms)):
                return {"streams": k, "error": lib.nlzm_hip_last_error().decode()}
        d0 = done.value
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if lib.nlzm_hip_blocks_step(B, C.byref(done), C.byref(fin), C.byref(ms)):
                return {"streams": k, "error": lib.nlzm_hip_last_error().decode()}
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lib.nlzm_hip_blocks_abandon()
    geo = nlzm_amd.geometry(per, WINDOW)
    return {"streams": k, "value": round((done.value - d0) / 1e6 / dt, 4), "unit": "MB/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt / max(1, steps), 2), "bytes_timed": int(done.value - d0),
            "workload": f"{STREAM_BYTES} B stand-in split into {k} independent blocks of {per} B (-window:{WINDOW} auto-shrinks to "
                        f"{geo['hist_bits']}), all in flight on one GPU; step = {B} chunks of every block"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-chunks", type=int, default=8)
    ap.add_argument("--full", action="store_true", help="compress the whole stream (steps = all batches, warmup 0)")
    ap.add_argument("--cpu-sample-mb", type=float, default=30.0, help="upper bound of the CPU leg's sample (it covers the GPU leg's bytes up to this)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--block-streams", type=int, default=32,
                    help="N=1 only: also time the independent-block mode with this many streams in flight on the GPU (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nlzm_amd.init(local_rank)          # fails loudly without the HIP extension / a gfx950 device
    lib = nlzm_amd.load_library()
    B = args.batch_chunks
    nlzm_amd.set_option("batch_chunks", B)

    lo, hi = shard.block_range(STREAM_BYTES, world, rank)
    n = hi - lo
    nchunks = -(-n // CHUNK)
    nbatches = -(-nchunks // B)
    if args.full:
        args.steps, args.warmup = nbatches, 0
    steps_total = min(args.steps + args.warmup, nbatches)
    need = min(n, steps_total * B * CHUNK + (1 << 20))

    # ---- synthetic input of the stream's shape, resident in HBM before timing -----------------------
    real = os.path.join(os.environ.get("NLZM_CORPUS_DIR", "/nonexistent"), "enwik9")
    if os.path.exists(real):
        host = np.fromfile(real, dtype=np.uint8, count=need, offset=lo)
        data_kind = "file:enwik9"
    else:
        # N = 1: the prefix of the 1e9-byte stand-in; N > 1: every block is its own seeded text of the same kind
        # (blocks are independent streams anyway, and this avoids generating up to 875 MB per rank to skip over)
        host = corpus.syn_text(need, corpus.SEED + rank)
        data_kind = "synthetic"
    d_in = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)     # bytes past `need` are never read by the timed steps
    d_in[:need].copy_(torch.from_numpy(host[:need]))
    cap = int(lib.nlzm_hip_compress_bound(n))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    rc = lib.nlzm_hip_stream_begin(d_in.data_ptr(), n, WINDOW, d_out.data_ptr(), cap)
    if rc:
        raise SystemExit(f"stream_begin failed: {lib.nlzm_hip_last_error().decode()}")

    in_done, out_done, fin = C.c_uint64(0), C.c_uint64(0), C.c_int(0)

    def step():
        r = lib.nlzm_hip_stream_step(B, C.byref(in_done), C.byref(out_done), C.byref(fin))
        if r:
            raise SystemExit(f"stream_step failed: {lib.nlzm_hip_last_error().decode()}")

    torch.cuda.synchronize()
    t_w0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t_w0
    st0, tm0 = nlzm_amd.stats(), nlzm_amd.timing()
    in0, out0 = in_done.value, out_done.value

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done_steps = 0
    for _ in range(args.steps):
        if fin.value:
            break
        step()
        done_steps += 1
    if world > 1:
        # the only collective of the path: gather what the ranks produced
        produced = d_out[out0:out_done.value]
        shard.gather_streams(produced, rank, world, dev)
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    st1, tm1 = nlzm_amd.stats(), nlzm_amd.timing()
    mine = torch.tensor([in_done.value - in0, dt], dtype=torch.float64, device=dev)
    if world > 1:
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        total_in = sum(float(v[0]) for v in allv)
        tmax = max(float(v[1]) for v in allv)
    else:
        total_in, tmax = float(mine[0]), dt

    if rank == 0:
        d = {k: st1[k] - st0[k] for k in ("bt_calls", "bt_tests", "cmp_bytes", "ht_rows", "rk_probes", "rk_inserts",
                                          "positions", "uncertain_positions")}
        d["in_bytes_step"] = in_done.value - in0
        d["out_bytes_step"] = out_done.value - out0
        launches = max(1, tm1["match_parse_launches"] - tm0["match_parse_launches"])
        k_ms = (tm1["match_parse_ms"] - tm0["match_parse_ms"]) / launches
        b_alg = algorithmic_bytes(d) / launches
        achieved = b_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        res = {
            "metric": "compress MB/s", "value": round(total_in / 1e6 / tmax, 4), "unit": "MB/s", "n_gpus": world,
            "steps": done_steps, "warmup": args.warmup, "ms_per_step": round(1e3 * tmax / max(1, done_steps), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": data_kind,
            "config": {
                "workload": f"enwik9 stand-in: {STREAM_BYTES} B stream, -window:{WINDOW}"
                            + (f" split into {world} independent blocks of {n} B (RCCL world size {dist.get_world_size()}, window "
                               f"{nlzm_amd.geometry(n, WINDOW)['hist_bits']} after the reference's auto-shrink)" if world > 1 else "")
                            + f"; step = {B} chunks ({B * CHUNK} B) of the stream, timed steps = batches {args.warmup}.."
                              f"{args.warmup + done_steps - 1} of {nbatches}",
                "window_bits": WINDOW, "batch_chunks": B, "bytes_timed": int(total_in),
                "bit_exact_with_reference": "checked by tests/test_gpu_parity.py (same code path)",
            },
            "roofline": {"bound": "hbm", "kernel": "pipeline2_kernel", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": None,
                         "peak_measured": round(copy_peak_gbs(torch, dev), 1),
                         "algorithmic_bytes_per_launch": int(b_alg), "kernel_ms_per_launch": round(k_ms, 3),
                         "b_min_bytes_per_launch": int((d["in_bytes_step"] + d["out_bytes_step"]) / launches),
                         "algorithmic_bytes_per_input_byte": round(algorithmic_bytes(d) / max(1, d["in_bytes_step"]), 2)},
            "kernel_ms": {"prep": round(tm1["prep_ms"] - tm0["prep_ms"], 3),
                          "match_parse": round(tm1["match_parse_ms"] - tm0["match_parse_ms"], 3),
                          "rans_gather": round(tm1["rans_ms"] - tm0["rans_ms"], 3)},
            "counters": d,
        }
        # HBM traffic of the same command, measured in separate rocprofv3 --pmc passes (tests/prof_run.sh) and
        # committed under profiles/: bench.py itself cannot read PMC counters
        try:
            import glob
            prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]
            pj = json.load(open(prof))
            if pj.get("kernel") == "pipeline2_kernel" and pj["bench_line_under_profiler"]["config"]["batch_chunks"] == B and world == 1:
                res["roofline"]["traffic"] = int(pj["traffic_bytes_per_launch"]["total_with_fetch_x2"])
                res["roofline"]["traffic_source"] = os.path.relpath(prof, ROOT)
        except Exception:
            pass
        if world == 1 and not args.no_cpu and not args.full:
            # the CPU leg covers the bytes the GPU consumed from offset 0 (warm-up launches included), bounded
            sample_n = int(min(in_done.value, args.cpu_sample_mb * 1e6))
            res["cpu_baseline"] = cpu_baseline(host[:sample_n])
            res["cpu_baseline"]["gpu_same_bytes"] = {"value": round(in_done.value / 1e6 / (t_warm + dt), 4), "unit": "MB/s",
                                                     "bytes": int(in_done.value), "note": "GPU rate over bytes [0, bytes), warm-up launches included"}
        else:
            res["cpu_baseline"] = None
        if args.full and world == 1:
            # the whole stream: its bytes must be the reference's (tests/golden/full.json, generated from the reference build)
            import hashlib
            dst_len = C.c_uint64(0)
            if lib.nlzm_hip_stream_finish(C.byref(dst_len)):
                raise SystemExit(f"stream_finish failed: {lib.nlzm_hip_last_error().decode()}")
            got = d_out[:dst_len.value].cpu().numpy().tobytes()
            sha = hashlib.sha256(got).hexdigest()
            res["stream_bytes"], res["stream_sha256"] = int(dst_len.value), sha
            gold = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "full.json")))["cases"]}.get("text_1g_w28")
            if data_kind == "synthetic" and gold:
                res["bit_exact"] = bool(gold["stream_size"] == dst_len.value and gold["stream_sha256"] == sha)
                if not res["bit_exact"]:
                    print(json.dumps(res))
                    raise SystemExit("bench --full: the stream differs from the reference's")
        if world == 1 and args.block_streams > 0 and not args.full:
            del d_in, d_out
            torch.cuda.empty_cache()
            res["blocks"] = blocks_leg(lib, torch, dev, args.block_streams, B, args.steps, args.warmup)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
