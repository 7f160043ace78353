#!/usr/bin/env python3
"""bench.py -- compress MB/s of the MI355X NLZM path on the enwik9 configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-chunks B]

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): the WHOLE 1,000,000,000-byte
stream at -window:28, synthetic stand-in for enwik9 (nlzm_amd/corpus.syn_text; the real file is used when
$NLZM_CORPUS_DIR holds it).  The stream is one serial job (window, model and finder state carry through the file),
so a *step* is a fixed slice of it: 1/25 of its persistent launches (41 launches of B = 8 chunks, 40 MB of input;
chunk = 122,368 input bytes = one frame, NLZM.cpp:1724), each launch = pre-pass kernels, the persistent
match-find/parse/emit kernel, rANS-coding of the launch's frames and their gather into the output stream.
`--steps 20 --warmup 5` (what the driver runs) therefore covers the stream from its first byte to its last: the
warm-up steps are the first 200 MB, the timed steps the other 800 MB, and when the last step has run the stream is
finished, hashed and compared with the REFERENCE's own stream for this input (tests/golden/full.json, made by
oracle/make_golden_full.py from the compiled reference): `config.bit_exact`.  Fewer steps time a prefix of the same
stream and say so (`bit_exact` null: there is nothing to compare a prefix with).  The input is resident in HBM
before the timed region starts.

With N > 1 (one process per GPU, torchrun) the input is N independent blocks of 1e9/N bytes, one stream per GPU
(SURVEY.md 8e); there is no data-path collective, only the final gather of the streams over RCCL, which is inside
the timed region.  value = input bytes all ranks consumed in the timed steps / max-rank time.

The JSON line also carries:
  roofline      dominant kernel (pipeline2_kernel) against the HBM roof: algorithmic bytes per launch from
                the device's own operation counters / the kernel's mean launch time measured
                with HIP events on the library's stream; peak_measured = a device copy on this GPU; b_min = the bytes
                that must move at least (input + output)
  cpu_baseline  the reference itself (oracle/_ref/nlzm_ref, built from /root/reference at build time), pinned to one core
                (taskset), started in the background BEFORE the GPU legs and collected after them: `value` on the first
                280 MB of the same stream at -window:28 (a file of more than 2^28 bytes keeps the window at 28: the depth the
                headline is quoted on), `shallow` on the first 25 MB as a file of its own (window 25, which favours the
                reference).  If that binary is absent: the oracle port on the small sample
  blocks        the path's only shard axis used INSIDE each GPU: this rank's input split into
                --block-streams independent blocks, all in flight at once.  A second measurement next to `value`,
                never part of it (k streams instead of one).  At N = 1 every block's stream is compared with the
                reference run on that block (tests/golden/blocks_1g.json).
  workloads     the same path on the inputs the stand-in flatters it on: 120 MB of real text (source code) at -window:28 and 100 MB of wiki-shaped
                markup at -window:26, each a whole stream timed from stream_begin to the finished stream and compared with the REFERENCE's own stream
                (tests/golden/workloads.json, which also holds the reference's wall time: cpu_reference).  --no-workloads skips them.
Any leg that fails puts an "error" key at the top level of the line and the exit code is 1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
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
STEPS_PER_STREAM = 25      # a step is 1/25 of the stream's launches: --steps 20 --warmup 5 is the whole stream
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s HBM3E
BLOCKS_BATCH = 8           # block-mode leg: chunks of every block per shared persistent launch


def algorithmic_bytes(st: dict) -> int:
    """bytes a bit-exact implementation must move, from the operation counters."""
    return (st["in_bytes_step"] + st["out_bytes_step"] + 16 * st["bt_calls"] + 12 * st["bt_tests"] + 2 * st["cmp_bytes"]
            + 8 * st["ht_rows"] + 4 * st["rk_probes"] + 4 * st["rk_inserts"])


def algorithmic_bytes_lines(st: dict) -> int:
    """The same operations at the granularity memory moves them: a 64-byte line per random access -- per BT4 test the pair read,
    the compare's first line and the link store's line (written back whole); per call the head's line read and written and the
    two terminal links; an HT row read and written; an RK256 probe / insert; byte compares and the streams as they are."""
    return (st["in_bytes_step"] + st["out_bytes_step"] + 256 * st["bt_calls"] + 192 * st["bt_tests"] + 2 * st["cmp_bytes"]
            + 128 * st["ht_rows"] + 64 * st["rk_probes"] + 64 * st["rk_inserts"])


STAGE_COUNTERS = ("positions", "finder_total_cycles", "finder_wait_cycles", "finder_bt_wait_cycles", "table_total_cycles", "table_wait_cycles",
                  "parser_total_cycles", "parser_wait_cycles", "parser_pass_cycles", "parser_setup_cycles", "parser_emit_cycles", "parser_passes",
                  "parser_blocks", "helper_jobs", "helper_taken", "helper_taken_nodes", "helper_wait_cycles", "worker_call_cycles",
                  "worker_call_tests", "worker_calls", "hot_bin_calls")
# machine constants measured in rounds 3 - 5, cycles
LAT_BARRIER, LAT_LDS, LAT_ISSUE, LAT_L2, LAT_HBM = 250, 130, 4.5, 700, 1500


def stage_counters() -> dict:
    return {k: nlzm_amd.counter(k) for k in STAGE_COUNTERS}


def latency_bound(c0: dict, c1: dict) -> dict:
    """The bound that applies to this path (the HBM roof never will): the three pipelined stages of the persistent kernel are chains of
    dependent round trips, and the slowest one sets the rate.  Per stage the busy cycles per position (total - waiting) over the timed
    launches, from the kernel's own counters; for the parser stage the floor of its pass loop from the machine constants of DESIGN.md
    section 9 -- a pass cannot take less than one workgroup barrier, one dependent LDS atomic and one dependent LDS read, the
    four DPP scans of the update (6 steps x 2 dependent instructions each) and one ds_bpermute round trip; for BT4 the measured lane
    cycles per test against one L2 / HBM round trip."""
    d = {k: c1[k] - c0[k] for k in STAGE_COUNTERS}
    n = max(1, d["positions"])
    per = lambda k: d[k] / n
    blocks, passes = max(1, d["parser_blocks"]), max(1, d["parser_passes"])
    own_nodes = n - d["helper_taken_nodes"]                  # nodes the parser stage computed itself (the rest: taken over from the helper)
    pass_floor = LAT_BARRIER + 2 * LAT_LDS + 4 * 6 * 2 * LAT_ISSUE + LAT_LDS
    stages = {
        "finder": {"total": round(per("finder_total_cycles"), 1), "busy": round(per("finder_total_cycles") - per("finder_wait_cycles"), 1),
                   "waiting_for_bt4_results": round(per("finder_bt_wait_cycles"), 1)},
        "table": {"total": round(per("table_total_cycles"), 1), "busy": round(per("table_total_cycles") - per("table_wait_cycles"), 1)},
        "parser": {"total": round(per("parser_total_cycles"), 1),
                   "busy": round(per("parser_total_cycles") - per("parser_wait_cycles") - per("helper_wait_cycles"), 1),
                   "waiting_for_records": round(per("parser_wait_cycles"), 1), "waiting_for_helper": round(per("helper_wait_cycles"), 1),
                   "passes": round(per("parser_pass_cycles"), 1), "block_setup": round(per("parser_setup_cycles"), 1), "emission": round(per("parser_emit_cycles"), 1),
                   "passes_per_block": round(passes / blocks, 2), "nodes_per_block": round(own_nodes / blocks, 1), "cycles_per_pass": round(d["parser_pass_cycles"] / passes),
                   "helper": {"jobs": d["helper_jobs"], "taken_over": d["helper_taken"], "share_of_positions": round(d["helper_taken_nodes"] / n, 4)}},
    }
    # what a stage cannot wait away: its busy cycles; for the finder also the BT4 results it stands in front of
    load = {"finder": stages["finder"]["busy"] + stages["finder"]["waiting_for_bt4_results"], "table": stages["table"]["busy"], "parser": stages["parser"]["busy"]}
    slowest = max(load, key=load.get)
    achieved = max(stages[s]["total"] for s in stages)
    bound = passes * pass_floor / n                          # the parser stage's pass loop at its floor, per position of the stream
    tests = max(1, d["worker_call_tests"])
    return {"unit": "cycles per position", "achieved": round(achieved, 1), "slowest_stage": slowest, "load": {k: round(v, 1) for k, v in load.items()},
            "stages": stages,
            "bound": round(bound, 1), "achieved_over_bound": round(achieved / bound, 2) if bound else None,
            "bound_model": f"parser passes x ({LAT_BARRIER} barrier + 3 x {LAT_LDS} LDS / bpermute round trips + 48 dependent scan instructions x {LAT_ISSUE}) "
                           f"= {pass_floor:.0f} cycles per pass, {passes / blocks:.2f} passes per block of {own_nodes / blocks:.1f} nodes",
            "bt4": {"lane_cycles_per_test": round(d["worker_call_cycles"] / tests), "l2_round_trip": LAT_L2, "hbm_round_trip": LAT_HBM,
                    "calls_by_hot_bin_waves": round(d["hot_bin_calls"] / max(1, d["worker_calls"]), 4)}}


def csrc_sha16() -> str:
    """the KERNEL sources (headers and .hip: what the counters of a launch depend on; the host file only queues launches) --
    tests/prof_summarise.py records the same over the profiled tree"""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nlzm_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "nlzm_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class CpuBaseline:
    """The reference's single-threaded CPU path (oracle/_ref/nlzm_ref, built from /root/reference at build time), pinned to
    one core each, started BEFORE the GPU legs and collected after them (the host idles while the GPU runs, so it costs
    no wall time; rank 0, N = 1 only):
      * the headline configuration's depth: the first `deep_bytes` (> 2^28, so the window stays at 28, NLZM.cpp:1716-1718)
        of the workload as a file, -window:28 -- `value`;
      * the first 25 MB as a file of its own (the window shrinks to 25 there, which favours it) -- `shallow`.
    Without the reference binary the oracle port is timed on the small sample only, in the foreground (kind "port")."""

    def __init__(self, host: np.ndarray, deep_bytes: int, shallow_bytes: int):
        import shutil
        self.ref = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
        self.host, self.deep_n, self.shallow_n = host, int(min(host.size, deep_bytes)), int(min(host.size, shallow_bytes))
        self.tmp, self.jobs = None, []
        cores = sorted(os.sched_getaffinity(0))
        self.pins = [cores[len(cores) // 2], cores[len(cores) // 2 + 1 if len(cores) > 2 else 0]]
        self.taskset = shutil.which("taskset")
        # (this process, the HIP runtime's threads and the block leg's host threads stay off the two cores the reference runs on)
        try:
            rest = set(cores) - set(self.pins)
            if len(rest) >= 2 and os.path.exists(self.ref):
                os.sched_setaffinity(0, rest)
        except OSError:
            pass
        if os.path.exists(self.ref):
            self.tmp = tempfile.TemporaryDirectory()
            for k, n in enumerate((self.deep_n, self.shallow_n)):
                inp, out = os.path.join(self.tmp.name, f"in{k}.bin"), os.path.join(self.tmp.name, f"out{k}.nlzm")
                host[:n].tofile(inp)
                pin = [self.taskset, "-c", str(self.pins[k])] if self.taskset else []
                # (wall time of the process itself, taken by a shell around it: the parent is busy with the GPU legs)
                cmd = pin + ["bash", "-c", f's=$(date +%s.%N); "{self.ref}" -window:{WINDOW} c "{inp}" "{out}" > /dev/null; e=$(date +%s.%N); echo "$s $e"']
                self.jobs.append((n, pin, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))

    def collect(self) -> dict:
        base = {"unit": "MB/s", "cores": 1, "host_cores_present": os.cpu_count()}
        if not self.jobs:
            from tests import oracle_py       # the oracle is only the timed CPU leg here, never the product
            sample = self.host[:self.shallow_n]
            t0 = time.perf_counter()
            oracle_py.compress(sample, WINDOW)
            dt = time.perf_counter() - t0
            return dict(base, value=round(sample.size / 1e6 / dt, 4), kind="port", seconds=round(dt, 2), pinned=None,
                        sample=f"bytes [0, {sample.size}) of the workload as a file of its own, -window:{WINDOW} (oracle port: the reference binary is absent)")
        res = []
        for n, pin, pr in self.jobs:
            out, _ = pr.communicate()
            s, e = (float(x) for x in out.split())
            geo = nlzm_amd.geometry(n, WINDOW)
            res.append({"bytes": n, "seconds": round(e - s, 2), "value": round(n / 1e6 / (e - s), 4), "window_after_shrink": geo["hist_bits"],
                        "pinned": " ".join(pin[1:]) if pin else None})
        self.tmp.cleanup()
        deep, shallow = res
        return dict(base, value=deep["value"], kind="reference", seconds=deep["seconds"], pinned=deep["pinned"],
                    sample=f"bytes [0, {deep['bytes']}) of the workload as a file, -window:{WINDOW} (a file of more than 2^28 bytes keeps the window at "
                           f"{deep['window_after_shrink']}: the depth the headline is quoted on), run in the background of the GPU legs on a core of its own; "
                           f"over the whole 1e9-byte stream it ran at 0.77 MB/s on a host core of an MI355X box (profiles/r02_ref_full_1e9.json)",
                    shallow={"value": shallow["value"], "seconds": shallow["seconds"], "pinned": shallow["pinned"],
                             "sample": f"bytes [0, {shallow['bytes']}) as a file of its own: the window shrinks to {shallow['window_after_shrink']}, which favours the reference"})


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


def blocks_leg(lib, torch, dev, d_in, n: int, k: int, B: int, steps: int, warmup: int, check: bool) -> dict:
    """The path's shard axis inside ONE GPU: this rank's n input bytes split into k independent blocks,
    every block's NLZM stream in flight at once (three stage CUs + worker CUs each, shared persistent launches).  Steps
    as in the single-stream leg: 1/25 of a block's launches each.  Reported next to the headline value, never instead
    of it: the bytes differ from the single-stream output -- each block's stream equals the reference run on that
    block alone, which `check` verifies against tests/golden/blocks_1g.json when all the steps were run."""
    per = -(-n // k)
    B = BLOCKS_BATCH                    # (chunks of every block per shared launch.  The launches of consecutive rounds are queued back
    nlzm_amd.set_option("block_batch_chunks", B)    #  to back, so their length matters little: 6 / 8 / 12 / 16 chunks 88.9 / 89.8 / 88.7 / 89.6 MB/s, profiles/r04_block_mode.txt)
    nb_launch = -(-(-(-per // CHUNK)) // B)
    per_step = max(1, -(-nb_launch // STEPS_PER_STREAM)) * B
    geo = nlzm_amd.geometry(per, WINDOW)
    out = {"streams": k,
           "workload": f"{n} B split into {k} independent blocks of {per} B (-window:{WINDOW} auto-shrinks to {geo['hist_bits']}), "
                       f"all in flight on one GPU; step = {per_step} chunks of every block"}
    t_b0 = time.perf_counter()
    if lib.nlzm_hip_blocks_begin(d_in.data_ptr(), n, k, WINDOW):
        return dict(out, error=lib.nlzm_hip_last_error().decode())
    out["begin_s"] = round(time.perf_counter() - t_b0, 2)
    try:
        out["pool_bytes"] = nlzm_amd.counter("block_pool_bytes")       # the set's ONE device allocation (kept for the next set)
    except nlzm_amd.NlzmError:
        pass
    done, fin, ms = C.c_uint64(0), C.c_int(0), C.c_double(0)
    for _ in range(warmup):
        if lib.nlzm_hip_blocks_step(per_step, C.byref(done), C.byref(fin), C.byref(ms)):
            return dict(out, error=lib.nlzm_hip_last_error().decode())      # (a failed step has closed the set)
    d0 = done.value
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ran = 0
    for _ in range(steps):
        if fin.value:
            break
        if lib.nlzm_hip_blocks_step(per_step, C.byref(done), C.byref(fin), C.byref(ms)):
            return dict(out, error=lib.nlzm_hip_last_error().decode())
        ran += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update({"value": round((done.value - d0) / 1e6 / dt, 4), "per_gpu": [round((done.value - d0) / 1e6 / dt, 4)], "unit": "MB/s", "steps": ran, "warmup": warmup,
                "ms_per_step": round(1e3 * dt / max(1, ran), 2), "bytes_timed": int(done.value - d0), "seconds": round(dt, 3)})
    if not fin.value:
        lib.nlzm_hip_blocks_abandon()
        out["bit_exact"] = None
        return out
    cap = int(lib.nlzm_hip_compress_bound(n)) + k * (16 + 131072)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    lens = (C.c_uint64 * k)()
    total = C.c_uint64(0)
    t_f0 = time.perf_counter()
    if lib.nlzm_hip_blocks_finish(d_out.data_ptr(), cap, lens, C.byref(total)):
        return dict(out, error=lib.nlzm_hip_last_error().decode())
    out["finish_s"] = round(time.perf_counter() - t_f0, 2)
    out["stream_bytes"] = int(total.value)
    # everything from opening the set to the gathered streams: begin (the first set of a process pays one large hipMalloc), the warm-up
    # steps, the timed steps, finish -- `value` above is the timed steps alone
    wall = time.perf_counter() - t_b0
    out["wall_s"] = round(wall, 2)
    out["wall_value"] = round(n / 1e6 / wall, 4)
    gold_path = os.path.join(ROOT, "tests", "golden", "blocks_1g.json")
    out["bit_exact"] = None
    if check and os.path.exists(gold_path):
        gold = json.load(open(gold_path))
        if gold["nblocks"] == k and gold["size"] == n and gold["window"] == WINDOW:
            host = d_out[: total.value].cpu().numpy()
            pos, bad = 0, []
            for i in range(k):
                s = host[pos: pos + int(lens[i])].tobytes()
                pos += int(lens[i])
                if (len(s), hashlib.sha256(s).hexdigest()) != (gold["blocks"][i]["stream_size"], gold["blocks"][i]["stream_sha256"]):
                    bad.append(i)
            out["bit_exact"] = not bad
            if bad:
                out["error"] = f"blocks {bad} differ from the reference's streams"
    return out


def workloads_leg(lib, torch, dev) -> dict:
    """The headline's stand-in is prose-like and flatters the pipeline: source code (one BT4 head -- runs of spaces -- holds 17 % of the positions) and wiki-shaped
    markup run several times slower.  Each workload below is a whole stream, input resident in HBM, timed from stream_begin to the
    finished stream, compared with the REFERENCE's stream for the same bytes (tests/golden/workloads.json, made by `oracle/make_golden_real.py bench`);
    `cpu_reference` is the reference's own wall time recorded there (one core of the build container).  real_text is the image's own source files: a box
    whose files differ gets a loud null instead of a number."""
    gold_path = os.path.join(ROOT, "tests", "golden", "workloads.json")
    if not os.path.exists(gold_path):
        return {"error": "tests/golden/workloads.json is missing"}
    out = {}
    for gold in json.load(open(gold_path))["cases"]:
        name, rec = gold["name"], {"kind": gold["kind"], "bytes": gold["size"], "window_bits": gold["window"], "unit": "MB/s"}
        out[name] = rec
        try:
            host = corpus.make(gold["kind"], gold["size"])
        except RuntimeError as e:                       # (real_text: the source trees are not there)
            rec.update(value=None, bit_exact=None, note=f"input not available on this box: {e}")
            continue
        if hashlib.sha256(host.tobytes()).hexdigest() != gold["input_sha256"]:
            rec.update(value=None, bit_exact=None, note="this box's files give other input bytes than the golden record was made from: not run")
            continue
        n = int(host.size)
        d_in = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)
        d_in[:n].copy_(torch.from_numpy(host))
        cap = int(lib.nlzm_hip_compress_bound(n))
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        dst_len = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = lib.nlzm_hip_compress_dev(d_in.data_ptr(), n, gold["window"], d_out.data_ptr(), cap, C.byref(dst_len))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rc:
            rec.update(value=None, bit_exact=None, error=lib.nlzm_hip_last_error().decode())
            continue
        sha = hashlib.sha256(d_out[:dst_len.value].cpu().numpy().tobytes()).hexdigest()
        lb = latency_bound({k: 0 for k in STAGE_COUNTERS}, stage_counters())       # (the counters are the finished stream's own)
        rec.update(value=round(n / 1e6 / dt, 4), seconds=round(dt, 2), stream_bytes=int(dst_len.value),
                   bit_exact=bool(dst_len.value == gold["stream_size"] and sha == gold["stream_sha256"]),
                   bit_exact_against="tests/golden/workloads.json (the reference's own stream for these bytes)",
                   cycles_per_position={"achieved": lb["achieved"], "finder_busy": lb["stages"]["finder"]["busy"], "finder_waiting_for_bt4_results": lb["stages"]["finder"]["waiting_for_bt4_results"],
                                        "table_busy": lb["stages"]["table"]["busy"], "parser_busy": lb["stages"]["parser"]["busy"]},
                   cpu_reference={"value": gold["reference_mb_per_s"], "seconds": gold["reference_seconds"], "cores": 1,
                                  "where": "the reference on one core of the build container, recorded with the golden stream"},
                   vs_cpu_reference=round(n / 1e6 / dt / gold["reference_mb_per_s"], 2))
        del d_in, d_out
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-chunks", type=int, default=8)
    ap.add_argument("--cpu-sample-mb", type=float, default=25.0, help="size of the CPU leg's small sample (a prefix of the stream as a file of its own)")
    ap.add_argument("--cpu-deep-mb", type=float, default=280.0, help="size of the CPU leg's sample at the headline's depth (> 2^28 bytes keeps -window:28)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the real-text and markup workloads that follow the headline (about 3 minutes)")
    ap.add_argument("--block-streams", type=int, default=32,
                    help="also time the independent-block mode with this many streams in flight on each GPU (0: skip)")
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
    nlaunch = -(-nchunks // B)
    per_step = -(-nlaunch // STEPS_PER_STREAM)           # launches per step
    steps_total = args.steps + args.warmup
    whole = steps_total * per_step >= nlaunch
    need = n if whole or args.block_streams > 0 else min(n, steps_total * per_step * B * CHUNK + (1 << 20))

    # ---- synthetic input of the stream's shape, resident in HBM before timing -----------------------
    real = os.path.join(os.environ.get("NLZM_CORPUS_DIR", "/nonexistent"), "enwik9")
    if os.path.exists(real):
        host = np.fromfile(real, dtype=np.uint8, count=need, offset=lo)
        data_kind = "file:enwik9"
    else:
        # N = 1: the 1e9-byte stand-in; N > 1: every block is its own seeded text of the same kind (blocks are
        # independent streams anyway, and this avoids generating up to 875 MB per rank to skip over)
        host = corpus.syn_text(need, corpus.SEED + rank)
        data_kind = "synthetic"
    cpu_leg = None
    if world == 1 and not args.no_cpu:
        # (needs the bytes on the host: a run of a few steps generates a prefix only, and the CPU leg takes what there is)
        cpu_leg = CpuBaseline(host, int(args.cpu_deep_mb * 1e6), int(args.cpu_sample_mb * 1e6))
    d_in = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)     # bytes past `need` are never read by the timed steps
    d_in[:need].copy_(torch.from_numpy(host[:need]))
    cap = int(lib.nlzm_hip_compress_bound(n))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    errors = []
    rc = lib.nlzm_hip_stream_begin(d_in.data_ptr(), n, WINDOW, d_out.data_ptr(), cap)
    if rc:
        raise SystemExit(f"stream_begin failed: {lib.nlzm_hip_last_error().decode()}")

    in_done, out_done, fin = C.c_uint64(0), C.c_uint64(0), C.c_int(0)

    def step():
        r = lib.nlzm_hip_stream_step(per_step * B, C.byref(in_done), C.byref(out_done), C.byref(fin))
        if r:
            raise SystemExit(f"stream_step failed: {lib.nlzm_hip_last_error().decode()}")

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        if not fin.value:
            step()
    torch.cuda.synchronize()
    st0, tm0 = nlzm_amd.stats(), nlzm_amd.timing()
    sc0 = stage_counters()
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
    sc1 = stage_counters()

    # ---- the stream is complete: the reference's bytes? ------------------------------------------------------------
    bit_exact, stream_len, stream_sha = None, None, None
    if fin.value:
        dst_len = C.c_uint64(0)
        if lib.nlzm_hip_stream_finish(C.byref(dst_len)):
            raise SystemExit(f"stream_finish failed: {lib.nlzm_hip_last_error().decode()}")
        stream_len = int(dst_len.value)
        stream_sha = hashlib.sha256(d_out[:stream_len].cpu().numpy().tobytes()).hexdigest()
        # N = 1: the 1e9-byte stand-in (tests/golden/full.json); N > 1: every rank's block has the reference's own stream in
        # tests/golden/gpus.json (oracle/make_golden_gpus.py: one reference run per block, NLZM.cpp:1711 with the auto-shrink of :1716-1718)
        if data_kind == "synthetic":
            bit_exact = shard.stream_matches(shard.golden_for_rank(os.path.join(ROOT, "tests", "golden"), world, rank, n), stream_len, stream_sha)
    # (the ranks' bytes and times, and their verdicts: nlzm_amd/shard.py -- tests/test_shard.py drives the same calls at world size 2 over gloo)
    total_in, tmax, bit_exact, ranks_checked = shard.aggregate_ranks(rank, world, in_done.value - in0, dt, bit_exact, dev)

    # ---- block mode inside each GPU ----------------------------------------------------------------------------------
    blocks = None
    if args.block_streams > 0:
        del d_out
        torch.cuda.empty_cache()
        blocks = blocks_leg(lib, torch, dev, d_in, n, args.block_streams, B, args.steps, args.warmup,
                            check=(world == 1 and data_kind == "synthetic"))
        blocks = shard.aggregate_blocks(rank, world, blocks)

    # ---- the inputs the stand-in flatters the pipeline on: real text (source code) and wiki-shaped markup, whole streams ------------------------------
    workloads = None
    if world == 1 and not args.no_workloads:
        try:
            del d_out
        except NameError:
            pass
        del d_in
        torch.cuda.empty_cache()
        workloads = workloads_leg(lib, torch, dev)

    if rank == 0:
        d = {k: st1[k] - st0[k] for k in ("bt_calls", "bt_tests", "cmp_bytes", "ht_rows", "rk_probes", "rk_inserts",
                                          "positions", "uncertain_positions")}
        d["in_bytes_step"] = in_done.value - in0
        d["out_bytes_step"] = out_done.value - out0
        launches = max(1, tm1["match_parse_launches"] - tm0["match_parse_launches"])
        k_ms = (tm1["match_parse_ms"] - tm0["match_parse_ms"]) / launches
        b_alg = algorithmic_bytes(d) / launches
        achieved = b_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        first_b, last_b = in0, in_done.value
        res = {
            "metric": "compress MB/s", "value": round(total_in / 1e6 / tmax, 4), "unit": "MB/s", "n_gpus": world,
            "steps": done_steps, "warmup": args.warmup, "ms_per_step": round(1e3 * tmax / max(1, done_steps), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": data_kind,
            "config": {
                "workload": f"enwik9 stand-in: {STREAM_BYTES} B stream, -window:{WINDOW}"
                            + (f" split into {world} independent blocks of {n} B, one per GPU (RCCL world size {dist.get_world_size()}, window "
                               f"{nlzm_amd.geometry(n, WINDOW)['hist_bits']} after the reference's auto-shrink; rank r's block is its own seeded text "
                               f"syn_text({n}, SEED + r) -- the same kind and amount of work as byte range r of the 1e9-byte stand-in, not those bytes; "
                               f"its reference stream is in tests/golden/gpus.json)" if world > 1 else "")
                            + f"; step = {per_step} launches of {B} chunks ({per_step * B * CHUNK} B) of the stream; timed: bytes "
                              f"[{first_b}, {last_b}) of {n}" + (", i.e. to the end of the stream" if fin.value else " (a prefix: fewer than 25 steps were asked for)"),
                "window_bits": WINDOW, "batch_chunks": B, "launches_per_step": per_step, "bytes_timed": int(total_in),
                "whole_stream": bool(fin.value), "bit_exact": bit_exact,
                "ranks_checked": ranks_checked,
                "bit_exact_against": (("tests/golden/full.json" if world == 1 else "tests/golden/gpus.json") + " (the reference's own stream for this input" + (", every rank's block)" if world > 1 else ")") if bit_exact is not None
                                      else "nothing: " + ("a prefix has no reference stream; run --steps 20 --warmup 5" if not fin.value else "no reference stream for this input")),
                "stream_bytes": stream_len, "stream_sha256": stream_sha,
            },
            "roofline": {"bound": "hbm", "kernel": "pipeline2_kernel", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": None,
                         "peak_measured": round(copy_peak_gbs(torch, dev), 1),
                         "algorithmic_bytes_per_launch": int(b_alg), "kernel_ms_per_launch": round(k_ms, 3),
                         "launches_timed": int(launches),
                         "b_min_bytes_per_launch": int((d["in_bytes_step"] + d["out_bytes_step"]) / launches),
                         "algorithmic_bytes_per_input_byte": round(algorithmic_bytes(d) / max(1, d["in_bytes_step"]), 2),
                         "algorithmic_bytes_line_granular": int(algorithmic_bytes_lines(d) / launches),
                         "frac_line_granular": round(algorithmic_bytes_lines(d) / launches / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 8) if k_ms > 0 else 0.0,
                         "latency_bound": latency_bound(sc0, sc1),
                         # (SURVEY 8d defines cmp_bytes as the iterations of the reference's compare loop, NLZM.cpp:863; the device counts the oracle's
                         #  cmp_bytes_needed -- an explicit rep probe measured up to the 264 bytes that can matter, the reference compares up to 4,096 and then
                         #  caps the length, :1605-1606.  The oracle counts both: 1,039,103,767 against 1,087,452,480 on the first 16 MB of this workload at
                         #  -window:28, identical on markup; 2 x cmp_bytes is 28 % of the algorithmic bytes)
                         "cmp_bytes_definition": {"counted": "oracle cmp_bytes_needed (rep probes up to 264 bytes)", "ratio_to_reference_loop_iterations": 0.9555,
                                                  "measured_on": "corpus.syn_text(16e6) at -window:28, tests/oracle_py.compress(want_stats=True)"}},
            "kernel_ms": {"prep": round(tm1["prep_ms"] - tm0["prep_ms"], 3),
                          "match_parse": round(tm1["match_parse_ms"] - tm0["match_parse_ms"], 3),
                          "rans_gather": round(tm1["rans_ms"] - tm0["rans_ms"], 3)},
            "counters": d,
        }
        res["roofline"]["latency_bound"]["bt4"]["tests_per_call"] = round(d["bt_tests"] / max(1, d["bt_calls"]), 1)
        if bit_exact is False:
            errors.append("the stream differs from the reference's")
        # HBM traffic of the same command, measured in separate rocprofv3 --pmc passes (tests/prof_run.sh) and
        # committed under profiles/: bench.py itself cannot read PMC counters
        try:
            import glob
            prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]
            pj = json.load(open(prof))
            if pj.get("kernel") == "pipeline2_kernel" and pj["bench_line_under_profiler"]["config"]["batch_chunks"] == B and world == 1:
                if pj.get("csrc_sha16") != csrc_sha16():
                    # (counters of other kernels than the ones timed here: not this run's traffic)
                    res["roofline"]["traffic_source"] = os.path.relpath(prof, ROOT) + " refused: recorded for other kernel sources (csrc_sha16 differs)"
                else:
                    t = pj["traffic_bytes_per_launch"]
                    res["roofline"]["traffic"] = int(t.get("total_calibrated", t["total_with_fetch_x2"]))
                    res["roofline"]["traffic_source"] = os.path.relpath(prof, ROOT) + (" (FETCH_SIZE calibrated on random 8-byte reads)" if "total_calibrated" in t else " (FETCH_SIZE x 2)")
        except Exception:
            pass
        res["cpu_baseline"] = cpu_leg.collect() if cpu_leg is not None else None
        if blocks is not None:
            res["blocks"] = blocks
            if "error" in blocks:
                errors.append("blocks: " + blocks["error"])
        if workloads is not None:
            res["workloads"] = workloads
            for name, w in workloads.items():
                if isinstance(w, dict) and (w.get("error") or w.get("bit_exact") is False):
                    errors.append(f"workload {name}: " + (w.get("error") or "the stream differs from the reference's"))
        try:
            res["gpu_max_hw_queues_effective"] = nlzm_amd.counter("gpu_max_hw_queues_effective")     # (4: the HIP runtime had started before the variable was set -- block mode then runs ~30 % slower)
        except nlzm_amd.NlzmError:
            pass
        if errors:
            res["error"] = "; ".join(errors)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if errors:
        sys.exit(1)


if __name__ == "__main__":
    main()
