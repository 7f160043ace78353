#!/usr/bin/env python3
"""Generate tests/golden/gpus.json: size + SHA-256 of the REFERENCE's stream for every rank's block of `bench.py --gpus N`.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container).  BASELINE config 5 is the
1e9-byte workload split into N independent blocks, one NLZM stream per GPU; its parity definition is the
reference run on each block alone (encode_file, NLZM.cpp:1711, with the window auto-shrink of :1716-1718).  bench.py feeds
rank r of N with corpus.syn_text(ceil(1e9 / N), SEED + r); this script writes exactly those inputs to scratch files and
compresses each with oracle/_ref/nlzm_ref (the reference sources compiled unchanged, oracle/Makefile).  Fixtures are data only.

    python oracle/make_golden_gpus.py [N ...] [-j jobs]      (default N: 8; jobs: 6.  N = 8: eight runs of ~5 min of CPU each)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nlzm_amd import corpus, shard  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
OUT = os.path.join(ROOT, "tests", "golden", "gpus.json")
STREAM_BYTES, WINDOW = 1_000_000_000, 28


def one(job):
    world, rank = job
    lo, hi = shard.block_range(STREAM_BYTES, world, rank)
    data = corpus.syn_text(hi - lo, corpus.SEED + rank)
    with tempfile.TemporaryDirectory(dir=os.environ.get("NLZM_SCRATCH", "/tmp")) as tmp:
        inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
        data.tofile(inp)
        t0 = time.time()
        subprocess.run([REF, f"-window:{WINDOW}", "c", inp, out], check=True, capture_output=True)
        dt = time.time() - t0
        h, n = hashlib.sha256(), 0
        with open(out, "rb") as f:
            hdr = f.read(4)
            h.update(hdr); n += len(hdr)
            for b in iter(lambda: f.read(1 << 24), b""):
                h.update(b); n += len(b)
    rec = {"world": world, "rank": rank, "size": hi - lo, "seed_off": rank, "window": WINDOW,
           "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "hist_bits": (hdr[0] << 8) | hdr[1],
           "stream_size": n, "stream_sha256": h.hexdigest(), "reference_seconds": round(dt, 1)}
    print(f"N={world} rank {rank}: {hi - lo} -> {n}  {h.hexdigest()[:16]}  ({dt:.0f} s)", flush=True)
    return rec


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    args = sys.argv[1:]
    jobs = 6
    if "-j" in args:
        i = args.index("-j")
        jobs = int(args[i + 1])
        del args[i:i + 2]
    worlds = [int(a) for a in args] or [8]
    have = {}
    if os.path.exists(OUT):
        have = {(r["world"], r["rank"]): r for r in json.load(open(OUT))["ranks"]}
    todo = [(w, r) for w in worlds for r in range(w) if (w, r) not in have]
    with ThreadPoolExecutor(jobs) as ex:
        for rec in ex.map(one, todo):
            have[(rec["world"], rec["rank"])] = rec
            with open(OUT, "w") as f:
                json.dump({"generator": "oracle/make_golden_gpus.py",
                           "reference": "nauful/NLZM 1.03 (NLZM.cpp), one run per rank's block",
                           "input": "corpus.syn_text(ceil(1e9 / world), SEED + rank), -window:28",
                           "ranks": [have[k] for k in sorted(have)]}, f, indent=1)


if __name__ == "__main__":
    main()
