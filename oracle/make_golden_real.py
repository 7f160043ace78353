#!/usr/bin/env python3
"""Generate tests/golden/real.json: size + SHA-256 of the REFERENCE's stream on REAL text -- the image's own source files
(nlzm_amd/corpus.real_text: Python's standard library, the C / C++ headers, the installed Python packages), where every other
fixture is a seeded generator.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container).  The input is pinned by its SHA-256: the
GPU boxes run the same image, and a test that finds other bytes there skips loudly instead of comparing.  Fixtures are data only.

    python oracle/make_golden_real.py          (real_30m_w24: ~1 min of CPU; real_300m_w28: ~10 min)
    python oracle/make_golden_real.py bench    (tests/golden/workloads.json: the two workloads bench.py runs beside its headline -- 120 MB of
                                                real text at -window:28 and 100 MB of wiki-shaped markup at -window:26; ~3 + ~2 min of CPU.
                                                `reference_seconds` is the reference's wall time on this container's core: bench.py quotes it
                                                as those workloads' CPU figure instead of running the reference again)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nlzm_amd import corpus  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
OUT = os.path.join(ROOT, "tests", "golden", "real.json")
CASES = [("real_30m_w24", "real_text", 30_000_000, 24), ("real_300m_w28", "real_text", 300_000_000, 28)]
BENCH_OUT = os.path.join(ROOT, "tests", "golden", "workloads.json")
BENCH_CASES = [("real_text_120m_w28", "real_text", 120_000_000, 28), ("xml_100m_w26", "xml_like", 100_000_000, 26)]


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    recs = []
    bench = len(sys.argv) > 1 and sys.argv[1] == "bench"
    out_path = BENCH_OUT if bench else OUT
    for name, kind, size, wbits in (BENCH_CASES if bench else CASES):
        data = corpus.make(kind, size)
        with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "oracle", "_ref")) as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
            data.tofile(inp)
            t0 = time.time()
            subprocess.run([REF, f"-window:{wbits}", "c", inp, out], check=True, capture_output=True)
            dt = time.time() - t0
            stream = open(out, "rb").read()
        recs.append({"name": name, "kind": kind, "size": size, "window": wbits, "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
                     "hist_bits": (stream[0] << 8) | stream[1], "stream_size": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
                     "reference_seconds": round(dt, 1), "reference_mb_per_s": round(size / 1e6 / dt, 3)})
        print(f"{name}: {size} -> {len(stream)}  {recs[-1]['stream_sha256'][:16]}  ({dt:.0f} s)", flush=True)
        with open(out_path, "w") as f:
            json.dump({"generator": "oracle/make_golden_real.py" + (" bench" if bench else ""), "reference": "nauful/NLZM 1.03 (NLZM.cpp)",
                       "input": "nlzm_amd/corpus.make(kind, size): real_text = the image's source files, concatenated; xml_like = the seeded markup generator",
                       "cases": recs}, f, indent=1)


if __name__ == "__main__":
    main()
