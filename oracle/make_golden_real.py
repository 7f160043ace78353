#!/usr/bin/env python3
"""Generate tests/golden/real.json: size + SHA-256 of the REFERENCE's stream on REAL text -- the image's own source files
(nlzm_amd/corpus.real_text: Python's standard library, the C / C++ headers, the installed Python packages), where every other
fixture is a seeded generator.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container).  The input is pinned by its SHA-256: the
GPU boxes run the same image, and a test that finds other bytes there skips loudly instead of comparing.  Fixtures are data only.

    python oracle/make_golden_real.py          (real_30m_w24: ~1 min of CPU; real_300m_w28: ~10 min)
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
CASES = [("real_30m_w24", 30_000_000, 24), ("real_300m_w28", 300_000_000, 28)]


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    recs = []
    for name, size, wbits in CASES:
        data = corpus.real_text(size)
        with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "oracle", "_ref")) as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
            data.tofile(inp)
            t0 = time.time()
            subprocess.run([REF, f"-window:{wbits}", "c", inp, out], check=True, capture_output=True)
            dt = time.time() - t0
            stream = open(out, "rb").read()
        recs.append({"name": name, "kind": "real_text", "size": size, "window": wbits, "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
                     "hist_bits": (stream[0] << 8) | stream[1], "stream_size": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
                     "reference_seconds": round(dt, 1), "reference_mb_per_s": round(size / 1e6 / dt, 3)})
        print(f"{name}: {size} -> {len(stream)}  {recs[-1]['stream_sha256'][:16]}  ({dt:.0f} s)", flush=True)
        with open(OUT, "w") as f:
            json.dump({"generator": "oracle/make_golden_real.py", "reference": "nauful/NLZM 1.03 (NLZM.cpp)",
                       "input": "nlzm_amd/corpus.real_text(size): the image's source files, concatenated", "cases": recs}, f, indent=1)


if __name__ == "__main__":
    main()
