#!/usr/bin/env python3
"""Generate tests/golden/full.json: size + SHA-256 of the REFERENCE's stream on the full-size BASELINE configurations.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container; minutes of CPU per case).
For every case in tests/cases.FULL_CASES the seeded input is written to a scratch file and compressed by
oracle/_ref/nlzm_ref (the reference sources compiled unchanged, oracle/Makefile).  Fixtures are data only.

    python oracle/make_golden_full.py [case names...]      (default: every case that is not in the file yet;
                                                            the 1e9-byte case takes about an hour of CPU)
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
from tests import cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
OUT = os.path.join(ROOT, "tests", "golden", "full.json")


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    have = {c["name"]: c for c in json.load(open(OUT))["cases"]} if os.path.exists(OUT) else {}
    want = sys.argv[1:] or [c[0] for c in cases.FULL_CASES if c[0] not in have]
    for case in cases.FULL_CASES:
        name, kind, size, seed_off, wbits = case
        if name not in want:
            continue
        data = cases.make_case(case)
        with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "oracle", "_ref")) as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
            data.tofile(inp)
            t0 = time.time()
            subprocess.run([REF, f"-window:{wbits}", "c", inp, out], check=True, capture_output=True)
            dt = time.time() - t0
            h, n = hashlib.sha256(), 0
            with open(out, "rb") as f:
                hdr = f.read(4)
                h.update(hdr); n += len(hdr)
                while True:
                    b = f.read(1 << 24)
                    if not b:
                        break
                    h.update(b); n += len(b)
        have[name] = {"name": name, "kind": kind, "size": size, "seed_off": seed_off, "window": wbits,
                      "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "hist_bits": (hdr[0] << 8) | hdr[1],
                      "stream_size": n, "stream_sha256": h.hexdigest(), "reference_seconds": round(dt, 1)}
        print(f"{name}: {size} -> {n}  {h.hexdigest()[:16]}  ({dt:.0f} s)", flush=True)
        with open(OUT, "w") as f:
            json.dump({"generator": "oracle/make_golden_full.py", "reference": "nauful/NLZM 1.03 (NLZM.cpp)",
                       "cases": [have[c[0]] for c in cases.FULL_CASES if c[0] in have]}, f, indent=1)


if __name__ == "__main__":
    main()
