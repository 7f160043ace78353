#!/usr/bin/env python3
"""Generate tests/golden/streams.json from the REFERENCE itself.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container).
For every case in tests/cases.py it writes the seeded input to a scratch file, runs

    oracle/_ref/nlzm_ref        -window:N c in out      (plain reference build)
    oracle/_ref/nlzm_ref_instr  -window:N c in out2     (same source + digest hooks)

checks that both outputs are identical, and records size, SHA-256, the F2/F3/F4 digests
(oracle/digest.h) and, for streams <= 4 KiB, the bytes themselves.  The fixtures are data
only; the reference source or binary never enters tests/.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")
REF_I = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref_instr")


def run_case(case, tmp):
    name, kind, size, seed_off, wbits = case
    data = cases.make_case(case)
    inp = os.path.join(tmp, "in.bin")
    data.tofile(inp)
    outs = []
    for exe, tag in ((REF, "a"), (REF_I, "b")):
        out = os.path.join(tmp, f"out_{tag}.nlzm")
        if os.path.exists(out):
            os.remove(out)
        env = dict(os.environ, NLZM_DIGEST_OUT=os.path.join(tmp, "dig.json"))
        subprocess.run([exe, f"-window:{wbits}", "c", inp, out], check=True, capture_output=True, env=env)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1], f"{name}: instrumented reference output differs from the plain build"
    if size:
        dig = json.load(open(os.path.join(tmp, "dig.json")))
        dig.pop("snap", None)
    else:
        dig = {"n_pos": 0, "n_seg": 0, "n_frames": 0}
    rec = {"name": name, "kind": kind, "size": size, "seed_off": seed_off, "window": wbits,
           "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
           "stream_size": len(outs[0]), "stream_sha256": hashlib.sha256(outs[0]).hexdigest(), "digest": dig}
    if len(outs[0]) <= 4096:
        rec["stream_hex"] = outs[0].hex()
    # the reference's own decoder round-trips it (not for hist_bits < 12: reference bug)
    hb = (outs[0][0] << 8) | outs[0][1]
    rec["hist_bits"] = hb
    if hb >= 12:
        back = os.path.join(tmp, "back.bin")
        if os.path.exists(back):
            os.remove(back)
        subprocess.run([REF, "d", os.path.join(tmp, "out_a.nlzm"), back], check=True, capture_output=True)
        assert open(back, "rb").read() == data.tobytes(), f"{name}: reference round trip failed"
    return rec


def main():
    if not (os.path.exists(REF) and os.path.exists(REF_I)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    recs = []
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "oracle", "_ref")) as tmp:
        for case in cases.CASES + cases.BIG_CASES:
            rec = run_case(case, tmp)
            print(f"{rec['name']:>18}: {rec['size']:>9} -> {rec['stream_size']:>8}  {rec['stream_sha256'][:16]}")
            recs.append(rec)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "golden", "streams.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py", "reference": "nauful/NLZM 1.03 (NLZM.cpp)", "cases": recs}, f, indent=1)


if __name__ == "__main__":
    main()
