#!/usr/bin/env python3
"""Generate tests/golden/blocks.json: size + SHA-256 of the REFERENCE's stream for every block of tests/cases.BLOCK_SET.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (the build container).  The reference side of block
mode is encode_file (NLZM.cpp:1711) called once per byte range: each block is written to a scratch file and compressed by
oracle/_ref/nlzm_ref (the reference sources compiled unchanged, oracle/Makefile), a few at a time.  Fixtures are data only.

    python oracle/make_golden_blocks.py [parallel jobs, default 6] [1g]
        (no set named: tests/cases.BLOCK_SET -> tests/golden/blocks.json, what the GPU suite checks;
         1g: tests/cases.BLOCK_SET_1G -> tests/golden/blocks_1g.json, what bench.py's block-mode leg checks)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nlzm_amd import shard  # noqa: E402
from tests import cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "nlzm_ref")


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    big = "1g" in sys.argv[1:]
    the_set = cases.BLOCK_SET_1G if big else cases.BLOCK_SET
    OUT = os.path.join(ROOT, "tests", "golden", "blocks_1g.json" if big else "blocks.json")
    name, kind, size, seed_off, wbits, k = the_set
    data = cases.make_case(the_set[:5])
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    jobs = int(nums[0]) if nums else 6

    def one(i):
        lo, hi = shard.block_range(data.size, k, i)
        with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "oracle", "_ref")) as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.nlzm")
            data[lo:hi].tofile(inp)
            subprocess.run([REF, f"-window:{wbits}", "c", inp, out], check=True, capture_output=True)
            s = open(out, "rb").read()
        print(f"block {i}: {hi - lo} -> {len(s)}", flush=True)
        return {"block": i, "lo": lo, "hi": hi, "hist_bits": (s[0] << 8) | s[1], "stream_size": len(s),
                "stream_sha256": hashlib.sha256(s).hexdigest()}

    with ThreadPoolExecutor(jobs) as ex:
        recs = list(ex.map(one, range(k)))
    with open(OUT, "w") as f:
        json.dump({"generator": "oracle/make_golden_blocks.py", "reference": "nauful/NLZM 1.03 (NLZM.cpp), one run per block",
                   "name": name, "kind": kind, "size": size, "seed_off": seed_off, "window": wbits, "nblocks": k,
                   "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "blocks": recs}, f, indent=1)


if __name__ == "__main__":
    main()
