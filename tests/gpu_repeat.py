"""Ad-hoc GPU check (not a test): compress a full-size case several times and compare with the reference's size / SHA.
    python tests/gpu_repeat.py text_100m_w26 3"""
import hashlib, json, os, sys, time
sys.path.insert(0, '.')
import nlzm_amd
from tests import cases
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
full = {c["name"]: c for c in json.load(open("tests/golden/full.json"))["cases"]}
g = full[name]
case = next(c for c in cases.FULL_CASES if c[0] == name)
data = cases.make_case(case)
nlzm_amd.init(0)
for r in range(reps):
    t = time.time(); got = nlzm_amd.compress(data, case[4]); dt = time.time() - t
    print(f"run {r}: {len(got)} bytes ({'EQUAL' if (len(got), hashlib.sha256(got).hexdigest()) == (g['stream_size'], g['stream_sha256']) else 'DIFFERENT'}; reference {g['stream_size']}) in {dt:.1f} s", flush=True)
