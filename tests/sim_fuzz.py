"""Fuzz of the pipeline stages in the fiber simulator against the oracle (not collected by pytest: a batch job for idle CPU time).
python tests/sim_fuzz.py <first seed> <count> [jobs=6] [kind=splice]     every seed: corpus.splice of 250 - 500 KB (kind real_text: a slice of that size out of the first 120 MB of
corpus.real_text, at a random offset), a window of 15 - 22 bits, 1 - 4 launches, either worker emulation"""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np
from nlzm_amd import corpus

SIM = os.path.join("tests", "host_sim", "sim2")
first, count = int(sys.argv[1]), int(sys.argv[2])
jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
kind = sys.argv[4] if len(sys.argv) > 4 else "splice"
REAL = corpus.make("real_text", 120_000_000) if kind == "real_text" else None
subprocess.run(["make", "-C", os.path.join("tests", "host_sim")], check=True, capture_output=True)

def one(seed):
    rng = np.random.default_rng(seed)
    lo, hi = (int(x) for x in os.environ.get("NLZM_FUZZ_SIZE", "250000,500000").split(","))      # (NLZM_FUZZ_SIZE=lo,hi: other sizes)
    size, hb = int(rng.integers(lo, hi)), int(rng.integers(15, 25 if hi > 600_000 else 23))
    workers, launches = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    env = dict(os.environ)
    if rng.random() < 0.3: env["NLZM_SIM_RANDOM_BLOCKS"] = str(int(rng.integers(1, 100)))
    if rng.random() < 0.3: env["NLZM_SIM_PSTRIDE"] = str(int(rng.integers(4, 40)))
    if rng.random() < 0.3: env["NLZM_SIM_TABLE_SHAPE"] = str(int(rng.integers(0, 3)))
    with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as f:
        if REAL is not None:
            off = int(rng.integers(0, REAL.size - size)); REAL[off:off + size].tofile(f.name)
        else:
            corpus.make("splice", size, seed).tofile(f.name)
        r = subprocess.run([SIM, f.name, str(hb), str(workers), str(launches)], capture_output=True, text=True, env=env)
    ok = r.returncode == 0 and ": OK" in r.stdout
    line = next((l for l in r.stdout.splitlines() if "cut-short" in l), "")
    return seed, ok, f"size {size} window {hb} workers {workers} launches {launches} " + " ".join(f"{k}={env[k]}" for k in env if k.startswith("NLZM_SIM_")) + " | " + line + ("" if ok else "\n" + r.stdout[-1500:] + r.stderr[-500:])

bad = 0
with ThreadPoolExecutor(jobs) as ex:
    for seed, ok, msg in ex.map(one, range(first, first + count)):
        print(("ok   " if ok else "FAIL ") + str(seed), msg, flush=True)
        bad += not ok
print(f"{count - bad} of {count} seeds exact")
sys.exit(1 if bad else 0)
