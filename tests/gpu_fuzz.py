"""Ad-hoc GPU job (not a test): corpus.splice inputs through the library against the oracle, as many seeds as fit the time.
python tests/gpu_fuzz.py <first seed> <seconds> [blocks=0]     (blocks > 0: every input also as that many independent blocks)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import nlzm_amd
from nlzm_amd import corpus, shard
from tests import oracle_py

first, budget = int(sys.argv[1]), float(sys.argv[2])
kblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nlzm_amd.init(0)
t0 = time.time()
pool = ThreadPoolExecutor(16)
pending, bad, done, nbytes = [], [], 0, 0
seed = first
def check(seed, data, hb, got, gotb):
    ok = oracle_py.compress(data, hb) == got
    if gotb is not None:
        for i, s in enumerate(gotb):
            lo, hi = shard.block_range(data.size, len(gotb), i)
            ok &= oracle_py.compress(data[lo:hi], hb) == s
    return seed, ok, data.size, hb
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    size, hb = int(np.exp(rng.uniform(np.log(200_000), np.log(5_000_000)))), int(rng.integers(15, 25))
    nlzm_amd.set_option("batch_chunks", int(rng.choice([1, 2, 5, 32])))
    nlzm_amd.set_option("table_shape", int(rng.integers(0, 3)))
    data = corpus.make("splice", size, seed)
    got = nlzm_amd.compress(data, hb)
    gotb = nlzm_amd.compress_blocks(data, kblocks, hb) if kblocks else None
    pending.append(pool.submit(check, seed, data, hb, got, gotb))
    seed += 1
for f in pending:
    s, ok, n, hb = f.result()
    done += 1; nbytes += n
    if not ok:
        bad.append((s, n, hb))
print(f"seeds {first}..{seed - 1}: {done - len(bad)} of {done} exact ({nbytes / 1e6:.0f} MB), {time.time() - t0:.0f} s" + (f"; DIFFERENT: {bad}" if bad else ""))
sys.exit(1 if bad else 0)
