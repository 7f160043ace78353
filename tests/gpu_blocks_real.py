"""Ad-hoc GPU probe (not a test): source code (corpus.real_text) as k independent blocks in flight on one GPU; a few of the streams checked against the oracle.
python tests/gpu_blocks_real.py [total_mb=300] [window=28] [k=32] [check=4]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
import nlzm_amd
from nlzm_amd import corpus, shard
from tests import oracle_py

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 28
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ncheck = int(sys.argv[4]) if len(sys.argv) > 4 else 4
nlzm_amd.init(0)
data = corpus.make("real_text", int(mb * 1e6))
for rep in range(2):        # (the second set uses the pool the first one left)
    t = time.time()
    got = nlzm_amd.compress_blocks(data, k, hb)
    dt = time.time() - t
    print(f"set {rep}: k={k} {data.size} B of real text, -window:{hb}: wall {dt:.2f} s = {data.size / 1e6 / dt:.2f} MB/s, out {sum(map(len, got))}", flush=True)
idx = [i * k // ncheck for i in range(ncheck)]
def ref(i):
    lo, hi = shard.block_range(data.size, k, i)
    return i, oracle_py.compress(data[lo:hi], hb)
t = time.time()
with ThreadPoolExecutor(ncheck) as ex:
    bad = [i for i, want in ex.map(ref, idx) if want != got[i]]
print(f"blocks {idx} against the oracle ({time.time() - t:.0f} s of CPU beside each other): {'bit-exact' if not bad else 'DIFFERENT: ' + str(bad)}")
sys.exit(1 if bad else 0)
