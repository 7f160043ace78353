"""Ad-hoc GPU probe (not a test): k independent streams in flight on one GPU.
python tests/gpu_blocks.py [mb_per_block] [window] [k,k,...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import numpy as np
import nlzm_amd
from nlzm_amd import corpus, shard
from tests import oracle_py

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
nlzm_amd.init(0)
if os.environ.get("NLZM_WAIT_PRINT"):       # (a switch of this script: the library has no environment knobs)
    nlzm_amd.set_option("stage_report", 1)
for k in ks:
    data = corpus.syn_text(int(mb * 1e6) * k, corpus.SEED + 11)
    t = time.time()
    try:
        got = nlzm_amd.compress_blocks(data, k, hb)
    except Exception as e:
        print("k", k, "ERROR", e); continue
    dt = time.time() - t
    ok = True
    if mb * k <= 8:
        for i, s in enumerate(got):
            lo, hi = shard.block_range(data.size, k, i)
            ok &= s == oracle_py.compress(data[lo:hi], hb)
    print(f"k={k} {data.size} B w{hb}: {'OK' if ok else 'MISMATCH'} wall={dt:.2f}s {data.size/1e6/dt:.3f} MB/s out={sum(map(len, got))}")
    sys.stdout.flush()
