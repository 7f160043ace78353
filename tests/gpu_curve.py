"""Ad-hoc GPU probe (not a test): block mode at a fixed block size over the number of streams in flight (the contention curve).
python tests/gpu_curve.py <mb_per_block> <window> <k,k,...> [option=value ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus

mb, hb = float(sys.argv[1]), int(sys.argv[2])
ks = [int(x) for x in sys.argv[3].split(',')]
nlzm_amd.init(0)
if os.environ.get("NLZM_WAIT_PRINT"):       # (a switch of this script: the library has no environment knobs)
    nlzm_amd.set_option("stage_report", 1)
for kv in sys.argv[4:]:
    key, val = kv.split('=')
    nlzm_amd.set_option(key, int(val))
per = int(mb * 1e6)
data = corpus.syn_text(per * max(ks), corpus.SEED + 11)
for k in ks:
    t = time.time()
    got = nlzm_amd.compress_blocks(data[: per * k], k, hb)
    dt = time.time() - t
    tm = nlzm_amd.timing()
    print(" ".join(sys.argv[4:]), end=": ")
    print(f"k={k:3d} streams x {mb} MB, -window:{hb}: wall {dt:6.2f} s = {per * k / 1e6 / dt:7.2f} MB/s ({per / 1e6 / dt:5.2f} MB/s per stream), out {sum(map(len, got))}", flush=True)
