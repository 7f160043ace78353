"""Ad-hoc GPU probe (not a test): one stream at different launch sizes.  python tests/gpu_batch.py <size> <window> <b,b,...>"""
import sys, time
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb = int(float(sys.argv[1])), int(sys.argv[2])
bs = [int(x) for x in sys.argv[3].split(',')]
nlzm_amd.init(0)
data = corpus.syn_text(size)
for b in bs:
    nlzm_amd.set_option("batch_chunks", b)
    t = time.time(); out = nlzm_amd.compress(data, hb); dt = time.time() - t
    tm = nlzm_amd.timing()
    print(f"batch {b}: {len(out)} B, wall {dt:.2f} s, device {tm['total_ms']:.0f} ms ({size / 1e3 / tm['total_ms']:.3f} MB/s), launches {tm['match_parse_launches']}", flush=True)
