"""Worker-CU sweep of one stream: python tests/gpu_sweep.py <size> <window> <worker_blocks> [<worker_blocks> ...]
(the stages take 3 CUs; worker_blocks CUs run the BT4 lanes and the hot bins' waves)"""
import sys, time
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb = int(float(sys.argv[1])), int(sys.argv[2])
nlzm_amd.init(0)
data = corpus.make("syn_text", size)
for wb in sys.argv[3:]:
    nlzm_amd.set_option("worker_blocks", int(wb))
    out = nlzm_amd.compress(data, hb)
    tm = nlzm_amd.timing()
    print(f"worker_blocks {wb}: {len(out)} B, pipeline {tm['match_parse_ms']:.0f} ms, {size / 1e3 / tm['match_parse_ms']:.3f} MB/s", flush=True)
