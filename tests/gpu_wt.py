"""Ad-hoc GPU probe (not a test): one stream with different numbers of worker lanes per CU.  python tests/gpu_wt.py <size> <window> <t,t,..>"""
import os, sys, time
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb = int(float(sys.argv[1])), int(sys.argv[2])
ts = [int(x) for x in sys.argv[3].split(',')]
nlzm_amd.init(0)
if os.environ.get("NLZM_WAIT_PRINT"):       # (a switch of this script: the library has no environment knobs)
    nlzm_amd.set_option("stage_report", 1)
data = corpus.syn_text(size)
ref = None
for t in ts:
    nlzm_amd.set_option("worker_threads", t)
    out = nlzm_amd.compress(data, hb)
    tm = nlzm_amd.timing()
    ref = ref or out
    print(f"worker_threads {t}: {len(out)} B same={out == ref}, device {tm['total_ms']:.0f} ms ({size / 1e3 / tm['total_ms']:.3f} MB/s)", flush=True)
