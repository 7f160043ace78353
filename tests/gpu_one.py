"""Single compress call for profiling: python tests/gpu_one.py <size> <window> <workers> [kind]"""
import os, sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb, w = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "syn_text"
nlzm_amd.init(0)
if os.environ.get("NLZM_WAIT_PRINT"):       # (a switch of this script: the library has no environment knobs)
    nlzm_amd.set_option("stage_report", 1)
nlzm_amd.set_option("workers", w)
data = corpus.make(kind, size)
out = nlzm_amd.compress(data, hb)
print(len(out), nlzm_amd.timing())
