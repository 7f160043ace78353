"""Single compress call for profiling: python tests/gpu_one.py <size> <window> <workers> [kind]"""
import sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb, w = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "syn_text"
nlzm_amd.init(0)
nlzm_amd.set_option("workers", w)
data = corpus.make(kind, size)
out = nlzm_amd.compress(data, hb)
print(len(out), nlzm_amd.timing())
