"""Single compress call with options: python tests/gpu_opt.py <size> <window> <kind> [key=value ...]   (stage report on)"""
import sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
size, hb, kind = int(float(sys.argv[1])), int(sys.argv[2]), sys.argv[3]
nlzm_amd.init(0)
nlzm_amd.set_option("stage_report", 1)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    nlzm_amd.set_option(k, int(v))
data = corpus.make(kind, size)
out = nlzm_amd.compress(data, hb)
import hashlib
print(len(out), hashlib.sha256(out).hexdigest()[:16], nlzm_amd.timing())
