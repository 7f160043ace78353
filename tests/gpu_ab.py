"""Helper parser on / off on the same inputs: python tests/gpu_ab.py <kind:size:window> ...   (prints size + hash of both streams)"""
import sys, hashlib
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
nlzm_amd.init(0)
bad = 0
for spec in sys.argv[1:]:
    kind, size, hb = spec.split(":")
    data = corpus.make(kind, int(float(size)))
    res = []
    for h in (0, 1):
        nlzm_amd.set_option("parser_helper", h)
        out = nlzm_amd.compress(data, int(hb))
        res.append((len(out), hashlib.sha256(out).hexdigest()[:16], round(nlzm_amd.timing()["match_parse_ms"])))
    same = res[0][:2] == res[1][:2]
    bad += not same
    print(spec, res, "same" if same else "DIFFERENT", flush=True)
sys.exit(1 if bad else 0)
