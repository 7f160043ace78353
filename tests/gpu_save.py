"""Ad-hoc: compress <kind> <size> at <window> on the GPU and save the stream under gpurun_out/ (to be compared with the reference's at home)."""
import sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
kind, size, hb = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
nlzm_amd.init(0)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    nlzm_amd.set_option(k, int(v))
out = nlzm_amd.compress(corpus.make(kind, size), hb)
open(f"gpurun_out/{kind}_{size}_w{hb}.gpu.nlzm", "wb").write(out)
print(kind, size, hb, len(out), nlzm_amd.timing()["match_parse_ms"])
