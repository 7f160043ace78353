"""Ad-hoc: a slice of corpus.<kind>(size) compressed on the GPU against the oracle, under several option sets.
python tests/gpu_slice.py <kind> <size> <lo> <hi> <window> ["k=v k=v" ...]"""
import sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
from tests import oracle_py
kind, size, lo, hi, hb = sys.argv[1], int(float(sys.argv[2])), int(float(sys.argv[3])), int(float(sys.argv[4])), int(sys.argv[5])
data = corpus.make(kind, size)[lo:hi].copy()
want = oracle_py.compress(data, hb)
nlzm_amd.init(0)
for opts in (sys.argv[6:] or [""]):
    for kv in opts.split():
        k, v = kv.split("=")
        nlzm_amd.set_option(k, int(v))
    got = nlzm_amd.compress(data, hb)
    print(repr(opts), len(got), len(want), "EQUAL" if got == want else "DIFFERENT", flush=True)
