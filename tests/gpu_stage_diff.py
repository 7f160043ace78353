"""Ad-hoc: which stage differs?  Match tables of a chunk and the symbol / bit streams of its frame, GPU against the oracle.
python tests/gpu_stage_diff.py <kind> <size> <window> <frame index>"""
import sys
sys.path.insert(0, '.')
import numpy as np
import nlzm_amd
from nlzm_amd import corpus
from tests import oracle_py
kind, size, hb, fi = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
data = corpus.make(kind, size)
nlzm_amd.init(0)
CH = 122368
lo, hi = fi * CH, min(size, (fi + 1) * CH)
got = nlzm_amd.find_matches(data, hb, lo, hi, 1 << 26)
want = oracle_py.capture_tables(data, hb, lo, hi, 1 << 26)
print("tables:", got.size, want.size, "EQUAL" if got.size == want.size and np.array_equal(got, want) else "DIFFERENT", flush=True)
if not (got.size == want.size and np.array_equal(got, want)):
    # walk both lists: {position, max_len, delta[2..max_len]}
    ig = iw = 0
    shown = 0
    while ig < got.size and iw < want.size and shown < 5:
        pg, mg = int(got[ig]), int(got[ig + 1]); pw, mw = int(want[iw]), int(want[iw + 1])
        ng, nw = 2 + max(0, mg - 1), 2 + max(0, mw - 1)
        if pg != pw or mg != mw or not np.array_equal(got[ig:ig + ng], want[iw:iw + nw]):
            print(" position", pg, pw, "max_len", mg, mw)
            print("   gpu   ", got[ig + 2:ig + min(ng, 40)].tolist())
            print("   oracle", want[iw + 2:iw + min(nw, 40)].tolist())
            shown += 1
        ig += ng; iw += nw
gs, gb, gm = nlzm_amd.parse_emit(data, hb, fi)
ws, wb, wm, _ = oracle_py.capture_frame(data, hb, fi)
print("frame:", len(gs), len(ws), len(gb), len(wb), gm, wm, "EQUAL" if np.array_equal(gs, ws) and np.array_equal(gb, wb) else "DIFFERENT", flush=True)
if not np.array_equal(gs, ws):
    n = min(len(gs), len(ws)); d = np.flatnonzero(np.asarray(gs[:n]) != np.asarray(ws[:n]))
    print(" first differing symbol", int(d[0]) if d.size else n, "of", len(gs), len(ws))
