"""Ad-hoc GPU diagnosis (not a test): where does the GPU stream of a full-size case leave the oracle's?
    python tests/gpu_diverge.py text_100m_w26
Finds the first frame that differs, then compares the match tables and the symbol stream of that frame."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import nlzm_amd
from tests import cases, oracle_py

name = sys.argv[1] if len(sys.argv) > 1 else "text_100m_w26"
case = next(c for c in cases.FULL_CASES + cases.BIG_CASES + cases.CASES if c[0] == name)
data = cases.make_case(case)
hb = case[4]
nlzm_amd.init(0)
t = time.time(); got = nlzm_amd.compress(data, hb); print(f"gpu: {len(got)} bytes in {time.time()-t:.1f} s", flush=True)
t = time.time(); want = oracle_py.compress(data, hb); print(f"oracle: {len(want)} bytes in {time.time()-t:.1f} s", flush=True)


def frames(s):
    out, pos = [], 4
    while True:
        ops = int.from_bytes(s[pos:pos + 4], "big")
        if ops == 0:
            break
        nb = int.from_bytes(s[pos + 4:pos + 8], "big"); nr = int.from_bytes(s[pos + 8:pos + 12], "big")
        out.append((pos, nb + nr, ops))
        pos += nb + nr
    return out


fg, fw = frames(got), frames(want)
print("frames", len(fg), len(fw))
first = next((i for i in range(min(len(fg), len(fw))) if got[fg[i][0]:fg[i][0] + fg[i][1]] != want[fw[i][0]:fw[i][0] + fw[i][1]]), None)
print("first differing frame", first, fg[first] if first is not None else None, fw[first] if first is not None else None, flush=True)
if first is None:
    sys.exit(0)
chunk = nlzm_amd.geometry(data.size, hb)["chunk_size"]
lo, hi = first * chunk, min(data.size, (first + 1) * chunk)
tg = nlzm_amd.find_matches(data, hb, lo, hi)
tw = oracle_py.capture_tables(data, hb, lo, hi)
print("tables: sizes", tg.size, tw.size, "equal", tg.size == tw.size and bool(np.array_equal(tg, tw)), flush=True)
if not (tg.size == tw.size and np.array_equal(tg, tw)):
    i = 0
    while i < min(tg.size, tw.size):
        pg, mg, pw, mw = tg[i], tg[i + 1], tw[i], tw[i + 1]
        ng = 2 + (mg - 1 if mg >= 2 else 0); nw = 2 + (mw - 1 if mw >= 2 else 0)
        if pg != pw or mg != mw or not np.array_equal(tg[i:i + ng], tw[i:i + nw]):
            print("first differing position", int(pg), int(pw), "max_len gpu", int(mg), "oracle", int(mw))
            print(" gpu   :", tg[i:i + min(ng, 40)].tolist())
            print(" oracle:", tw[i:i + min(nw, 40)].tolist())
            break
        i += ng
sg, bg, og = nlzm_amd.parse_emit(data, hb, first)
sw, bw, ow, _ = oracle_py.capture_frame(data, hb, first)
n = min(sg.size, sw.size)
d = next((i for i in range(n) if sg[i] != sw[i]), None)
print("frame symbols: gpu", sg.size, "oracle", sw.size, "ops", og, ow, "first differing symbol", d, flush=True)
