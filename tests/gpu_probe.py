"""Ad-hoc GPU probe (not a test): python tests/gpu_probe.py [size_mb] [window] [workers]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import nlzm_amd
from nlzm_amd import corpus
from tests import oracle_py

size = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3_000_000
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
modes = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1]
kind = sys.argv[4] if len(sys.argv) > 4 else "syn_text"
check = size <= 20_000_000 and "NLZM_LIB" not in os.environ
nlzm_amd.init(0)
if os.environ.get("NLZM_WB"):
    nlzm_amd.set_option("worker_blocks", int(os.environ["NLZM_WB"]))
data = corpus.make(kind, size)
want = oracle_py.compress(data, hb) if check else None
for w in modes:
    nlzm_amd.set_option("workers", w)
    t = time.time()
    try:
        got = nlzm_amd.compress(data, hb)
    except Exception as e:
        print("workers", w, "ERROR", e); continue
    dt = time.time() - t
    tm = nlzm_amd.timing(); st = nlzm_amd.stats()
    ok = "unchecked" if want is None else ("OK" if got == want else "MISMATCH")
    print(f"workers={w} {kind} {size} w{hb}: {ok} out={len(got)} wall={dt:.2f}s {size/1e6/dt:.3f} MB/s  kernel={tm['match_parse_ms']:.0f}ms "
          f"({size/1e3/tm['match_parse_ms']:.3f} MB/s) prep={tm['prep_ms']:.1f}ms rans={tm['rans_ms']:.1f}ms unc={st['uncertain_positions']} "
          f"bt_tests={st['bt_tests']}")
    sys.stdout.flush()
