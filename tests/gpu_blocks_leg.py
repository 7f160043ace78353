"""Ad-hoc GPU probe (not a test): bench.py's block-mode leg alone, whole job, with library options set first.
python tests/gpu_blocks_leg.py [key=value ...] [-- key=value ...]     (each `--` starts another run in the same process: the pool is kept between sets)"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import torch
import nlzm_amd
from nlzm_amd import corpus
import bench

runs, cur = [], []
for a in sys.argv[1:]:
    if a == "--":
        runs.append(cur); cur = []
    else:
        cur.append(a)
runs.append(cur)
nlzm_amd.init(0)
lib = nlzm_amd.load_library()
dev = torch.device("cuda:0")
n = bench.STREAM_BYTES
data = corpus.syn_text(n)
d_in = torch.from_numpy(data).to(dev)
for kv in runs:
    for x in kv:
        k, v = x.split("=")
        nlzm_amd.set_option(k, int(v))
    c0 = bench.stage_counters()
    out = bench.blocks_leg(lib, torch, dev, d_in, n, 32, bench.BLOCKS_BATCH, 30, 5, True)
    c1 = bench.stage_counters()
    d = {k: c1[k] - c0[k] for k in c1 if isinstance(c1[k], int)}
    out.pop("workload", None)
    pos = max(1, d.get("positions", 1))
    print(kv, json.dumps(out), {k: round(v / pos, 2) for k, v in d.items() if k.endswith("cycles")}, flush=True)
