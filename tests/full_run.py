"""Whole-stream run of the enwik9 configuration (1e9-byte stand-in, -window:28) on one GPU.
Writes size, SHA-256 and timing of the compressed stream to gpurun_out/full_<tag>.json so that it can be
compared with the reference's output for the same seeded input (tests/full_ref.sh).  Not a pytest test:
it takes ~40 minutes at round-1 speed.   usage: python tests/full_run.py [bytes] [tag]"""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import nlzm_amd
from nlzm_amd import corpus

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
t0 = time.time(); data = corpus.syn_text(n); tgen = time.time() - t0
nlzm_amd.init(0); lib = nlzm_amd.load_library(); nlzm_amd.set_option("batch_chunks", 32)
dev = torch.device("cuda", 0)
d_in = torch.zeros(n + 4096, dtype=torch.uint8, device=dev); d_in[:n].copy_(torch.from_numpy(data))
cap = int(lib.nlzm_hip_compress_bound(n)); d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
rc = lib.nlzm_hip_stream_begin(d_in.data_ptr(), n, 28, d_out.data_ptr(), cap); assert rc == 0, lib.nlzm_hip_last_error()
i, o, f = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
t0 = time.time(); marks = []
while not f.value:
    rc = lib.nlzm_hip_stream_step(256, C.byref(i), C.byref(o), C.byref(f)); assert rc == 0, lib.nlzm_hip_last_error()
    marks.append((i.value, o.value, round(time.time() - t0, 1)))
    json.dump({"partial": marks}, open(f"gpurun_out/full_{tag}_progress.json", "w"))
ln = C.c_uint64(0); rc = lib.nlzm_hip_stream_finish(C.byref(ln)); assert rc == 0
torch.cuda.synchronize(); dt = time.time() - t0
out = d_out[:ln.value].cpu().numpy().tobytes()
res = {"input_bytes": n, "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(), "window": 28, "stream_bytes": len(out),
       "stream_sha256": hashlib.sha256(out).hexdigest(), "seconds": round(dt, 1), "MB_per_s": round(n / 1e6 / dt, 4),
       "progress": marks[::8], "stats": nlzm_amd.stats(), "timing": nlzm_amd.timing(), "gen_seconds": round(tgen, 1)}
json.dump(res, open(f"gpurun_out/full_{tag}.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("input_bytes", "stream_bytes", "stream_sha256", "seconds", "MB_per_s")}))
