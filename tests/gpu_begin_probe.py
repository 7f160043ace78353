"""Ad-hoc GPU probe (not a test): how long opening a block set takes behind a large single stream.
python tests/gpu_begin_probe.py"""
import ctypes as C
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import numpy as np
import nlzm_amd
from nlzm_amd import corpus

nlzm_amd.init(0)
nlzm_amd.set_option("stage_report", 1)
lib = nlzm_amd.load_library()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
big = corpus.syn_text(400_000_000, corpus.SEED)          # -window:28 stays 28: the 16 GiB pre-filter table
blk = corpus.syn_text(32 * 4_000_000, corpus.SEED + 3)
cap = int(lib.nlzm_hip_compress_bound(big.size))
d_in, d_out = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(d_in), big.size + 512) == 0 and hip.hipMalloc(C.byref(d_out), cap) == 0
assert hip.hipMemcpy(d_in, big.ctypes.data, big.size, 1) == 0
for mode in ("blocks first", "blocks again", "after big", "after big + tiny", "after big + tiny + 3 s", "again"):
    if mode.startswith("after big"):
        nlzm_amd.set_option("batch_chunks", 8)
        t = time.time()
        assert lib.nlzm_hip_stream_begin(d_in, big.size, 28, d_out, cap) == 0
        done, out, fin = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        assert lib.nlzm_hip_stream_step(8, C.byref(done), C.byref(out), C.byref(fin)) == 0
        print(f"  big stream opened + one launch: {time.time() - t:.2f} s", flush=True)
    if "tiny" in mode:
        t = time.time(); nlzm_amd.compress(corpus.syn_text(4096), 20); print(f"  tiny stream (frees the big one's buffers): {time.time() - t:.2f} s", flush=True)
    if "3 s" in mode:
        time.sleep(3)
    t = time.time()
    got = nlzm_amd.compress_blocks(blk, 32, 28)
    print(f"{mode}: compress_blocks wall {time.time() - t:.2f} s", flush=True)
