"""nlzm_amd -- MI355X (gfx950) implementation of NLZM 1.03's compress-side hot path.

The product is the C-ABI shared library ``libnlzm_hip.so`` (include/nlzm_hip.h) and the
``nlzm`` command line built from ``nlzm_amd/csrc``.  This module is the thin ctypes
binding the tests and ``bench.py`` use; it adds no algorithmic code and has no CPU
fallback: every call below fails loudly if the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

# block mode keeps several persistent launches in flight; the HIP runtime's default of 4 hardware queues would serialise
# them (only effective if the runtime has not started yet)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NLZM_LIB", os.path.join(_HERE, "libnlzm_hip.so"))     # override: diagnostic builds only
CLI_PATH = os.path.join(_HERE, "nlzm")

# every symbol include/nlzm_hip.h declares
ABI_SYMBOLS = [
    "nlzm_hip_init", "nlzm_hip_shutdown", "nlzm_hip_last_error", "nlzm_hip_compress_bound",
    "nlzm_hip_geometry", "nlzm_hip_compress", "nlzm_hip_compress_dev", "nlzm_hip_stream_begin",
    "nlzm_hip_stream_step", "nlzm_hip_stream_finish", "nlzm_hip_get_stats", "nlzm_hip_get_timing",
    "nlzm_hip_rans_frames", "nlzm_hip_find_matches", "nlzm_hip_parse_emit", "nlzm_hip_set_option",
    "nlzm_hip_blocks_begin", "nlzm_hip_blocks_step", "nlzm_hip_blocks_finish", "nlzm_hip_blocks_abandon",
    "nlzm_hip_compress_blocks_dev", "nlzm_hip_compress_blocks", "nlzm_hip_compress_blocks_multi",
    "nlzm_hip_feed_begin", "nlzm_hip_feed", "nlzm_hip_feed_output", "nlzm_hip_feed_finish", "nlzm_hip_feed_end",
    "nlzm_hip_block_placement", "nlzm_hip_get_counter",
]


class NlzmError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "in_bytes", "out_bytes", "bt_calls", "bt_tests", "cmp_bytes", "ht_rows", "rk_probes", "rk_inserts",
        "positions", "nice_positions", "segments", "n_literal", "n_dict", "n_rep", "rans_syms", "bit_ops",
        "frames", "shifts", "uncertain_positions")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("prep_ms", C.c_double),
                ("match_parse_ms", C.c_double), ("rans_ms", C.c_double), ("match_parse_launches", C.c_uint32),
                ("rans_launches", C.c_uint32), ("prep_launches", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> None:
    """Compile the gfx950 library and the CLI in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise NlzmError("building libnlzm_hip.so failed:\n" + r.stdout + r.stderr)


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library (no device needed just to load it)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NlzmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.nlzm_hip_init.argtypes = [C.c_int]
    lib.nlzm_hip_last_error.restype = C.c_char_p
    lib.nlzm_hip_compress_bound.argtypes = [C.c_uint64]
    lib.nlzm_hip_compress_bound.restype = C.c_uint64
    lib.nlzm_hip_geometry.argtypes = [C.c_uint64, C.c_uint32, u32p, u32p, u32p, u32p]
    lib.nlzm_hip_geometry.restype = None
    lib.nlzm_hip_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
    lib.nlzm_hip_compress_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
    lib.nlzm_hip_stream_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.nlzm_hip_stream_step.argtypes = [C.c_uint32, u64p, u64p, C.POINTER(C.c_int)]
    lib.nlzm_hip_stream_finish.argtypes = [u64p]
    lib.nlzm_hip_get_stats.argtypes = [C.POINTER(Stats)]
    lib.nlzm_hip_get_timing.argtypes = [C.POINTER(Timing)]
    if hasattr(lib, "nlzm_hip_get_counter"):       # (absent from older diagnostic builds loaded through NLZM_LIB)
        lib.nlzm_hip_get_counter.argtypes = [C.c_char_p, u64p]
    lib.nlzm_hip_rans_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint64, C.c_void_p]
    lib.nlzm_hip_find_matches.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p,
                                          C.c_uint64, u64p]
    lib.nlzm_hip_parse_emit.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_void_p]
    lib.nlzm_hip_set_option.argtypes = [C.c_char_p, C.c_int64]
    lib.nlzm_hip_blocks_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.nlzm_hip_blocks_step.argtypes = [C.c_uint32, u64p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.nlzm_hip_blocks_finish.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
    lib.nlzm_hip_blocks_abandon.restype = None
    lib.nlzm_hip_compress_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p, u64p]
    lib.nlzm_hip_feed_begin.argtypes = [C.c_uint64, C.c_uint32]
    lib.nlzm_hip_feed.argtypes = [C.c_void_p, C.c_uint64]
    lib.nlzm_hip_feed_output.argtypes = [C.c_void_p, C.c_uint64, u64p]
    lib.nlzm_hip_feed_end.restype = None
    lib.nlzm_hip_compress_blocks_multi.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                                                   C.c_void_p, C.c_uint64, u64p, u64p]
    lib.nlzm_hip_compress_blocks_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p, u64p]
    _lib = lib
    return lib


def _chk(rc: int) -> None:
    if rc != 0:
        raise NlzmError(f"nlzm_hip error {rc}: {load_library().nlzm_hip_last_error().decode()}")


def init(device: int = 0) -> None:
    _chk(load_library().nlzm_hip_init(device))


def shutdown() -> None:
    load_library().nlzm_hip_shutdown()


def set_option(key: str, value: int) -> None:
    _chk(load_library().nlzm_hip_set_option(key.encode(), int(value)))


def geometry(flen: int, hist_bits: int) -> dict:
    lib = load_library()
    v = [C.c_uint32() for _ in range(4)]
    lib.nlzm_hip_geometry(flen, hist_bits, *[C.byref(x) for x in v])
    return dict(zip(("hist_bits", "frame_bits", "chunk_size", "feed_size"), (int(x.value) for x in v)))


def compress(data, hist_bits: int = 22) -> bytes:
    """encode_file replacement on host buffers (NLZM.cpp:1711); returns the whole stream."""
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    n = int(src.size)
    cap = int(lib.nlzm_hip_compress_bound(n))
    dst = np.empty(cap, dtype=np.uint8)
    out_len = C.c_uint64(0)
    _chk(lib.nlzm_hip_compress(src.ctypes.data if n else None, n, hist_bits, dst.ctypes.data, cap, C.byref(out_len)))
    return dst[: out_len.value].tobytes()


def compress_blocks(data, nblocks: int, hist_bits: int = 22) -> list[bytes]:
    """k independent streams, all in flight on the one GPU (shard.block_range gives the byte ranges); the reference
    equivalent is encode_file (NLZM.cpp:1711) run on each range."""
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    n = int(src.size)
    cap = int(lib.nlzm_hip_compress_bound(n)) + nblocks * (16 + 131072)
    dst = np.empty(cap, dtype=np.uint8)
    lens = (C.c_uint64 * nblocks)()
    out_len = C.c_uint64(0)
    _chk(lib.nlzm_hip_compress_blocks(src.ctypes.data if n else None, n, nblocks, hist_bits, dst.ctypes.data, cap, lens, C.byref(out_len)))
    out, pos = [], 0
    for i in range(nblocks):
        out.append(dst[pos: pos + int(lens[i])].tobytes())
        pos += int(lens[i])
    assert pos == out_len.value
    return out


def compress_fed(data, hist_bits: int = 22, piece: int = 1 << 20) -> bytes:
    """The streaming form of compress(): the input handed over `piece` bytes at a time (pinned staging, uploads overlapped
    with the launches), the stream taken back as its frames are finished (NLZM.cpp:1774-1778, :1853, :1870-1885)."""
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    n = int(src.size)
    _chk(lib.nlzm_hip_feed_begin(n, hist_bits))
    buf = np.empty(1 << 20, dtype=np.uint8)
    got = C.c_uint64(0)
    out = []

    def drain():
        while True:
            _chk(lib.nlzm_hip_feed_output(buf.ctypes.data, buf.size, C.byref(got)))
            if not got.value:
                return
            out.append(buf[: got.value].tobytes())

    try:
        for lo in range(0, n, piece):
            m = min(piece, n - lo)
            _chk(lib.nlzm_hip_feed(src[lo:].ctypes.data, m))
            drain()
        _chk(lib.nlzm_hip_feed_finish())
        drain()
    finally:
        lib.nlzm_hip_feed_end()
    return b"".join(out)


def compress_blocks_multi(data, devices: list[int], blocks_per_dev: int, hist_bits: int = 22) -> list[bytes]:
    """len(devices) * blocks_per_dev independent streams, blocks_per_dev of them in flight on each listed GPU of this node
    (one host thread per GPU inside the library, no traffic between the GPUs but the final gather onto devices[0])."""
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    n = int(src.size)
    nblocks = len(devices) * blocks_per_dev
    cap = int(lib.nlzm_hip_compress_bound(n)) + nblocks * (16 + 131072)
    dst = np.empty(cap, dtype=np.uint8)
    lens = (C.c_uint64 * nblocks)()
    out_len = C.c_uint64(0)
    devs = (C.c_int * len(devices))(*devices)
    _chk(lib.nlzm_hip_compress_blocks_multi(devs, len(devices), blocks_per_dev, src.ctypes.data if n else None, n, hist_bits,
                                            dst.ctypes.data, cap, lens, C.byref(out_len)))
    out, pos = [], 0
    for i in range(nblocks):
        out.append(dst[pos: pos + int(lens[i])].tobytes())
        pos += int(lens[i])
    assert pos == out_len.value
    return out


def stats() -> dict:
    s = Stats()
    _chk(load_library().nlzm_hip_get_stats(C.byref(s)))
    return s.as_dict()


def counter(key: str) -> int:
    """A diagnostic counter of the pipeline stages for the last stream (nlzm_hip_get_counter)."""
    v = C.c_uint64(0)
    _chk(load_library().nlzm_hip_get_counter(key.encode(), C.byref(v)))
    return int(v.value)


def timing() -> dict:
    t = Timing()
    _chk(load_library().nlzm_hip_get_timing(C.byref(t)))
    return t.as_dict()


def rans_frames(frames: list[tuple[np.ndarray, np.ndarray, int]]) -> list[bytes]:
    """CodeFrame::Flush replacement: frames = [(syms u32, bits u8 incl. pad, num_ops)]."""
    lib = load_library()
    nf = len(frames)
    sym_off = np.zeros(nf + 1, dtype=np.uint64)
    bit_off = np.zeros(nf + 1, dtype=np.uint64)
    for i, (s, b, _) in enumerate(frames):
        sym_off[i + 1] = sym_off[i] + len(s)
        bit_off[i + 1] = bit_off[i] + len(b)
    syms = np.concatenate([np.asarray(s, dtype=np.uint32) for s, _, _ in frames] + [np.zeros(1, np.uint32)])
    bits = np.concatenate([np.asarray(b, dtype=np.uint8) for _, b, _ in frames] + [np.zeros(1, np.uint8)])
    ops = np.array([o for _, _, o in frames], dtype=np.uint32)
    stride = int(max(12 + len(b) + 16 + 2 * len(s) for s, b, _ in frames)) + 16
    out = np.zeros(nf * stride, dtype=np.uint8)
    out_len = np.zeros(nf, dtype=np.uint32)
    _chk(lib.nlzm_hip_rans_frames(syms.ctypes.data, sym_off.ctypes.data, bits.ctypes.data, bit_off.ctypes.data,
                                  ops.ctypes.data, nf, out.ctypes.data, stride, out_len.ctypes.data))
    return [out[i * stride: i * stride + int(out_len[i])].tobytes() for i in range(nf)]


def find_matches(data, hist_bits: int, pos_lo: int, pos_hi: int, cap_words: int = 1 << 24) -> np.ndarray:
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    out = np.empty(cap_words, dtype=np.uint32)
    used = C.c_uint64(0)
    _chk(lib.nlzm_hip_find_matches(src.ctypes.data, src.size, hist_bits, pos_lo, pos_hi, out.ctypes.data, cap_words,
                                   C.byref(used)))
    return out[: used.value].copy()


def parse_emit(data, hist_bits: int, frame_idx: int):
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    syms = np.empty(1 << 19, dtype=np.uint32)
    bits = np.empty(1 << 18, dtype=np.uint8)
    sizes = np.zeros(3, dtype=np.uint32)
    _chk(lib.nlzm_hip_parse_emit(src.ctypes.data, src.size, hist_bits, frame_idx, syms.ctypes.data, syms.size,
                                 bits.ctypes.data, bits.size, sizes.ctypes.data))
    return syms[: sizes[0]].copy(), bits[: sizes[1]].copy(), int(sizes[2])
