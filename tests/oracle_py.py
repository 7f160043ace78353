"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "nlzm_ref")
REF_INSTR = os.path.join(ORACLE_DIR, "_ref", "nlzm_ref_instr")


class OStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "in_bytes", "out_bytes", "bt_calls", "bt_tests", "cmp_bytes", "ht_rows", "rk_probes", "rk_inserts",
        "positions", "nice_positions", "segments", "seg_rep_grow", "n_literal", "n_dict", "n_rep", "rans_syms",
        "bit_ops", "frames", "shifts", "cmp_bytes_needed")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(with_ref: bool = True) -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)
    if with_ref and os.path.exists("/root/reference/NLZM.cpp"):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(with_ref=False)
        L = C.CDLL(LIB)
        L.nlzm_oracle_bound.restype = C.c_uint64
        L.nlzm_oracle_bound.argtypes = [C.c_uint64]
        L.nlzm_oracle_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                           C.POINTER(C.c_uint64), C.POINTER(OStats), C.c_void_p]
        L.nlzm_oracle_compress_digest.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                                  C.POINTER(C.c_uint64), C.POINTER(OStats), C.c_void_p, C.c_void_p,
                                                  C.c_uint32]
        L.nlzm_oracle_decompress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.nlzm_oracle_crc32.restype = C.c_uint32
        L.nlzm_oracle_crc32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.nlzm_oracle_log2_lut.restype = C.POINTER(C.c_uint16)
        L.nlzm_oracle_match_min.restype = C.c_uint32
        L.nlzm_oracle_match_min.argtypes = [C.c_uint32]
        L.nlzm_oracle_hash4.restype = C.c_uint32
        L.nlzm_oracle_hash4.argtypes = [C.c_uint32]
        L.nlzm_oracle_rk_hash256.restype = C.c_uint32
        L.nlzm_oracle_rk_hash256.argtypes = [C.c_void_p]
        L.nlzm_oracle_cdf_run.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.nlzm_oracle_geometry.argtypes = [C.c_uint64, C.c_uint32] + [C.POINTER(C.c_uint32)] * 4
        L.nlzm_oracle_flush_frame.restype = C.c_uint32
        L.nlzm_oracle_flush_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_uint32]
        L.nlzm_oracle_capture_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                                C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.nlzm_oracle_capture_tables.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p,
                                                 C.c_uint64, C.POINTER(C.c_uint64)]
        L.nlzm_oracle_init()
        _lib = L
    return _lib


def _arr(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def compress(data, hist_bits: int = 22, want_stats: bool = False):
    L = lib()
    src = _arr(data)
    cap = int(L.nlzm_oracle_bound(src.size))
    dst = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    st = OStats()
    rc = L.nlzm_oracle_compress(src.ctypes.data, src.size, hist_bits, dst.ctypes.data, cap, C.byref(n), C.byref(st), None)
    if rc:
        raise RuntimeError(f"oracle compress failed: {rc}")
    out = dst[: n.value].tobytes()
    return (out, st.as_dict()) if want_stats else out


def compress_digest(data, hist_bits: int = 22):
    """-> (stream bytes, {'mt','seg','frm' hex, 'n_pos','n_seg','n_frames'})"""
    L = lib()
    src = _arr(data)
    cap = int(L.nlzm_oracle_bound(src.size))
    dst = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    st = OStats()
    d = np.zeros(6, dtype=np.uint64)
    rc = L.nlzm_oracle_compress_digest(src.ctypes.data, src.size, hist_bits, dst.ctypes.data, cap, C.byref(n),
                                       C.byref(st), d.ctypes.data, None, 0)
    if rc:
        raise RuntimeError(f"oracle compress failed: {rc}")
    dig = {"mt": f"{int(d[0]):016x}", "seg": f"{int(d[1]):016x}", "frm": f"{int(d[2]):016x}",
           "n_pos": int(d[3]), "n_seg": int(d[4]), "n_frames": int(d[5])}
    return dst[: n.value].tobytes(), dig


def decompress(stream) -> bytes:
    L = lib()
    src = _arr(stream)
    n = C.c_uint64(0)
    rc = L.nlzm_oracle_decompress(src.ctypes.data, src.size, None, 0, C.byref(n))
    if rc:
        raise RuntimeError(f"oracle decompress failed: {rc}")
    dst = np.empty(max(1, n.value), dtype=np.uint8)
    rc = L.nlzm_oracle_decompress(src.ctypes.data, src.size, dst.ctypes.data, n.value, C.byref(n))
    if rc:
        raise RuntimeError(f"oracle decompress failed: {rc}")
    return dst[: n.value].tobytes()


def crc32(data) -> int:
    a = _arr(data)
    return int(lib().nlzm_oracle_crc32(a.ctypes.data, a.size, 0))


def capture_frame(data, hist_bits: int, frame_idx: int):
    """-> (syms u32[], bits u8[] incl. pad, num_ops, frame bytes)"""
    L = lib()
    src = _arr(data)
    syms = np.empty(1 << 19, dtype=np.uint32)
    bits = np.empty(1 << 18, dtype=np.uint8)
    frame = np.empty(1 << 18, dtype=np.uint8)
    sizes = np.zeros(4, dtype=np.uint32)
    rc = L.nlzm_oracle_capture_frame(src.ctypes.data, src.size, hist_bits, frame_idx, syms.ctypes.data, syms.size,
                                     bits.ctypes.data, bits.size, frame.ctypes.data, frame.size, sizes.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle capture_frame failed: {rc}")
    return syms[: sizes[0]].copy(), bits[: sizes[1]].copy(), int(sizes[3]), frame[: sizes[2]].tobytes()


def capture_tables(data, hist_bits: int, pos_lo: int, pos_hi: int, cap_words: int = 1 << 24) -> np.ndarray:
    L = lib()
    src = _arr(data)
    out = np.empty(cap_words, dtype=np.uint32)
    used = C.c_uint64(0)
    rc = L.nlzm_oracle_capture_tables(src.ctypes.data, src.size, hist_bits, pos_lo, pos_hi, out.ctypes.data, cap_words,
                                      C.byref(used))
    if rc:
        raise RuntimeError(f"oracle capture_tables failed: {rc}")
    return out[: used.value].copy()


def flush_frame(syms: np.ndarray, bits: np.ndarray, num_ops: int) -> bytes:
    L = lib()
    s = np.ascontiguousarray(syms, dtype=np.uint32)
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.empty(12 + b.size + 16 + 2 * s.size + 64, dtype=np.uint8)
    n = L.nlzm_oracle_flush_frame(s.ctypes.data, s.size, b.ctypes.data, b.size, num_ops, out.ctypes.data, out.size)
    if not n:
        raise RuntimeError("oracle flush_frame failed")
    return out[:n].tobytes()
