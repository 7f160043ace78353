"""CPU suite: the C-ABI library loads and exports exactly what include/nlzm_hip.h declares;
without a GPU every compute entry fails loudly (there is no CPU fallback)."""
import os
import re
import subprocess

import pytest

import nlzm_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    nlzm_amd.build()
    return nlzm_amd.load_library()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nlzm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nlzm_hip_\w+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert syms == sorted(nlzm_amd.ABI_SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nlzm_hip.h but not exported"


def test_only_c_abi_is_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", nlzm_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(not s.startswith("nlzm_oracle") for s in exported), "the product must not contain the oracle"
    assert set(nlzm_amd.ABI_SYMBOLS) <= set(exported)


def test_library_has_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={nlzm_amd.LIB_PATH}"], capture_output=True, text=True)
    blob = open(nlzm_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"pipeline2_kernel" in blob and b"rans_frames_kernel" in blob and b"prefilter_hash_kernel" in blob


def test_persistent_kernel_has_no_scratch_flat_or_calls():
    """Any of them puts HBM latency into the serial half: checked in the gfx950 ISA."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "nlzm_amd", "csrc"), "asmcheck"], capture_output=True, text=True)
    assert r.returncode == 0 and "asmcheck: ok" in r.stdout, r.stdout + r.stderr


def test_geometry_needs_no_device(lib):
    g = nlzm_amd.geometry(1_000_000_000, 28)
    assert g == {"hist_bits": 28, "frame_bits": 17, "chunk_size": 122368, "feed_size": 122633}
    assert nlzm_amd.geometry(125_000_000, 28)["hist_bits"] == 27
    assert nlzm_amd.geometry(0, 22)["hist_bits"] == 10
    assert lib.nlzm_hip_compress_bound(0) >= 8


def test_block_mode_placement_rule(lib):
    """The persistent launch of a block set deals every stream's workgroups (three stages + its worker CUs) to ONE XCD when the
    number of streams is a multiple of eight (workgroups go to the XCDs round-robin: equal index modulo 8), and in every case
    each (stream, block) pair is some workgroup exactly once.  No device needed."""
    import ctypes as C
    f = lib.nlzm_hip_block_placement
    f.restype = None
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for nstreams in (1, 3, 8, 16, 24, 32, 40, 64):
        for bps in (4, 5, 7, 8, 31):
            seen = {}
            for wg in range(nstreams * bps):
                s, l = C.c_uint32(), C.c_uint32()
                f(nstreams, bps, wg, C.byref(s), C.byref(l))
                assert s.value < nstreams and l.value < bps
                assert (s.value, l.value) not in seen
                seen[(s.value, l.value)] = wg
            assert len(seen) == nstreams * bps
            if nstreams % 8 == 0:
                for st in range(nstreams):
                    assert len({seen[(st, l)] % 8 for l in range(bps)}) == 1, (nstreams, bps, st)


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.nlzm_hip_init(0) != 0
    assert b"device" in lib.nlzm_hip_last_error().lower()
    with pytest.raises(nlzm_amd.NlzmError):
        nlzm_amd.compress(b"hello world", 15)


def test_cli_host_side_decodes_and_refuses_to_overwrite(tmp_path, lib):
    """`d`/`t`/`h` are host-only (NLZM.cpp:2119-2162); `c` needs the GPU."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "streams.json")))["cases"]
    g = next(c for c in gold if c["name"] == "tiny_1000" )
    # hist_bits < 12 here: the reference's own decoder would assert (NLZM.cpp:1918); ours accepts it
    from tests import cases
    data = cases.make_case(next(c for c in cases.CASES if c[0] == "tiny_1000")).tobytes()
    stream = tmp_path / "s.nlzm"
    stream.write_bytes(bytes.fromhex(g["stream_hex"]))
    out = tmp_path / "o.bin"
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(stream), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and out.read_bytes() == data
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(stream), str(out)], capture_output=True, text=True)
    assert r.returncode != 0 and "already exists" in r.stdout
    r = subprocess.run([nlzm_amd.CLI_PATH, "t", str(stream)], capture_output=True, text=True)
    assert "Done (output CRC32" in r.stdout
    r = subprocess.run([nlzm_amd.CLI_PATH, "h", str(out)], capture_output=True, text=True)
    from tests import oracle_py
    assert r.stdout.strip().splitlines()[-1] == f"{oracle_py.crc32(data):X}"
    r = subprocess.run([nlzm_amd.CLI_PATH, "-bogus", "c", str(out), str(tmp_path / 'x')], capture_output=True, text=True)
    assert r.returncode != 0 and "Unrecognized flag" in r.stdout
    r = subprocess.run([nlzm_amd.CLI_PATH], capture_output=True, text=True)
    assert "Commands:" in r.stdout


def test_cli_decodes_streams_back_to_back(tmp_path, lib):
    """Block mode's artifact: k independent streams, concatenated; `d`/`t` find the boundaries by
    hopping over the frames and decode the blocks on host threads."""
    import numpy as np
    from nlzm_amd import corpus, shard
    from tests import oracle_py
    data = corpus.mixed(700_000, corpus.SEED + 9)
    k = 5
    streams = [oracle_py.compress(data[slice(*shard.block_range(data.size, k, i))], 18) for i in range(k)]
    f = tmp_path / "blocks.nlzm"
    f.write_bytes(b"".join(streams))
    out = tmp_path / "o.bin"
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(f), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and f"Blocks: {k}" in r.stdout, r.stdout
    assert out.read_bytes() == data.tobytes()
    # a container cut off inside its last block: the complete blocks are decoded and written, the exit status says it was cut
    f.write_bytes(b"".join(streams)[:-3])
    out2 = tmp_path / "o2.bin"
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(f), str(out2)], capture_output=True, text=True)
    assert r.returncode not in (0, 255) and "cut off inside block 5" in r.stdout and f"Blocks: {k - 1}" in r.stdout, r.stdout
    lo4, _ = shard.block_range(data.size, k, k - 1)
    assert out2.read_bytes() == data.tobytes()[:lo4]
    # a valid stream followed by bytes that are no stream header: decoded as the reference would (it stops at the first
    # terminator, NLZM.cpp:646-648), with a note
    f.write_bytes(streams[0] + b"\x00\x13\x07garbage")
    r = subprocess.run([nlzm_amd.CLI_PATH, "t", str(f)], capture_output=True, text=True)
    assert r.returncode == 0 and "ignored" in r.stdout, r.stdout
    # cut inside the FIRST stream: nothing to decode
    f.write_bytes(streams[0][:-9])
    r = subprocess.run([nlzm_amd.CLI_PATH, "t", str(f)], capture_output=True, text=True)
    assert r.returncode != 0 and "malformed" in r.stdout



def test_cli_block_index_sidecar(tmp_path, lib):
    """The block container's index: `c -blocks:k` writes <output>.idx -- every block's offset, stream length and input length --
    and `d` / `t` take the boundaries from it when it fits the file: no hop over every frame header before the parallel decode starts, and a
    container that is DAMAGED INSIDE an early block still gives up the later ones' boundaries... here: cut off inside block 3 of 5, blocks 1 and 2
    come out and the exit status says so; an index that does not fit the file is ignored with a note."""
    from nlzm_amd import corpus, shard
    from tests import oracle_py
    data = corpus.mixed(600_000, corpus.SEED + 10)
    k = 5
    ranges = [shard.block_range(data.size, k, i) for i in range(k)]
    streams = [oracle_py.compress(data[lo:hi], 18) for lo, hi in ranges]
    blob = b"".join(streams)
    f, idx, out = tmp_path / "c.nlzm", tmp_path / "c.nlzm.idx", tmp_path / "o.bin"

    def write_index(total):
        off, lines = 0, [f"NLZMIDX 1 {k} {data.size} {total}"]
        for s_, (lo, hi) in zip(streams, ranges):
            lines.append(f"{off} {len(s_)} {hi - lo}")
            off += len(s_)
        idx.write_text("\n".join(lines) + "\n")

    f.write_bytes(blob); write_index(len(blob))
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(f), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and f"Blocks: {k}" in r.stdout and "does not fit" not in r.stdout, r.stdout
    assert out.read_bytes() == data.tobytes()
    # cut inside block 3: the index still says where blocks 1 and 2 are
    cut = len(streams[0]) + len(streams[1]) + len(streams[2]) // 2
    f.write_bytes(blob[:cut])
    out2 = tmp_path / "o2.bin"
    r = subprocess.run([nlzm_amd.CLI_PATH, "d", str(f), str(out2)], capture_output=True, text=True)
    assert r.returncode not in (0, 255) and "cut off inside block 3" in r.stdout and "Blocks: 2" in r.stdout, r.stdout
    assert out2.read_bytes() == data.tobytes()[:ranges[2][0]]
    # an index of another file: ignored, the frame headers are hopped over as before
    f.write_bytes(blob); write_index(len(blob) - 7)
    r = subprocess.run([nlzm_amd.CLI_PATH, "t", str(f)], capture_output=True, text=True)
    assert r.returncode == 0 and "does not fit" in r.stdout and f"Blocks: {k}" in r.stdout, r.stdout
