"""CPU suite: the product's pipeline stages (nlzm_amd/csrc/nlzm_v2.h: finder, table, parser), compiled for the host with
every GPU lane a fiber (tests/host_sim), must reproduce the oracle's per-position match tables, per-frame symbol / bit
streams and operation counters exactly."""
import os
import subprocess

import pytest

from nlzm_amd import corpus
from tests import cases

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "host_sim", "sim2")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "host_sim")], check=True, capture_output=True)
    return SIM


# worker emulation: 1 = worker lanes as slow as possible (a head's lane runs only when the finder stage asks for one of
# its positions), 2 = worker lanes infinitely fast (each runs until it blocks on a decision of the finder stage);
# launches: how many persistent launches the input is cut into (state save / restore between them)
# (the simulator runs 5-8 KB/s: the small cases take both emulations, the large ones one each)
@pytest.mark.parametrize("name,workers,launches",
                         [(n, w, l) for n in ("one_byte", "tiny_1000", "chunk_plus1", "overlap_265") for w, l in ((1, 1), (2, 3))]
                         + [("text_200k_w15", 2, 3), ("runs_300k_w18", 1, 1), ("random_100k_w15", 2, 3)])
def test_stages_match_oracle(sim, name, workers, launches, tmp_path):
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    cases.make_case(case)[:170_000].tofile(p)       # (the simulator runs 5-8 KB/s: the two 300 KB cases give it their first 170 KB)
    r = subprocess.run([sim, str(p), str(case[4]), str(workers), str(launches)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["dups_400k_w16"])
def test_stages_match_oracle_rebased_duplicates(sim, name, tmp_path):
    """window rebases, the p >= W masking regime of HT/RK, carried RK256 matches across a rebase"""
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    cases.make_case(case)[:300_000].tofile(p)       # (W = 65,536: four rebases)
    r = subprocess.run([sim, str(p), str(case[4]), "2", "2"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["u16cut_734k_w24", "u16tie_734k_w24"])
def test_rk_entry_cut_short_by_uint16_grows_again(sim, name, tmp_path):
    """An RK256 match found where `la_end - p` is just above 65,536 is compared over (uint16)(la_end - p) bytes only (NLZM.cpp:760, :1096);
    when that entry is the table's longest, the next position extends it again (:1503-1512).  The finder stage held such an entry for a
    closed one until round 5 (real text meets it once in ~100 MB).  The cases must contain the situation: the oracle
    counts it, and the stage says what it did -- in the second case the entry ends exactly where another entry ends, and grows only where
    it is the nearer of the two (:835-852)."""
    import ctypes
    from tests import oracle_py
    case = next(c for c in cases.CASES if c[0] == name)
    data = cases.make_case(case)
    L = oracle_py.lib()
    L.nlzm_oracle_debug_rk_u16_cuts.restype = ctypes.c_uint64
    n0 = L.nlzm_oracle_debug_rk_u16_cuts()
    oracle_py.compress(data, case[4])
    assert L.nlzm_oracle_debug_rk_u16_cuts() - n0 >= 2, "the case no longer contains a cut-short RK256 entry that is the table's longest"
    p = tmp_path / "in.bin"
    data.tofile(p)
    r = subprocess.run([sim, str(p), str(case[4]), "2", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    import re
    grown, ties, won = map(int, re.findall(r"\d+", next(l for l in r.stdout.splitlines() if l.startswith("cut-short RK256 entries:")).split(":", 1)[1]))
    assert (grown >= 1 and ties >= 3 and 1 <= won < ties) if "tie" in name else grown >= 2, r.stdout[-600:]


def test_spliced_fuzz_input(sim, tmp_path):
    """One seed of the fuzz that tests/sim_fuzz.py runs in batches (corpus.splice: pieces of every kind the finders treat differently, spliced at random), so that the
    generator and the harness stay alive in the suite."""
    p = tmp_path / "in.bin"
    corpus.make("splice", 130_000, 4242).tofile(p)
    r = subprocess.run([sim, str(p), "19", "2", "3"], capture_output=True, text=True, timeout=900, env=dict(os.environ, NLZM_SIM_PSTRIDE="9"))
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_table_stage_changes_shape_on_real_text(sim, tmp_path):
    """The table stage runs a launch with 16-entry fronts on seven waves or with 24-entry fronts on five, as the launch before it asked (nlzm_v2.h, TLds): source code
    has positions with more BT4 record-setters than 16 in one block out of ten, the stand-ins in none.  300 KB of the image's own headers (corpus.real_text) in three
    launches must go wide and stay exact."""
    import hashlib
    import json
    gold = next(c for c in json.load(open(os.path.join(HERE, "golden", "real.json")))["cases"] if c["name"] == "real_30m_w24")
    data = corpus.make("real_text", 30_000_000)
    if hashlib.sha256(data.tobytes()).hexdigest() != gold["input_sha256"]:
        pytest.skip("this machine's files are not the ones the fixture was made from")
    p = tmp_path / "in.bin"
    data[20_000_000:20_300_000].tofile(p)
    for shape, want_wide in (("0", True),):        # (the forced shapes: tests/test_gpu_parity.py)
        r = subprocess.run([sim, str(p), "20", "2", "3"], capture_output=True, text=True, timeout=900, env=dict(os.environ, NLZM_SIM_TABLE_SHAPE=shape))
        assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        line = next(l for l in r.stdout.splitlines() if l.startswith("table shape:"))
        wide, changed = int(line.split()[2]), int(line.split()[6])
        assert (wide >= 2) == want_wide and (shape != "0" or changed >= 1), line


@pytest.mark.parametrize("env", [{"NLZM_SIM_RANDOM_BLOCKS": "7", "NLZM_SIM_POISON": "3"}, {"NLZM_SIM_PSTRIDE": "5", "NLZM_SIM_POISON": "4"}])
def test_forced_cuts_and_block_sizes(sim, env, tmp_path):
    """Segments that run into the forced cut at 4,096 positions or end just before it (records re-listed for the cut, and
    put back when the segment ends early); with the parser's blocks cut at random and every buffer the kernel does not
    initialise itself full of junk, as both are on the device.  Second variant: five pairs reserved per position in bt_pairs and the
    rest of a position's BT4 pairs in extension blocks of the launch's arena (the layout of a block set's streams, nlzm_core.h)."""
    case = next(c for c in cases.CASES if c[0] == "chains_150k_w17")
    p = tmp_path / "in.bin"
    cases.make_case(case).tofile(p)
    r = subprocess.run([sim, str(p), str(case[4]), "1", "2"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "put back 0)" not in r.stdout, "the case no longer exercises the put-back path"


@pytest.mark.parametrize("name,env,launches", [("dense_150k_w17", {}, 1), ("dense_150k_w17", {"NLZM_SIM_RANDOM_BLOCKS": "7", "NLZM_SIM_POISON": "3"}, 3),
                                               ("denseb_250k_w18", {"NLZM_SIM_POISON": "5"}, 2)])
def test_helper_parser_takes_segments_over(sim, name, env, launches, tmp_path):
    """The helper parser (nlzm_v2.h HelpBox): segments that are cut at 4,096 positions get a second workgroup that
    parses their back from node 1,792; the parser stage compares the 264-node frontier and takes the helper's nodes over -- streams,
    tables and every counter must still be the oracle's, and the case must exercise the take-over (and, with breaks, the drop / give-up
    paths: segments that end inside the re-listed zone in front of a forced cut)."""
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    data = cases.make_case(case)
    (data[:100_000] if env and name == "dense_150k_w17" else data).tofile(p)       # (the variant with random blocks: 24 forced cuts are enough)
    r = subprocess.run([sim, str(p), str(case[4]), "2", str(launches)], capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env))
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    line = next(l for l in r.stdout.splitlines() if l.startswith("helper parser:"))
    posted, taken = int(line.split()[2]), int(line.split()[5])
    assert posted >= 5 and taken >= 3, line
    # ... and without the helper the same streams (once per input: the plain variant)
    if env:
        return
    r0 = subprocess.run([sim, str(p), str(case[4]), "2", str(launches)], capture_output=True, text=True, timeout=1500,
                        env=dict(os.environ, NLZM_SIM_HELPER="0", **env))
    assert r0.returncode == 0 and ": OK" in r0.stdout, r0.stdout[-2000:] + r0.stderr[-2000:]
