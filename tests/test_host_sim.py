"""CPU suite: the product's pipeline stages (nlzm_amd/csrc/nlzm_v2.h: finder, table, parser), compiled for the host with
every GPU lane a fiber (tests/host_sim), must reproduce the oracle's per-position match tables, per-frame symbol / bit
streams and operation counters exactly.

The simulator runs 5-8 KB/s on one core, so the module's runs (one process each, ~0.8 GB) are started together when its first
test asks for one -- a few beside each other, the longest first -- and every test takes the result of its own run."""
import ctypes
import hashlib
import json
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from nlzm_amd import corpus
from tests import cases

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "host_sim", "sim2")
WORKERS = max(1, min(6, (os.cpu_count() or 2) - 2))

# key -> (input maker (None: this machine cannot make it), simulator arguments behind the file, environment, timeout in s); in the order the runs are started
JOBS = {}


def job(key, make, args, env=None, timeout=900):
    assert key not in JOBS
    JOBS[key] = (make, [str(a) for a in args], env or {}, timeout)
    return key


def case_of(name):
    return next(c for c in cases.CASES if c[0] == name)


def case_input(name, upto=None):
    return lambda: cases.make_case(case_of(name))[:upto]


# ---- the runs, longest first -----------------------------------------------------------------------------------------------
# arguments: window bits, worker emulation (1 = worker lanes as slow as possible: a head's lane runs only when the finder stage asks
# for one of its positions; 2 = infinitely fast: each runs until it blocks on a decision of the finder stage), the number of persistent
# launches the input is cut into (state save / restore between them).  The small cases take both emulations, the large ones one each.
U16 = ["u16cut_734k_w24", "u16tie_734k_w24"]
for _n in U16:
    job(("u16", _n), case_input(_n), [case_of(_n)[4], 2, 3])
HELPER = [("dense_150k_w17", {}, 1), ("dense_150k_w17", {"NLZM_SIM_RANDOM_BLOCKS": "7", "NLZM_SIM_POISON": "3"}, 3), ("denseb_250k_w18", {"NLZM_SIM_POISON": "5"}, 2)]
for _n, _env, _l in HELPER:
    # (the variant with random blocks: 24 forced cuts are enough)
    job(("helper", _n, tuple(sorted(_env.items())), _l), case_input(_n, 100_000 if _env and _n == "dense_150k_w17" else None), [case_of(_n)[4], 2, _l], _env, 1500)
    if not _env:    # ... and without the helper the same streams (once per input: the plain variant)
        job(("helper_off", _n, _l), case_input(_n), [case_of(_n)[4], 2, _l], {"NLZM_SIM_HELPER": "0"}, 1500)
job(("dups", "dups_400k_w16"), case_input("dups_400k_w16", 300_000), [case_of("dups_400k_w16")[4], 2, 2], timeout=1500)      # (W = 65,536: four rebases)


def real_slice():
    gold = next(c for c in json.load(open(os.path.join(HERE, "golden", "real.json")))["cases"] if c["name"] == "real_30m_w24")
    try:
        data = corpus.make("real_text", 30_000_000)
    except RuntimeError:
        return None
    if hashlib.sha256(data.tobytes()).hexdigest() != gold["input_sha256"]:
        return None
    return data[20_000_000:20_300_000]


job("real_shape", real_slice, [20, 2, 3], {"NLZM_SIM_TABLE_SHAPE": "0"})
STAGES = [(n, w, l) for n in ("one_byte", "tiny_1000", "chunk_plus1", "overlap_265") for w, l in ((1, 1), (2, 3))] \
    + [("text_200k_w15", 2, 3), ("runs_300k_w18", 1, 1), ("random_100k_w15", 2, 3)]
for _n, _w, _l in sorted(STAGES, key=lambda t: -case_of(t[0])[2]):
    job(("stages", _n, _w, _l), case_input(_n, 170_000), [case_of(_n)[4], _w, _l])     # (the two 300 KB cases give the simulator their first 170 KB)
CUTS = [{"NLZM_SIM_RANDOM_BLOCKS": "7", "NLZM_SIM_POISON": "3"}, {"NLZM_SIM_PSTRIDE": "5", "NLZM_SIM_POISON": "4"}]
for _env in CUTS:
    job(("cuts", tuple(sorted(_env.items()))), case_input("chains_150k_w17"), [case_of("chains_150k_w17")[4], 1, 2], _env)
SPLICE = [(4242, 2, 3, {"NLZM_SIM_PSTRIDE": "9"}), (777, 2, 2, {"NLZM_SIM_RANDOM_BLOCKS": "5", "NLZM_SIM_POISON": "2"}), (31337, 1, 2, {})]
for _seed, _w, _l, _env in SPLICE:
    job(("splice", _seed), (lambda sd: lambda: corpus.make("splice", 130_000, sd))(_seed), [19, _w, _l], _env)


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    subprocess.run(["make", "-C", os.path.join(HERE, "host_sim")], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("sim_inputs")
    ex = ThreadPoolExecutor(WORKERS)
    futs = {}
    for n, (key, (make, args, env, timeout)) in enumerate(JOBS.items()):
        data = make()
        if data is None:
            futs[key] = None
            continue
        p = d / f"{n}.bin"
        data.tofile(p)
        futs[key] = ex.submit(subprocess.run, [SIM, str(p)] + args, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env))
    yield futs
    ex.shutdown(wait=False, cancel_futures=True)


def ok(runs, key):
    f = runs[key]
    if f is None:
        pytest.skip("this machine's files are not the ones the fixture was made from")
    r = f.result()
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return r


@pytest.mark.parametrize("name,workers,launches", STAGES)
def test_stages_match_oracle(runs, name, workers, launches):
    ok(runs, ("stages", name, workers, launches))


@pytest.mark.parametrize("name", ["dups_400k_w16"])
def test_stages_match_oracle_rebased_duplicates(runs, name):
    """window rebases, the p >= W masking regime of HT/RK, carried RK256 matches across a rebase"""
    ok(runs, ("dups", name))


@pytest.mark.parametrize("name", U16)
def test_rk_entry_cut_short_by_uint16_grows_again(runs, name):
    """An RK256 match found where `la_end - p` is just above 65,536 is compared over (uint16)(la_end - p) bytes only (NLZM.cpp:760, :1096);
    when that entry is the table's longest, the next position extends it again (:1503-1512).  The finder stage held such an entry for a
    closed one until round 5 (real text meets it once in ~100 MB).  The cases must contain the situation: the oracle
    counts it, and the stage says what it did -- in the second case the entry ends exactly where another entry ends, and grows only where
    it is the nearer of the two (:835-852)."""
    from tests import oracle_py
    case = case_of(name)
    L = oracle_py.lib()
    L.nlzm_oracle_debug_rk_u16_cuts.restype = ctypes.c_uint64
    n0 = L.nlzm_oracle_debug_rk_u16_cuts()
    oracle_py.compress(cases.make_case(case), case[4])
    assert L.nlzm_oracle_debug_rk_u16_cuts() - n0 >= 2, "the case no longer contains a cut-short RK256 entry that is the table's longest"
    r = ok(runs, ("u16", name))
    grown, ties, won = map(int, re.findall(r"\d+", next(l for l in r.stdout.splitlines() if l.startswith("cut-short RK256 entries:")).split(":", 1)[1]))
    assert (grown >= 1 and ties >= 3 and 1 <= won < ties) if "tie" in name else grown >= 2, r.stdout[-600:]


@pytest.mark.parametrize("seed", [t[0] for t in SPLICE])
def test_spliced_fuzz_input(runs, seed):
    """Three seeds of the fuzz that tests/sim_fuzz.py runs in batches (corpus.splice: pieces of every kind the finders treat differently, spliced at random), so that the
    generator and the harness stay alive in the suite: one with nine pairs reserved per position and the rest in the launch's arena, one with the parser's blocks cut at
    random over poisoned buffers, one with the slowest worker lanes."""
    ok(runs, ("splice", seed))


def test_table_stage_changes_shape_on_real_text(runs):
    """The table stage runs a launch with 16-entry fronts on seven waves or with 24-entry fronts on five, as the launch before it asked (nlzm_v2.h, TLds): source code
    has positions with more BT4 record-setters than 16 in one block out of ten, the stand-ins in none.  300 KB of the image's own headers (corpus.real_text) in three
    launches must go wide and stay exact.  (The forced shapes: tests/test_gpu_parity.py.)"""
    r = ok(runs, "real_shape")
    line = next(l for l in r.stdout.splitlines() if l.startswith("table shape:"))
    wide, changed = int(line.split()[2]), int(line.split()[6])
    assert wide >= 2 and changed >= 1, line


@pytest.mark.parametrize("env", CUTS)
def test_forced_cuts_and_block_sizes(runs, env):
    """Segments that run into the forced cut at 4,096 positions or end just before it (records re-listed for the cut, and
    put back when the segment ends early); with the parser's blocks cut at random and every buffer the kernel does not
    initialise itself full of junk, as both are on the device.  Second variant: five pairs reserved per position in bt_pairs and the
    rest of a position's BT4 pairs in extension blocks of the launch's arena (the layout of a block set's streams, nlzm_core.h)."""
    r = ok(runs, ("cuts", tuple(sorted(env.items()))))
    assert "put back 0)" not in r.stdout, "the case no longer exercises the put-back path"


@pytest.mark.parametrize("name,env,launches", HELPER)
def test_helper_parser_takes_segments_over(runs, name, env, launches):
    """The helper parser (nlzm_v2.h HelpBox): segments that are cut at 4,096 positions get a second workgroup that
    parses their back from node 1,792; the parser stage compares the 264-node frontier and takes the helper's nodes over -- streams,
    tables and every counter must still be the oracle's, and the case must exercise the take-over (and, with breaks, the drop / give-up
    paths: segments that end inside the re-listed zone in front of a forced cut)."""
    r = ok(runs, ("helper", name, tuple(sorted(env.items())), launches))
    line = next(l for l in r.stdout.splitlines() if l.startswith("helper parser:"))
    posted, taken = int(line.split()[2]), int(line.split()[5])
    assert posted >= 5 and taken >= 3, line
    if not env:
        ok(runs, ("helper_off", name, launches))
