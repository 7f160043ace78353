"""CPU suite: the product's master logic (nlzm_amd/csrc/nlzm_core.h), compiled for the host
with a 1-lane wave policy (tests/host_sim), must reproduce the oracle's per-position match
tables and per-frame symbol/bit streams exactly."""
import os
import subprocess

import pytest

from tests import cases

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "host_sim", "sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "host_sim")], check=True, capture_output=True)
    return SIM


# worker emulation: 0 = BT4 inside the master, 1 = worker lanes as slow as possible (results computed only when
# the master asks), 2 = worker lanes infinitely fast (each head runs until it blocks on a decision of the master)
@pytest.mark.parametrize("workers", [0, 1, 2])
@pytest.mark.parametrize("name", ["overlap_265", "text_200k_w15", "dups_400k_w16", "runs_300k_w18", "random_100k_w15",
                                  "text_2m_w15"])
def test_master_logic_matches_oracle(sim, name, workers, tmp_path):
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    cases.make_case(case).tofile(p)
    # (the simulator's waits have no timeout of their own: NLZM_SIM_WATCH dumps the hand-off words and exits)
    r = subprocess.run([sim, str(p), str(case[4]), "1", str(workers)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, NLZM_SIM_WATCH="300"))
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name", ["text_200k_w15", "dups_400k_w16", "runs_300k_w18"])
def test_master_logic_with_late_worker_results(sim, name, tmp_path):
    """Same, with every 8th BT4 result not in when the look-ahead reads it: the refresh and wait paths."""
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    cases.make_case(case).tofile(p)
    r = subprocess.run([SIM + "_late", str(p), str(case[4]), "1", "1"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, NLZM_SIM_WATCH="300"))
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout + r.stderr

