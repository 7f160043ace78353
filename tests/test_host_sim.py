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


@pytest.mark.parametrize("name", ["overlap_265", "text_200k_w15", "dups_400k_w16", "runs_300k_w18", "random_100k_w15"])
def test_master_logic_matches_oracle(sim, name, tmp_path):
    case = next(c for c in cases.CASES if c[0] == name)
    p = tmp_path / "in.bin"
    cases.make_case(case).tofile(p)
    r = subprocess.run([sim, str(p), str(case[4]), "1"], capture_output=True, text=True)
    assert r.returncode == 0 and ": OK" in r.stdout, r.stdout + r.stderr
