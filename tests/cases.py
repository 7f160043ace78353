"""Parity cases shared by the golden-vector generator (oracle/make_golden.py) and the tests.

Each case is (name, generator kind, size, seed offset, window bits as given to the CLI).
The edge cases are the ones SURVEY.md section 8c lists: empty input, < 2 KiB input (window
shrinks below the decoder's minimum), exactly chunk_size +-1, EOF inside the 265-byte
overlap, input > 2W (window rebases and the p >= W masking regime of HT/RK), long
duplicated spans (RK carry, uint16 truncation -- and, corpus.u16_cut, a truncated entry that grows again), long runs (nice-length skip, BT4 early
return), random bytes (expansion), segments that reach the forced cut at 4,096 positions or end
just before it (the sampled lengths near the cut depend on the segment the position ends up in), a nice region that
starts exactly at such a cut (round 2's block-mode stall: the finder stage waited for the parser's word on the segment,
the parser for the finder's record of the segment's first position).
"""
from __future__ import annotations

import numpy as np

from nlzm_amd import corpus

# chunk_size for hist_bits <= 16 is 14,848 (NLZM.cpp:1722-1724)
CASES = [
    ("empty", "random", 0, 0, 22),
    ("one_byte", "syn_text", 1, 0, 15),
    ("tiny_1000", "syn_text", 1000, 1, 15),
    ("under_2k", "syn_text", 2047, 2, 20),
    ("chunk_minus1", "syn_text", 14847, 3, 15),
    ("chunk_exact", "syn_text", 14848, 3, 15),
    ("chunk_plus1", "syn_text", 14849, 3, 15),
    ("overlap_264", "syn_text", 14848 + 264, 4, 15),
    ("overlap_265", "syn_text", 14848 + 265, 4, 15),
    ("overlap_266", "syn_text", 14848 + 266, 4, 15),
    ("text_200k_w15", "syn_text", 200_000, 5, 15),
    ("text_300k_w20", "syn_text", 300_000, 0, 20),
    ("text_500k_w17", "syn_text", 500_000, 6, 17),
    ("mixed_1m_w20", "mixed", 1_000_000, 7, 20),
    ("dups_600k_w20", "dups", 600_000, 8, 20),
    ("dups_400k_w16", "dups", 400_000, 9, 16),
    ("runs_300k_w15", "runs", 300_000, 10, 15),
    ("runs_300k_w18", "runs", 300_000, 11, 18),
    ("random_100k_w15", "random", 100_000, 12, 15),
    ("text_2m_w15", "syn_text", 2_000_000, 13, 15),
    ("chains_150k_w17", "chains", 150_000, 14, 17),
    ("cutnice_60k_w17", "cutnice", 60_000, 15, 17),
    # the helper parser's cases: nearly every segment cut at 4,096 positions, as on deep text; and segments of
    # every length behind cut ones, some ending inside the 264 + 64 positions in front of a forced cut
    ("dense_150k_w17", "dense_text", 150_000, 16, 17),
    ("denseb_250k_w18", "dense_breaks", 250_000, 17, 18),
    # markup shaped like the wiki dumps the reference's README quotes (enwik8 / enwik9): fixed tags at slowly varying distances, digit runs
    ("xml_400k_w19", "xml_like", 400_000, 20, 19),
    # RK256 matches whose compare the uint16 length parameter ends and that the next position extends again (NLZM.cpp:760, :1503-1512;
    # corpus.u16_cut): the bug real text exposed in round 5, two or three of them here
    ("u16cut_734k_w24", "u16_cut", 6 * 122_368 + 100, 22, 24),
    # ... and such an entry ending exactly where another entry ends: the nearer of the two is what the table's end holds (corpus.u16_tie)
    ("u16tie_734k_w24", "u16_tie", 6 * 122_368 + 100, 24, 24),
    # positions with 36 - 120 BT4 record-setters of their own (corpus.spines): more than a block set reserves per position -- the pair lists' extension
    # arena, and (GPU suite, with a small arena) the stream that is made again when a launch uses it up
    ("spines_400k_w18", "spines", 400_000, 26, 18),
]

# larger cases: checked on the GPU box against the oracle run live (and golden sha)
BIG_CASES = [
    ("text_3m_w20", "syn_text", 3_000_000, 0, 20),
    ("mixed_3m_w20", "mixed", 3_152_896, 0, 20),
    ("dense_3m_w20", "dense_text", 3_000_000, 18, 20),
    ("denseb_6m_w22", "dense_breaks", 6_000_000, 19, 22),
    ("xml_5m_w23", "xml_like", 5_000_000, 21, 23),
    ("u16cut_4m_w24", "u16_cut", 35 * 122_368 + 5000, 23, 24),
]


# the full-size BASELINE configurations (SURVEY.md appendix D.2 regimes), reference SHA in tests/golden/full.json
# (oracle/make_golden_full.py):
#   cfg 3  100,000,000 B at -window:26: no rebase, the last ~33 MB in the p >= W masking regime of HT/RK
#   cfg 5  one 125,000,000-B block at -window:28: the header must read 27 (NLZM.cpp:1716-1718), every position < W
#   cfg 4  1,000,000,000 B at -window:28: rebases near 537 MB and 805 MB
#   cfg 1  100,000,000 B at -window:24: five rebases with 128 KiB frames (NLZM.cpp:1786-1792), most of the file in the
#          p >= W regime
FULL_CASES = [
    ("text_100m_w24", "syn_text", 100_000_000, 0, 24),
    ("text_100m_w26", "syn_text", 100_000_000, 0, 26),
    ("block_125m_w28", "syn_text", 125_000_000, 1, 28),
    ("text_1g_w28", "syn_text", 1_000_000_000, 0, 28),
]


_last_made = [None, None]          # (two consecutive cases of the full-size suite are the same 100 MB under two window sizes: generated once)


def make_case(case) -> np.ndarray:
    _, kind, size, seed_off, _ = case
    key = (kind, size, seed_off)
    if _last_made[0] != key:
        _last_made[0], _last_made[1] = None, None
        _last_made[1] = corpus.make(kind, size, corpus.SEED + seed_off)
        _last_made[0] = key
    return _last_made[1]


# bench.py's block-mode leg in small (its geometry: 32 streams in flight, -window:28 -> 25, more than 130 chunks per block):
# every block against the reference run on that block alone (tests/golden/blocks.json, oracle/make_golden_blocks.py)
BLOCK_SET = ("blocks32_w28", "block_set", 32 * 17_000_000, 0, 28, 32)
# the leg itself: the 1e9-byte stand-in of config 4 split into 32 blocks (tests/golden/blocks_1g.json; bench.py checks
# every block's stream against it)
BLOCK_SET_1G = ("blocks_1g_w28", "syn_text", 1_000_000_000, 0, 28, 32)
