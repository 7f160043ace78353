#!/bin/bash
# per-section cycle accounting of the stages (profile build) on 3 MB of text at -window:20; $1: log name
mkdir -p gpurun_out
NLZM_LIB=nlzm_amd/libnlzm_hip_prof.so NLZM_WAIT_PRINT=1 python tests/gpu_one.py ${2:-3e6} ${3:-20} 1 > gpurun_out/$1.log 2>&1
python - <<'PY' >> gpurun_out/$1.log 2>&1
import sys, hashlib
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
from tests import oracle_py
nlzm_amd.init(0)
d = corpus.syn_text(3_000_000)
print("product library, 3 MB w20: bit-exact", nlzm_amd.compress(d, 20) == oracle_py.compress(d, 20), nlzm_amd.timing()["match_parse_ms"])
PY
cat gpurun_out/$1.log
