#!/bin/bash
# The reference on the whole 1e9-byte stand-in (BASELINE config 4): about an hour of one CPU core.
# TEST INFRASTRUCTURE ONLY.  Runs where oracle/_ref/nlzm_ref exists (built from /root/reference by oracle/Makefile).
#   bash tests/full_ref.sh [out.json]
# Equivalent to `python oracle/make_golden_full.py text_1g_w28` (which also updates tests/golden/full.json); this script
# only times the run and prints size + SHA-256, e.g. on the GPU box's host cores for the full-depth CPU figure.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/dev/stdout}
TMP=$(mktemp -d)
python3 - "$TMP/in.bin" <<'PY'
import sys
sys.path.insert(0, '.')
from nlzm_amd import corpus
corpus.syn_text(1_000_000_000, corpus.SEED).tofile(sys.argv[1])
PY
T0=$(date +%s.%N)
taskset -c $(( $(nproc) / 2 )) oracle/_ref/nlzm_ref -window:28 c "$TMP/in.bin" "$TMP/out.nlzm" > /dev/null
T1=$(date +%s.%N)
python3 - "$TMP/out.nlzm" "$T0" "$T1" > "$OUT" <<'PY'
import hashlib, json, os, sys
h = hashlib.sha256()
with open(sys.argv[1], 'rb') as f:
    for b in iter(lambda: f.read(1 << 24), b''):
        h.update(b)
dt = float(sys.argv[3]) - float(sys.argv[2])
print(json.dumps({"input": "corpus.syn_text(1_000_000_000, SEED)", "window": 28, "stream_bytes": os.path.getsize(sys.argv[1]),
                  "stream_sha256": h.hexdigest(), "seconds": round(dt, 1), "MB_per_s": round(1000 / dt, 4), "host_cores": os.cpu_count()}))
PY
rm -rf "$TMP"
