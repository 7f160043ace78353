#!/bin/bash
# Same-box comparison behind profiles/r03_worker_ab.log (run through gpurun): the worker lanes that wait for the finder's decisions
# (nlzm_amd/libnlzm_hip_old.so: commit 443affa built with tests/build_variant.sh from a checkout of that commit) against the lanes
# that assume them, at 300 MB.  (The NLZM_HOT_* settings of that day's build selected a variant that was dropped.)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_stream_bit_exact or test_find_matches or test_batching or large_window or cli_ or blocks_in_flight or ragged" 2>&1 | tail -3
echo "== old"
NLZM_LIB=nlzm_amd/libnlzm_hip_old.so NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^[0-9]"
for cfg in "0 1024" "6 256" "6 1024" "6 4096" "2 4096"; do
set -- $cfg
echo "== hot waves $1 min $2"
NLZM_HOT_WAVES=$1 NLZM_HOT_MIN=$2 NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^hot\|^worker lanes: [0-9]* cyc\|^[0-9]"
done
echo "== 20 MB old / new"
NLZM_LIB=nlzm_amd/libnlzm_hip_old.so NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^[0-9]"
NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^hot\|^[0-9]"
} > gpurun_out/ab3.log 2>&1
cat gpurun_out/ab3.log
