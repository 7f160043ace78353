#!/bin/bash
# Same-box comparison behind profiles/r03_hot_bin_waves_ab.log (run through gpurun): parity with a wave on nearly every bin, then
# 300 MB without the hot bins' waves (NLZM_HOT_WAVES=0) and with them at several thresholds (NLZM_HOT_MIN positions per launch).
mkdir -p gpurun_out
{
NLZM_HOT_MIN=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_stream_bit_exact or test_find_matches or test_batching or large_window" 2>&1 | tail -3
for cfg in "0 4096" "2 4096" "2 8192" "2 16384" "2 32768"; do
set -- $cfg
echo "== hot waves $1 min $2"
NLZM_HOT_WAVES=$1 NLZM_HOT_MIN=$2 NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^hot\|^worker lanes: [0-9]* calls\|^[0-9]"
done
} > gpurun_out/hot7.log 2>&1
cat gpurun_out/hot7.log
