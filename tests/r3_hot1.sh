#!/bin/bash
# hot-bin waves: parity with a low threshold (so that small inputs use them), then the depth check
mkdir -p gpurun_out
{
NLZM_HOT_MIN=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_stream_bit_exact or test_find_matches or test_batching or large_window or cli_" 2>&1 | tail -3
NLZM_HOT_MIN=64 NLZM_WAIT_PRINT=1 timeout 300 python tests/gpu_one.py 3e6 20 1 2>&1 | grep "^cycles\|^worker\|^hot\|^[0-9]"
NLZM_WAIT_PRINT=1 timeout 300 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^worker\|^hot\|^finder: worker\|^[0-9]"
NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^worker\|^hot\|^finder: worker\|^[0-9]"
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
