#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_stream_bit_exact or test_find_matches or test_batching or large_window or multi_device or blocks_in_flight or ragged or cli_" 2>&1 | tail -3
NLZM_LIB=nlzm_amd/libnlzm_hip_prof.so NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1 2>&1 | grep "^cycles\|^worker\|^[0-9]"
NLZM_WAIT_PRINT=1 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^worker\|^[0-9]"
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
