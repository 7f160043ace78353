#!/bin/bash
# quick parity + timing: golden small cases, 3 MB w20, 20 MB w28 (window 25), accounting at 3 MB and 40 MB w28; $1: log name
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_stream_bit_exact or test_find_matches or test_batching or large_window" 2>&1 | tail -3
NLZM_LIB=nlzm_amd/libnlzm_hip_prof.so NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1 2>&1 | grep -v "^table\|^parser, cycles per pass (\|^parser loader\|^parser wave"
NLZM_WAIT_PRINT=1 python tests/gpu_one.py ${2:-60e6} 28 1 2>&1 | grep -v "^table\|^parser, cycles per pass (\|^parser loader\|^parser wave"
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
