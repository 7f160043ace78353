#!/bin/bash
mkdir -p gpurun_out
{
for lib in old "" a2 a4; do
echo "== lib '$lib'"
L=nlzm_amd/libnlzm_hip${lib:+_$lib}.so
NLZM_LIB=$L NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^worker lanes: [0-9]* calls\|^[0-9]"
done
echo "== 20 MB old / new"
NLZM_LIB=nlzm_amd/libnlzm_hip_old.so NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^[0-9]"
NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^[0-9]"
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
