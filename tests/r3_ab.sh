#!/bin/bash
mkdir -p gpurun_out
{
for rep in 1 2; do
for m in 0 1; do
echo "== assume_skip $m"
NLZM_ASSUME_SKIP=$m NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^worker lanes: [0-9]* calls\|^[0-9]"
done
done
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
