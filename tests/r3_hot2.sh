#!/bin/bash
mkdir -p gpurun_out
{
for hm in 256 2048 8192; do
echo "== hot_min $hm"
NLZM_HOT_MIN=$hm NLZM_WAIT_PRINT=1 timeout 300 python tests/gpu_one.py 20e6 28 1 2>&1 | grep "^cycles\|^worker\|^hot\|^finder: worker\|^[0-9]"
done
for hm in 2048 8192; do
echo "== hot_min $hm"
NLZM_HOT_MIN=$hm NLZM_WAIT_PRINT=1 timeout 600 python tests/gpu_one.py 300e6 28 1 2>&1 | grep "^cycles\|^worker\|^hot\|^finder: worker\|^[0-9]"
done
} > gpurun_out/$1.log 2>&1
cat gpurun_out/$1.log
