#!/bin/bash
# round 3, GPU run 1: parity suite without the two cases whose fixtures are still being generated, then a short bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "not text_1g and not text_100m_w24" > gpurun_out/r3_pytest1.log 2>&1; tail -3 gpurun_out/r3_pytest1.log
python bench.py --steps 2 --warmup 1 > gpurun_out/r3_bench1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r3_bench1.log | cut -c1-3000
