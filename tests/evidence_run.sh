#!/bin/bash
# Round evidence, run on the GPU box through gpurun:  bash tests/evidence_run.sh r02
# (parity suite, bench line, stage accounting of two configurations, then the rocprofv3 passes of tests/prof_run.sh)
R=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${R}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${R}_pytest_gpu.log
python bench.py > gpurun_out/${R}_bench.log 2>&1; tail -1 gpurun_out/${R}_bench.log | cut -c1-200
{
  echo "# NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1   (3 MB of text, -window:20)"
  NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1 2>&1
  echo; echo "# NLZM_WAIT_PRINT=1 python tests/gpu_one.py 8e6 28 1   (8 MB of text, -window:28 -> 23)"
  NLZM_WAIT_PRINT=1 python tests/gpu_one.py 8e6 28 1 2>&1
  echo; echo "# NLZM_WAIT_PRINT=1 python tests/gpu_blocks.py 8 20 32   (32 independent blocks of 8 MB in flight)"
  NLZM_WAIT_PRINT=1 python tests/gpu_blocks.py 8 20 32 2>&1 | grep -v "^cycles\|^finder\|^table\|^worker\|^parser" 
} > gpurun_out/${R}_wave_accounting.txt
bash tests/prof_run.sh $R > gpurun_out/${R}_prof_run.log 2>&1; tail -2 gpurun_out/${R}_prof_run.log
