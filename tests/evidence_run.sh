#!/bin/bash
# Round evidence, run on the GPU box through gpurun:  bash tests/evidence_run.sh r05
# (stage accounting of three configurations, the profile build's per-section numbers, block mode; then the rocprofv3 passes
#  of tests/prof_run.sh.  The parity suite and the whole-stream bench line are separate calls: they take 12 and 10 minutes.)
R=${1:-r05}
mkdir -p gpurun_out
bash tests/build_prof.sh      # the profile library does not travel with the snapshot (.gpurunignore): built here
{
  echo "# NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1   (3 MB of text, -window:20; product library)"
  NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1 2>&1
  echo; echo "# the same with the profile build (nlzm_amd/libnlzm_hip_prof.so: per-section cycle counters; slower by the counters)"
  NLZM_LIB=nlzm_amd/libnlzm_hip_prof.so NLZM_WAIT_PRINT=1 python tests/gpu_one.py 3e6 20 1 2>&1
  echo; echo "# NLZM_WAIT_PRINT=1 python tests/gpu_one.py 20e6 28 1   (20 MB of text, -window:28 -> 25)"
  NLZM_WAIT_PRINT=1 python tests/gpu_one.py 20e6 28 1 2>&1
  echo; echo "# NLZM_WAIT_PRINT=1 python tests/gpu_one.py 300e6 28 1   (300 MB of text, -window:28: depth)"
  NLZM_WAIT_PRINT=1 python tests/gpu_one.py 300e6 28 1 2>&1
  echo; echo "# the same with the profile build"
  NLZM_LIB=nlzm_amd/libnlzm_hip_prof.so NLZM_WAIT_PRINT=1 python tests/gpu_one.py 300e6 28 1 2>&1
  echo; echo "# NLZM_WAIT_PRINT=1 python tests/gpu_curve.py 17 28 1,8,32 block_batch_chunks=8   (independent blocks of 17 MB in flight: per-stream accounting)"
  NLZM_WAIT_PRINT=1 python tests/gpu_curve.py 17 28 1,8,32 block_batch_chunks=8 2>&1
} > gpurun_out/${R}_wave_accounting.txt
{
  echo "# REAL text (corpus.real_text: the image's own source files; hottest BT4 head 17 % of the positions): python tests/gpu_opt.py 30e6 24 real_text"
  python tests/gpu_opt.py 30e6 24 real_text 2>&1
  echo; echo "# python tests/gpu_opt.py 300e6 28 real_text   (the reference on the build container's copy of these bytes: 471 s = 0.64 MB/s, tests/golden/real.json)"
  python tests/gpu_opt.py 300e6 28 real_text 2>&1
  echo; echo "# the same without the helper parsers: python tests/gpu_opt.py 300e6 28 real_text parser_helper=0"
  python tests/gpu_opt.py 300e6 28 real_text parser_helper=0 2>&1
  echo; echo "# markup (corpus.xml_like): python tests/gpu_opt.py 100e6 26 xml_like"
  python tests/gpu_opt.py 100e6 26 xml_like 2>&1
} > gpurun_out/${R}_real_text.txt
bash tests/prof_run.sh $R > gpurun_out/${R}_prof_run.log 2>&1; tail -2 gpurun_out/${R}_prof_run.log
