#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box through gpurun):
#   bash tests/prof_run.sh r02
# kernel trace + stats of the bench command, then two separate PMC passes (never combined with traces).
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
CMD="python3 bench.py --no-cpu --block-streams 0 --no-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_pmc_write.log 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE for this path's access pattern (random 8-byte reads, random 4-byte stores in 4 GiB)
hipcc --offload-arch=gfx950 -O3 -o /tmp/random_access tests/microbench/random_access.hip 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- /tmp/random_access > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- /tmp/random_access > $OUT/cal_write.log 2>&1
find $OUT -name "*.csv" | head -20
tail -n 2 $OUT/bench_trace.log
