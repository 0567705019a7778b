#!/bin/bash
# rocprofv3 evidence for bench.py (label path): kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (run on the GPU box through gpurun).  Writes under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --cpu-seconds 0 --parity-frames 0 --no-secondary --timed-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cd $R/gpurun_out && find . -name "*.csv" | head -30; tail -2 prof_${TAG}_trace.log | cut -c1-300
# keep the merge small: drop the per-dispatch traces of the long run, keep stats + pmc csv
find prof_${TAG}_trace -name "*kernel_trace.csv" -size +20M -delete
