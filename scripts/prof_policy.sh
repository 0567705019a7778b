#!/bin/bash
# rocprofv3 kernel trace of the policy train step (bench.py --path policy); run on the GPU box via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1_policy}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py --path policy --steps 10 --warmup 3 > $R/gpurun_out/prof_${TAG}_trace.log 2>&1
cd $R/gpurun_out && find prof_${TAG}_trace -name "*kernel_stats.csv" | head; find prof_${TAG}_trace -name "*kernel_trace.csv" -size +20M -delete
