#!/bin/bash
# rocprofv3 kernel traces of the two training benches (bench.py --path policy / --path finetune); run on the GPU box via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in policy finetune; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_${P}_trace -- python3 $R/bench.py --path $P --steps 10 --warmup 3 --cpu-seconds 0 > $R/gpurun_out/prof_r3_${P}_trace.log 2>&1
  find $R/gpurun_out/prof_r3_${P}_trace -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/prof_r3_${P}_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3_${P}_kernel_stats.csv
done
ls -la $R/gpurun_out/*.csv
