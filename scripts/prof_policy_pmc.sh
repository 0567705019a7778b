#!/bin/bash
# HBM traffic of the policy train step's kernels: rocprofv3 FETCH_SIZE / WRITE_SIZE in separate --pmc passes (kernel-trace only) over a
# short `bench.py --path policy` run with eager launches; run on the GPU box via gpurun, then scripts/summarize_policy_pmc.py here.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export ARP_DT_GRAPH=0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_r3_policy_$C -- python3 $R/bench.py --path policy --steps 3 --warmup 2 --cpu-seconds 0 > $R/gpurun_out/prof_r3_policy_$C.log 2>&1
  find $R/gpurun_out/prof_r3_policy_$C -name "*kernel_trace.csv" -delete
done
ls -la $R/gpurun_out/prof_r3_policy_*/*/ | head
