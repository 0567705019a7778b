#!/bin/bash
# round 5, run 14: gemm_tn_kernel's f32 tile staged through LDS (whole 256-B row segments) vs stores straight from the accumulators
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run14.txt
rm -f $F
for rep in 1 2; do
echo "== staged" >> $F; timeout 300 scripts/gemm_tn_bench.bin 2>&1 | grep -E "CHECK|128 x|M=128|197376|err" | tail -6 >> $F
echo "== direct (ARP_TN_DIRECT_STORE)" >> $F; timeout 300 scripts/gemm_tn_bench_direct.bin 2>&1 | grep -E "CHECK|128 x|M=128|197376|err" | tail -6 >> $F
done
timeout 300 scripts/gemm_tn_bench.bin > $O/r5_run14_full.txt 2>&1
cat $F
