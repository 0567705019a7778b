#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g256_groupm_time.txt
rm -f $O
for G in 1 2 4 8 16 32 8 1; do
  echo "== ARP_GEMM_GROUP_M=$G" >> $O
  ARP_GEMM_GROUP_M=$G $R/scripts/gemm256_bench.bin 2>&1 | grep -E "^(qkv|c_fc |c_proj |out_proj|c_fc_half|c_proj_half|4096)" | sed -E 's/\| 32x32x16.*//' >> $O
done
cat $O
