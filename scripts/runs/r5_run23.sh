#!/bin/bash
# round 5, run 23: tokens_bwd inside the merged gradient launch with batched unconditional loads; old = arp_amd/alt/prev
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run23.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'], d['final_aux']['loss'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- new" >> $F; pol
echo "-- old (arp_amd/alt/prev)" >> $F; ARP_LIB=arp_amd/alt/prev/libarp_hip.so pol
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_run23 -- python3 $R/bench.py --path policy --steps 50 --warmup 3 --cpu-seconds 0 > $R/gpurun_out/prof_run23.log 2>&1
find $R/gpurun_out/prof_run23 -name "*kernel_trace.csv" -delete
grep -E "pf_param_grads|policy_fused_kernel" $(find $R/gpurun_out/prof_run23 -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-120
