#!/bin/bash
# round 5, run 13: merged prologue launch (conversion + weight packing + W2 transpose), block order of the merged gradient launch
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run13.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'], d['final_aux']['loss'], d['final_aux']['weight_l2'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- default" >> $F; pol
echo "-- ARP_DT_MERGE=0" >> $F; ARP_DT_MERGE=0 pol
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8) >> $F
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_run13 -- python3 $R/bench.py --path policy --steps 50 --warmup 3 --cpu-seconds 0 > $R/gpurun_out/prof_run13.log 2>&1
find $R/gpurun_out/prof_run13 -name "*kernel_trace.csv" -delete
cp $(find $R/gpurun_out/prof_run13 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/run13_policy_kernel_stats.csv
cd $R
cat $F
