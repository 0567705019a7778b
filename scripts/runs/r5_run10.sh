#!/bin/bash
# round 5, run 10: image_text_input's K-tiles dealt round-robin over the workgroups (one contiguous stretch of every row in flight) vs one K range per workgroup
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run10.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2; do
echo "-- round-robin K-tiles (default)" >> $F; pol
echo "-- ARP_DT_ITI_CYCLIC=0" >> $F; ARP_DT_ITI_CYCLIC=0 pol
echo "-- round-robin, 256 workgroups" >> $F; ARP_DT_ITI_WGS=257 pol
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5) >> $F
cat $F
