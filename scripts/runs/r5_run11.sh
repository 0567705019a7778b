#!/bin/bash
# round 5, run 11: the policy step's small dependent launches merged (ARP_DT_MERGE)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run11.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'], d.get('final_aux'))" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- merged (default)" >> $F; pol
echo "-- ARP_DT_MERGE=0" >> $F; ARP_DT_MERGE=0 pol
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8) >> $F
cat $F
