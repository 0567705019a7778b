#!/bin/bash
# round 5, run 16: the mix inside image_text_input reading the encodings' binary16 copy (ARP_DT_MIX_X16=1): speed, and the 16-seed logits gate
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run16.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- default" >> $F; pol
echo "-- ARP_DT_MIX_X16=1" >> $F; ARP_DT_MIX_X16=1 pol
done
echo "== 16 seeds, default" >> $F
(timeout 1200 python -m pytest tests/test_policy_gpu.py -q -m gpu -x -s -k "sixteen_seeds" 2>&1 | grep -E "passed|failed|seed|max|err" | tail -6) >> $F
echo "== 16 seeds, ARP_DT_MIX_X16=1" >> $F
(ARP_DT_MIX_X16=1 timeout 1200 python -m pytest tests/test_policy_gpu.py -q -m gpu -x -s -k "sixteen_seeds" 2>&1 | grep -E "passed|failed|seed|max|err" | tail -6) >> $F
cat $F
