#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 1500 python -m pytest tests/test_policy_gpu.py tests/test_finetune_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|differ" | tail -12) > $O/r4_t_policy.txt
rm -f $O/r4_iti_x3.txt
for v in "ARP_DT_ITI_X3=0 ARP_SPLITK_REDUCE4=0" "ARP_DT_ITI_X3=0 ARP_SPLITK_REDUCE4=1" "ARP_DT_ITI_X3=1 ARP_SPLITK_REDUCE4=1" "ARP_DT_ITI_X3=0 ARP_SPLITK_REDUCE4=0" "ARP_DT_ITI_X3=1 ARP_SPLITK_REDUCE4=1"; do
  echo "== $v" >> $O/r4_iti_x3.txt
  env $v python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('policy', d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:6]})" >> $O/r4_iti_x3.txt
done
for v in "ARP_SPLITK_REDUCE4=0" "ARP_SPLITK_REDUCE4=1"; do
  echo "== finetune $v" >> $O/r4_iti_x3.txt
  env $v python bench.py --path finetune --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'])" >> $O/r4_iti_x3.txt
done
cat $O/r4_t_policy.txt; cat $O/r4_iti_x3.txt
