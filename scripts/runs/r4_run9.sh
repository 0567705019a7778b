#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 1500 python -m pytest tests/test_policy_gpu.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8) > $O/r4_t_policy.txt
rm -f $O/r4_pf_x3.txt
for x3 in 0 1 0 1; do
  echo "== ARP_PF_X3=$x3" >> $O/r4_pf_x3.txt
  ARP_PF_X3=$x3 python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('policy', d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:6]})" >> $O/r4_pf_x3.txt
done
cat $O/r4_t_policy.txt; cat $O/r4_pf_x3.txt
