#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 1500 python -m pytest tests/test_policy_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6) > $O/r4_t_policy.txt
rm -f $O/r4_side.txt
for v in 0 1 0 1; do
  echo "== ARP_DT_SIDE=$v" >> $O/r4_side.txt
  ARP_DT_SIDE=$v python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print('policy', d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], {k:s[k] for k in list(s)[:8]})" >> $O/r4_side.txt
done
cat $O/r4_t_policy.txt; cat $O/r4_side.txt
