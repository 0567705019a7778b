#!/bin/bash
# default 22d with the binary16 x: floor probe 2, the policy / N1 tests, plans, policy lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python scripts/policy_floor_probe2.py 8 > $O/r6_policy_floor2.txt 2>&1
(time timeout 1200 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py -q -m gpu 2>&1 | tail -15) > $O/r6_gpu_suite_seventh.txt 2>&1
PER_SEED=1 python scripts/adapter_plan_gpu.py 22d > $O/r6_adapter_plans_d2.txt 2>&1
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
p=d.get('parity') or {}
print('$L', 'ms_per_step', d['ms_per_step'], 'parity', p.get('max_logit_err_vs_oracle', p.get('max_cosine_err_vs_oracle')), dict(list(s.items())[:6]))"
}
P="python bench.py --path policy --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  one "policy alone 22d (x16)         " $P
  one "policy alone 22d, x f32        " env ARP_DT_MIX_X16=0 $P
  one "policy alone 22h               " env ARP_DT_ADAPTER_PLAN=22h $P
  one "policy alone, --no-adapter-c   " $P --no-adapter-c
done
} > $O/r6_policy_d.txt 2>&1
tail -12 $O/r6_policy_floor2.txt | cut -c1-250; tail -8 $O/r6_gpu_suite_seventh.txt | cut -c1-300; tail -5 $O/r6_adapter_plans_d2.txt | cut -c1-250; cut -c1-230 $O/r6_policy_d.txt
