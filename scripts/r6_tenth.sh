#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 600 python -m pytest tests/test_m3ae_gpu.py -q -m gpu -k "trajectory" -s 2>&1 | grep -v "^$" | tail -30) > $O/r6_gpu_suite_tenth.txt 2>&1
python scripts/n1_plan_sweep.py 8 1220,1210,1120,2220 > $O/r6_n1_plan_sweep_fixed2.txt 2>&1
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
p=d.get('parity') or {}
print('$L', 'ms_per_step', d['ms_per_step'], 'parity', p.get('max_logit_err_vs_oracle', p.get('max_cosine_err_vs_oracle')), dict(list(s.items())[:6]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== rep $rep"
  for plan in 1221 1220 1210 1120 1110; do
    one "N1 f16c plan $plan              " env ARP_F16C_PLAN=$plan $N1
  done
done
} > $O/r6_n1_plans_time_fixed2.txt 2>&1
cut -c1-400 $O/r6_gpu_suite_tenth.txt | tail -14; grep "^plan" $O/r6_n1_plan_sweep_fixed2.txt | cut -c1-200; cut -c1-200 $O/r6_n1_plans_time_fixed2.txt
