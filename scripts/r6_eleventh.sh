#!/bin/bash
# default plans 1110 (encoder) / 22d (adapter) with the unused side outputs skipped: full suite, N1 and policy lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/r6_gpu_suite_eleventh.txt 2>&1
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
p=d.get('parity') or {}
print('$L', 'ms_per_step', d['ms_per_step'], 'parity', p.get('max_logit_err_vs_oracle', p.get('max_cosine_err_vs_oracle')), dict(list(s.items())[:7]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== rep $rep"
  one "N1 f16c (1110)                 " $N1 --encoder-mode f16c
  one "N1 f16c plan 1221              " env ARP_F16C_PLAN=1221 $N1 --encoder-mode f16c
  one "N1 f16c plan 1111              " env ARP_F16C_PLAN=1111 $N1 --encoder-mode f16c
  one "N1 f16c plan 1100              " env ARP_F16C_PLAN=1100 $N1 --encoder-mode f16c
  one "N1 plain f16 encoder           " $N1
  one "policy alone                   " python bench.py --path policy --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary
done
} > $O/r6_n1_final_plans.txt 2>&1
python scripts/n1_plan_sweep.py 8 1110,1100,1000 > $O/r6_n1_plan_sweep_fixed3.txt 2>&1
nproc > $O/r6_nproc.txt
tail -14 $O/r6_gpu_suite_eleventh.txt | cut -c1-300; cut -c1-250 $O/r6_n1_final_plans.txt; grep "^plan" $O/r6_n1_plan_sweep_fixed3.txt | cut -c1-200; cat $O/r6_nproc.txt
