#!/bin/bash
# floor probe 2 on the f32 hand-offs (x and a), the three repaired tests, the encoder's correction plans (8 seeds) and their step times
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
ARP_DT_ADAPTER_PLAN=22e ARP_DT_MIX_X16=0 python scripts/policy_floor_probe2.py 8 > $O/r6_policy_floor2.txt 2>&1
(timeout 900 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py -q -m gpu -k "hi_lo_binary16 or round5_launch or trajectory or many_steps or sixteen" 2>&1 | tail -30) > $O/r6_gpu_suite_eighth.txt 2>&1
python scripts/n1_plan_sweep.py 8 1221,1211,1121,1111 > $O/r6_n1_plan_sweep.txt 2>&1
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
  for plan in 1221 1211 1121 1111; do
    one "N1 f16c plan $plan              " env ARP_F16C_PLAN=$plan $N1
  done
  one "N1 f16c plan 1221 split 60     " env ARP_ENC_SPLIT=60 $N1
done
} > $O/r6_n1_plans_time.txt 2>&1
tail -12 $O/r6_policy_floor2.txt | cut -c1-250; tail -12 $O/r6_gpu_suite_eighth.txt | cut -c1-300; tail -10 $O/r6_n1_plan_sweep.txt | cut -c1-250; cut -c1-230 $O/r6_n1_plans_time.txt
