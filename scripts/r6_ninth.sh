#!/bin/bash
# after the x4 side-output fix (bit_cast of an ext-vector element): the chain probe, the new test, adapter plans and encoder plans again
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python scripts/adapter_chain_probe.py > $O/r6_adapter_chain_probe_fixed.txt 2>&1
(timeout 900 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py tests/test_ops_gpu.py -q -m gpu -k "operand_rows or sixteen or trajectory or f16c or eight_seeds or encoder_outputs" 2>&1 | tail -30) > $O/r6_gpu_suite_ninth.txt 2>&1
PER_SEED=1 python scripts/adapter_plan_gpu.py off 22d 22h 12h 21h 11h 12d 11d > $O/r6_adapter_plans_fixed.txt 2>&1
python scripts/n1_plan_sweep.py 8 1221,1211,1121,1111,1110 > $O/r6_n1_plan_sweep_fixed.txt 2>&1
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
P="python bench.py --path policy --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== rep $rep"
  for plan in 1221 1211 1111 1110; do
    one "N1 f16c plan $plan              " env ARP_F16C_PLAN=$plan $N1
  done
  for plan in 22d 22h 12h 11h 11d; do
    one "policy alone $plan              " env ARP_DT_ADAPTER_PLAN=$plan $P
  done
  one "policy alone, --no-adapter-c   " $P --no-adapter-c
done
} > $O/r6_plans_time_fixed.txt 2>&1
cut -c1-250 $O/r6_adapter_chain_probe_fixed.txt | tail -14; tail -8 $O/r6_gpu_suite_ninth.txt | cut -c1-300; grep "^plan" $O/r6_adapter_plans_fixed.txt | cut -c1-300; grep "^plan" $O/r6_n1_plan_sweep_fixed.txt | cut -c1-200; cut -c1-200 $O/r6_plans_time_fixed.txt
