#!/bin/bash
# default plan 22e, comm stream before RCCL, the "d" hand-off (binary16 adapter output + e2m1 error code): suite, fp4 decode probe, plans, bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
./scripts/fp4_cvt_probe2.bin > $O/r6_fp4_cvt_probe2.txt 2>&1
(time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -25) > $O/r6_gpu_suite_sixth.txt 2>&1
PER_SEED=1 python scripts/adapter_plan_gpu.py 22e 22d 22h 12d > $O/r6_adapter_plans_d.txt 2>&1
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
  one "N1 f16c (22e default)          " $N1
  one "N1 f16c, plan 22d              " env ARP_DT_ADAPTER_PLAN=22d $N1
  one "policy alone 22e               " $P
  one "policy alone 22d               " env ARP_DT_ADAPTER_PLAN=22d $P
  one "policy alone 22h               " env ARP_DT_ADAPTER_PLAN=22h $P
  one "policy alone, --no-adapter-c   " $P --no-adapter-c
  one "policy staged 22e              " $P --staged
  one "policy staged 22d              " env ARP_DT_ADAPTER_PLAN=22d $P --staged
done
} > $O/r6_n1_flow5.txt 2>&1
(time python bench.py) > $O/r6_bench_sixth.jsonl 2> $O/r6_bench_sixth.err
cp $O/bench_full.json $O/r6_bench_sixth_full.json
cat $O/r6_fp4_cvt_probe2.txt; tail -14 $O/r6_gpu_suite_sixth.txt | cut -c1-300; tail -12 $O/r6_adapter_plans_d.txt | cut -c1-250; cut -c1-230 $O/r6_n1_flow5.txt
python - <<PY
import json
for l in open("$O/r6_bench_sixth.jsonl"):
    d=json.loads(l); print(d.get("secondary","HEADLINE"), d.get("value"), d.get("ms_per_step"), (d.get("parity") or {}).get("err"))
PY
tail -3 $O/r6_bench_sixth.err
