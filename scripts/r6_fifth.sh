#!/bin/bash
# after prime_runtime + adapter corrections default on: full suite, N1 flows (gate / no gate), policy line, default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > $O/r6_gpu_suite_fifth.txt 2>&1
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
  echo "== rep $rep (prime_runtime in every create)"
  one "gate,    two slots, ahead      " $N1
  one "no gate, two slots, ahead      " $N1 --parity-frames 0
  one "no gate, two slots, at head    " $N1 --parity-frames 0 --no-encode-ahead
  one "no gate, single slot           " $N1 --parity-frames 0 --single-slot
  one "no gate, 1 stream, ahead       " env ARP_ENC_STREAMS=1 $N1 --parity-frames 0
  one "no gate, vperm off             " env ARP_F16C_VPERM=0 $N1 --parity-frames 0
  one "policy alone (corrected, 22h)  " python bench.py --path policy --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary
  one "policy alone, --no-adapter-c   " python bench.py --path policy --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --no-adapter-c
  one "label headline                 " python bench.py --cpu-seconds 0 --no-secondary
done
} > $O/r6_n1_flow4.txt 2>&1
(time python bench.py) > $O/r6_bench_fifth.jsonl 2> $O/r6_bench_fifth.err
cp $O/bench_full.json $O/r6_bench_fifth_full.json
tail -12 $O/r6_gpu_suite_fifth.txt | cut -c1-300; cut -c1-230 $O/r6_n1_flow4.txt
python - <<PY
import json
for l in open("$O/r6_bench_fifth.jsonl"):
    d=json.loads(l); print(d.get("secondary","HEADLINE"), d.get("value"), d.get("ms_per_step"), (d.get("parity") or {}).get("err"))
PY
tail -3 $O/r6_bench_fifth.err
