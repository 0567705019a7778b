#!/bin/bash
# round 6, first GPU call: the full -m gpu suite, the encoder part-stream A/B (VERDICT r5 next #1a), the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
mkdir -p $O
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > $O/r6_gpu_suite_first.txt 2>&1
N1="python bench.py --path policy --with-encoder --mode f16 --steps 20 --warmup 5 --cpu-seconds 0 --parity-frames 0 --no-secondary"
run() {  # label, env...
  L=$1; shift
  env "$@" $N1 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
print('$L', 'ms_per_step', d['ms_per_step'], 'samples/s', round(d['value'],1), dict(list(s.items())[:6]))"
}
{
for rep in 1 2; do
for EXTRA in "--encoder-mode f16c" ""; do
  echo "== encoder mode: ${EXTRA:-f16} (rep $rep)"
  run "streams=1        " ARP_ENC_STREAMS=1
  run "streams=2        " ARP_ENC_STREAMS=2
  run "streams=2 captured" ARP_ENC_STREAMS=2 ARP_DT_ENC_EAGER=0
  run "streams=1 captured" ARP_ENC_STREAMS=1 ARP_DT_ENC_EAGER=0
  run "streams=2 split60" ARP_ENC_STREAMS=2 ARP_ENC_SPLIT=60
  run "streams=2 split68" ARP_ENC_STREAMS=2 ARP_ENC_SPLIT=68
  run "streams=3        " ARP_ENC_STREAMS=3
done
done
} > $O/r6_n1_streams.txt 2>&1
(time python bench.py) > $O/r6_bench_first.jsonl 2> $O/r6_bench_first.err
cp $O/bench_full.json $O/r6_bench_first_full.json
cat $O/r6_gpu_suite_first.txt; cat $O/r6_n1_streams.txt; tail -c 6000 $O/r6_bench_first.jsonl; tail -5 $O/r6_bench_first.err
