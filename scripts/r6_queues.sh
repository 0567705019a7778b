#!/bin/bash
# GPU_MAX_HW_QUEUES 4 vs 8 over the whole default bench + the N1 flow with ONE copy stream; gpu tests of the touched files
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time timeout 1200 python -m pytest tests/test_m3ae_gpu.py tests/test_policy_gpu.py -q -x -m gpu 2>&1 | tail -12) > $O/r6_gpu_suite_fourth.txt 2>&1
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
print('$L', 'ms_per_step', d['ms_per_step'], dict(list(s.items())[:6]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== rep $rep (one copy stream)"
  for Q in 4 8; do
    one "queues $Q: gate, two slots, ahead      " env GPU_MAX_HW_QUEUES=$Q $N1
    one "queues $Q: no gate, two slots, ahead   " env GPU_MAX_HW_QUEUES=$Q $N1 --parity-frames 0
    one "queues $Q: no gate, two slots, at head " env GPU_MAX_HW_QUEUES=$Q $N1 --parity-frames 0 --no-encode-ahead
    one "queues $Q: no gate, single slot        " env GPU_MAX_HW_QUEUES=$Q $N1 --parity-frames 0 --single-slot
  done
done
} > $O/r6_n1_flow2.txt 2>&1
for Q in 4 8; do
  (time GPU_MAX_HW_QUEUES=$Q python bench.py --cpu-seconds 0) > $O/r6_bench_q$Q.jsonl 2> $O/r6_bench_q$Q.err
done
tail -6 $O/r6_gpu_suite_fourth.txt; cut -c1-220 $O/r6_n1_flow2.txt
for Q in 4 8; do python - <<PY
import json
print("== GPU_MAX_HW_QUEUES=$Q")
for l in open("$O/r6_bench_q$Q.jsonl"):
    d=json.loads(l); print(d.get("secondary","HEADLINE"), d.get("value"), d.get("ms_per_step"))
PY
done
