#!/bin/bash
# round 5, run 5: after the masked over-issue -- harness A/B (KV = 0 vs KV = 1 with tail), MIXC unit test + race screen, N1 timing, headline
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run5.txt
rm -f $F
echo "== harness (col 1: KV = 0, col 2: KV = 1)" >> $F
(cd scripts && timeout 150 ./gemm256_bench.bin | grep -v masked | cut -c1-175) >> $F 2>&1
echo "== unit" >> $F
(timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_m3ae_gpu.py -q -m gpu -x 2>&1 | tail -4) >> $F
echo "== N1 step time" >> $F
for m in "--mode f16 --encoder-mode f16c" "--mode f16"; do
  echo "-- $m" >> $F
  timeout 300 python bench.py --path policy --with-encoder $m --cpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity'), d.get('top_sites_ms'))" >> $F 2>&1
done
echo "== headline" >> $F
for i in 1 2; do timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d.get('roofline_isolated',{}).get('avg_launch_ms'), d['parity'])" >> $F 2>&1; done
cat $F
