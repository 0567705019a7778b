#!/bin/bash
# round 5, run 2: ARP_MODE_F16C on the fp4 MFMA -- unit test (exact restatement + error reduction), encoder parity, N1 probe, step time
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run2.txt
rm -f $F
echo "== unit: MIXC gemm (fp4 corrections)" >> $F
(timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "f16c" -x -s 2>&1 | grep -E "f16c gemm|passed|failed|Error|error" | tail -12) >> $F
echo "== N1 step time" >> $F
for m in "--mode f32 --encoder-mode f16c" "--mode f16 --encoder-mode f16c" "--mode f16"; do
  echo "-- $m" >> $F
  timeout 300 python bench.py --path policy --with-encoder $m --cpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity'), d.get('top_sites_ms'))" >> $F 2>&1
done
echo "== N1 probe: enc/policy pairs, ${SEEDS:-4} seeds" >> $F
(timeout 1200 python scripts/n1_parity_probe.py ${SEEDS:-4} ${PAIRS:-f16c:f32,f16c:f16} 2>&1 | tail -8) >> $F
cat $F
