#!/bin/bash
# round 5, run 3: F16C with the adapter corrections -- unit test, step time per plan, 8-seed probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run3.txt
rm -f $F
echo "== unit: MIXC gemm (fp4 corrections)" >> $F
(timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "f16c" -x -s 2>&1 | grep -E "f16c gemm|passed|failed|Error|error" | tail -12) >> $F
echo "== N1 step time (policy f16 + adapter corrections; ARP_F16C_PLAN = in_proj, out_proj, fc1, fc2)" >> $F
for plan in 1221 1222 2222 1111; do
  echo "-- plan $plan" >> $F
  ARP_F16C_PLAN=$plan timeout 300 python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --cpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity'), d.get('top_sites_ms'))" >> $F 2>&1
done
echo "-- policy alone, f16 (+ adapter corrections)" >> $F
for x in "" "--adapter-c"; do
timeout 300 python bench.py --path policy --mode f16 $x --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity'), d.get('top_sites_ms'))" >> $F 2>&1
done
echo "== N1 probe, 8 seeds (plan ${PLAN:-1221})" >> $F
(ARP_F16C_PLAN=${PLAN:-1221} timeout 1500 python scripts/n1_parity_probe.py 8 ${PAIRS:-f16c:f32,f16c:f16+c,f32:f16+c,f32:f16} 2>&1 | tail -14) >> $F
cat $F
