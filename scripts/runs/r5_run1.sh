#!/bin/bash
# round 5, run 1: (a) MIX8 GEMM unit test, (b) KV=1 vs KV=0 on the whole labelling pass (ARP_LIB selects the build: the product is never overwritten),
# (c) row N1: which encoder / policy mode pairs meet 1e-3, and what the f16c step costs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run1.txt
rm -f $F
echo "== unit: MIX8 gemm" >> $F
(timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "f16c or gemm256_race or gemm_nt" -x 2>&1 | tail -5) >> $F
echo "== KV A/B (frames/s, ms/step, c_fc, c_proj, qkv_attn, out_proj site ms)" >> $F
for v in kv1 kv0 kv1 kv0; do
  if [ $v = kv0 ]; then export ARP_LIB=arp_amd/alt/kv0/libarp_hip.so; else unset ARP_LIB; fi
  echo "-- $v" >> $F
  timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['top_sites_ms']; print(round(d['value']), d['ms_per_step'], s, d.get('roofline_isolated',{}).get('avg_launch_ms'))" >> $F 2>&1
done
unset ARP_LIB
echo "== N1 probe: enc/policy pairs, 4 seeds" >> $F
(timeout 900 python scripts/n1_parity_probe.py 4 f16c:f32,f16c:f16,f16:f32 2>&1 | tail -8) >> $F
echo "== N1 step time" >> $F
for m in "--mode f32 --encoder-mode f16c" "--mode f16 --encoder-mode f16c" "--mode f16"; do
  echo "-- $m" >> $F
  timeout 300 python bench.py --path policy --with-encoder $m --cpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity'), d.get('top_sites_ms'))" >> $F 2>&1
done
cat $F
