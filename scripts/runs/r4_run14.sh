#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
cp arp_amd/libarp_hip.so /tmp/p1.so; cp arp_amd/libarp_hip_p2.so /tmp/p2.so; cp arp_amd/libarp_hip_p3.so /tmp/p3.so
rm -f $O/r4_adamw_depth.txt
for v in p1 p2 p3 p1 p2 p3; do
  cp /tmp/$v.so arp_amd/libarp_hip.so
  echo "== fused AdamW epilogue, loads $v iterations ahead" >> $O/r4_adamw_depth.txt
  python bench.py --path finetune --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'], d['roofline'].get('achieved'), {k:v for k,v in list(d['sites_ms_per_step'].items())[:5]})" >> $O/r4_adamw_depth.txt
done
cp /tmp/p1.so arp_amd/libarp_hip.so
(timeout 900 python -m pytest tests/test_finetune_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2) >> $O/r4_adamw_depth.txt
cat $O/r4_adamw_depth.txt
