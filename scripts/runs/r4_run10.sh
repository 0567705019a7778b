#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
cp arp_amd/libarp_hip.so /tmp/main.so; cp arp_amd/libarp_hip_alt.so /tmp/alt.so
(timeout 1500 python -m pytest tests/test_finetune_gpu.py tests/test_policy_gpu.py -q -x -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8) > $O/r4_t_ft.txt
rm -f $O/r4_adamw_pipe.txt
for v in main alt main alt; do
  cp /tmp/$v.so arp_amd/libarp_hip.so
  echo "== $v (main = ARP_ADAMW_PIPE 1, alt = 0)" >> $O/r4_adamw_pipe.txt
  python bench.py --path finetune --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('achieved'), d['roofline'].get('avg_launch_ms'), {k:v for k,v in list(d['sites_ms_per_step'].items())[:8]})" >> $O/r4_adamw_pipe.txt
done
cp /tmp/main.so arp_amd/libarp_hip.so
cat $O/r4_t_ft.txt; cat $O/r4_adamw_pipe.txt
