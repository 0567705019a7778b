#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
cp arp_amd/libarp_hip.so /tmp/main.so; cp arp_amd/libarp_hip_alt.so /tmp/alt.so
rm -f $O/r4_pre4.txt
for v in main alt main alt; do
  cp /tmp/$v.so arp_amd/libarp_hip.so
  echo "== $v (alt = preprocess input copy with four loads in flight)" >> $O/r4_pre4.txt
  timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print(round(d['value']), d['ms_per_step'], s['preprocess'], s['vit.patch_embed'])" >> $O/r4_pre4.txt 2>&1
done
cp /tmp/alt.so arp_amd/libarp_hip.so
(timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "preprocess" 2>&1 | grep -E "passed|failed" | tail -2) >> $O/r4_pre4.txt
cp /tmp/main.so arp_amd/libarp_hip.so
cat $O/r4_pre4.txt
