#!/bin/bash
# round 5, run 15: attention without the per-score mask on fully valid key tiles (non-causal): ViT-B/16 and ViT-B/32 passes, same box, old library = arp_amd/alt/iti_r4
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run15.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- ViT-B/16 batch 256: new / old" >> $F
one --model ViT-B/16 --batch 256
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so one --model ViT-B/16 --batch 256
echo "-- ViT-B/32 batch 1024: new / old" >> $F
one
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so one
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_clip_gpu.py tests/test_ops_gpu.py tests/test_m3ae_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
