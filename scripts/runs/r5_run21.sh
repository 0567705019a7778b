#!/bin/bash
# round 5, run 21: gemm2w's bias fragments unconditional, preprocess prologue batched; old = arp_amd/alt/prev (the tree two commits earlier)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run21.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- ViT-B/32: new / old" >> $F
one
ARP_LIB=arp_amd/alt/prev/libarp_hip.so one
done
echo "-- ViT-B/16: new / old" >> $F
one --model ViT-B/16 --batch 256
ARP_LIB=arp_amd/alt/prev/libarp_hip.so one --model ViT-B/16 --batch 256
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_clip_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
