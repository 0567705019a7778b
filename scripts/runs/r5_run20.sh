#!/bin/bash
# round 5, run 20: LayerNorm folded into the consumer GEMMs again, with the folded epilogue's loads batched (ARP_LN_FOLD=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run20.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2; do
echo "-- ViT-B/16 batch 256: default / ARP_LN_FOLD=1" >> $F
one --model ViT-B/16 --batch 256
ARP_LN_FOLD=1 one --model ViT-B/16 --batch 256
echo "-- ViT-B/32 batch 1024: default / ARP_QKV_FUSED=0 / ARP_LN_FOLD=1" >> $F
one
ARP_QKV_FUSED=0 one
ARP_LN_FOLD=1 one
done
echo "== tests" >> $F
(ARP_LN_FOLD=1 timeout 2400 python -m pytest tests/test_clip_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
