#!/bin/bash
# round 5, run 25: out_proj on gemm256 when the pass shares the chip with another part stream (default now) vs gemm2w (ARP_OUT_G256=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run25.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- ViT-B/32: default (gemm256 beside the other stream) / ARP_OUT_G256=0" >> $F
one
ARP_OUT_G256=0 one
done
for rep in 1 2; do
echo "-- ViT-B/16: default / ARP_OUT_G256=0" >> $F
one --model ViT-B/16 --batch 256
ARP_OUT_G256=0 one --model ViT-B/16 --batch 256
done
echo "-- one stream (gemm2w either way): default" >> $F
one --streams 1
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_clip_gpu.py tests/test_m3ae_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
