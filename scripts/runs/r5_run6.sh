#!/bin/bash
# round 5, run 6: tile-round split (full rounds on gemm256, the rest of the rows on the 128 x 128 kernel) in the whole pass
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run6.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2; do
echo "-- split on (default)" >> $F; one
echo "-- ARP_GEMM_SPLITM=0" >> $F; ARP_GEMM_SPLITM=0 one
echo "-- split on + out_proj on gemm256 (ARP_OUT_G256=1)" >> $F; ARP_OUT_G256=1 one
done
echo "-- ViT-B/16 batch 256: split on / off" >> $F
one --model ViT-B/16 --batch 256
ARP_GEMM_SPLITM=0 one --model ViT-B/16 --batch 256
echo "-- single stream: split on / off" >> $F
one --streams 1
ARP_GEMM_SPLITM=0 one --streams 1
echo "== tests" >> $F
(timeout 1200 python -m pytest tests/test_clip_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3) >> $F
cat $F
