#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
F=gpurun_out/r5_b16_batch.txt; rm -f $F
for b in 256 512 1024; do for s in 2 1; do
echo "-- ViT-B/16 batch $b streams $s" >> $F
timeout 300 python bench.py --model ViT-B/16 --batch $b --streams $s --no-secondary --cpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity'])" >> $F 2>&1
done; done
cat $F
