#!/bin/bash
# round 5, run 17: residual rows / mask rows of the GEMM epilogues loaded unconditionally (clamped addresses) instead of behind a branch + vmcnt(0) each
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run17.txt
rm -f $F
one() { timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['top_sites_ms'], d['parity']['max_cosine_err_vs_oracle'])" >> $F 2>&1; }
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- ViT-B/32 batch 1024: new / old (arp_amd/alt/iti_r4)" >> $F
one
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so one
echo "-- ViT-B/16 batch 256: new / old" >> $F
one --model ViT-B/16 --batch 256
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so one --model ViT-B/16 --batch 256
done
echo "-- single stream ViT-B/32: new / old" >> $F
one --streams 1
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so one --streams 1
echo "-- policy step: new (x2)" >> $F
pol; pol
echo "-- finetune step: new / old" >> $F
timeout 300 python bench.py --path finetune --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $F 2>&1
ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so timeout 300 python bench.py --path finetune --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $F 2>&1
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_clip_gpu.py tests/test_ops_gpu.py tests/test_policy_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
