#!/bin/bash
# round 5, run 19: image_text_input with two workgroups per CU (the fused binary16-x instance needs 240 registers)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run19.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2; do
echo "-- 256 (default)" >> $F; pol
echo "-- ARP_DT_ITI_WGS=512" >> $F; ARP_DT_ITI_WGS=512 pol
echo "-- ARP_DT_ITI_WGS=384" >> $F; ARP_DT_ITI_WGS=384 pol
done
cat $F
