#!/bin/bash
# round 5, run 7: the policy step's update and the Infinity Cache (dWi last, Adam from the end), image_text_input with two K-tiles in flight
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run7.txt
rm -f $F
echo "== scripts/l3_probe.bin" >> $F
timeout 120 scripts/l3_probe.bin >> $F 2>&1
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'))" >> $F 2>&1; }
for rep in 1 2; do
echo "-- default (dWi last, Adam reversed, 2 K-tiles ahead)" >> $F; pol
echo "-- ARP_DT_DWI_LAST=0 ARP_DT_ADAM_REV=0 ARP_DT_ITI_AHEAD=1 (round-5 closing state)" >> $F; ARP_DT_DWI_LAST=0 ARP_DT_ADAM_REV=0 ARP_DT_ITI_AHEAD=1 pol
echo "-- ARP_DT_DWI_LAST=0" >> $F; ARP_DT_DWI_LAST=0 pol
echo "-- ARP_DT_ADAM_REV=0" >> $F; ARP_DT_ADAM_REV=0 pol
echo "-- ARP_DT_ITI_AHEAD=1" >> $F; ARP_DT_ITI_AHEAD=1 pol
done
echo "-- ARP_DT_ITI_WGS=512, ahead 2 / ahead 1" >> $F
ARP_DT_ITI_WGS=512 pol
ARP_DT_ITI_WGS=512 ARP_DT_ITI_AHEAD=1 pol
echo "-- ARP_DT_ITI_WGS=384, ahead 2" >> $F
ARP_DT_ITI_WGS=384 pol
echo "== tests" >> $F
(timeout 1500 python -m pytest tests/test_policy_gpu.py -q -m gpu -x 2>&1 | tail -3) >> $F
cat $F
