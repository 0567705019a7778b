#!/bin/bash
# round 5, run 12: dY kernel on the binary16 encodings, bit-identity test of the round-5 launch forms, same-box reference
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run12.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'], d['final_aux']['loss'], d['final_aux']['weight_l2'])" >> $F 2>&1; }
for rep in 1 2 3; do
echo "-- default" >> $F; pol
echo "-- ARP_DT_DY_X16=0" >> $F; ARP_DT_DY_X16=0 pol
echo "-- ARP_DT_MERGE=0 ARP_DT_DY_X16=0" >> $F; ARP_DT_MERGE=0 ARP_DT_DY_X16=0 pol
echo "-- round-5 closing state (alt/iti_r4, every switch at its old value)" >> $F; ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so ARP_DT_ITI_MIX=0 ARP_DT_DWI_LAST=0 ARP_DT_ADAM_REV=0 ARP_DT_MERGE=0 ARP_DT_DY_X16=0 pol
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|error|Error|assert|residual_weight" | tail -8) >> $F
cat $F
