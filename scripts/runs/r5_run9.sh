#!/bin/bash
# round 5, run 9: the adapter's mix inside image_text_input's operand load; same-box A/B against the round-4 kernel (alt/iti_r4) and the old launch order
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run9.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d.get('parity'))" >> $F 2>&1; }
for rep in 1 2; do
echo "-- default (mix inside image_text_input, dWi last, Adam reversed)" >> $F; pol
echo "-- ARP_DT_ITI_MIX=0" >> $F; ARP_DT_ITI_MIX=0 pol
echo "-- round-5 closing state: alt/iti_r4 kernel, ARP_DT_ITI_MIX=0 ARP_DT_DWI_LAST=0 ARP_DT_ADAM_REV=0" >> $F; ARP_LIB=arp_amd/alt/iti_r4/libarp_hip.so ARP_DT_ITI_MIX=0 ARP_DT_DWI_LAST=0 ARP_DT_ADAM_REV=0 pol
done
echo "-- default, 512 / 238x2 slices" >> $F
ARP_DT_ITI_WGS=512 pol
echo "-- with the encoder in front (f16c), default / ARP_DT_ITI_MIX=0" >> $F
pol --with-encoder --mode f16 --encoder-mode f16c --steps 10 --warmup 3
ARP_DT_ITI_MIX=0 pol --with-encoder --mode f16 --encoder-mode f16c --steps 10 --warmup 3
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5) >> $F
cat $F
