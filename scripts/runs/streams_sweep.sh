#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4_streams.txt
rm -f $O
for s in 1 2 3 4 2; do
  echo "== --streams $s" >> $O
  timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])" >> $O 2>&1
done
cat $O
