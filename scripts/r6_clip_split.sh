#!/bin/bash
# labelling pass: two part streams cut evenly (ARP_CLIP_SPLIT=-1) or unevenly (default 15/32, 448, 416), ViT-B/32 at 1024 and ViT-B/16 at 256 frames
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { L=$1; shift; "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], (d.get('parity') or {}).get('err'))"; }
for rep in 1 2 3; do
  for s in -1 0 448 416; do one "B/32 1024 split $s:" env ARP_CLIP_SPLIT=$s python bench.py --no-secondary --cpu-seconds 0 --timed-only; done
  for s in -1 0 112 104; do one "B/16 256 split $s:" env ARP_CLIP_SPLIT=$s python bench.py --no-secondary --cpu-seconds 0 --timed-only --model ViT-B/16 --batch 256; done
done
