for s in 1 2 3 4; do
  python bench.py --streams $s --no-secondary --cpu-seconds 0 --parity-frames 0 --timed-only --steps 30 2>/dev/null > /tmp/b_$s.json
  python -c "import json; d=json.load(open('/tmp/b_$s.json')); print('streams $s', round(d['value']), round(d['ms_per_step'],3))"
done
