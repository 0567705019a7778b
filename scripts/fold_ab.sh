for f in 0 1; do
  ARP_LN_FOLD=$f python bench.py --mode bf16 --no-secondary --cpu-seconds 0 --parity-frames 8 --timed-only --steps 30 2>/dev/null > /tmp/f_$f.json
  python -c "import json; d=json.load(open('/tmp/f_$f.json')); print('bf16 ARP_LN_FOLD=$f', round(d['value']), round(d['ms_per_step'],3), d['parity']['max_cosine_err_vs_oracle'])"
done
