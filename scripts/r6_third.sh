#!/bin/bash
# round 6, third GPU call: suite; N1 A/B after the stream-count fix; adapter plans (time + error); h5 stages; epilogue ablations of the c_fc kernel; ViT-B/16 batch; bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
mkdir -p $O
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > $O/r6_gpu_suite_third.txt 2>&1
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
p=d.get('parity') or {}
print('$L', 'ms_per_step', d['ms_per_step'], 'value', round(d['value'],1), 'parity', p.get('max_logit_err_vs_oracle', p.get('max_cosine_err_vs_oracle')), dict(list(s.items())[:6]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 20 --warmup 5 --cpu-seconds 0 --parity-frames 0 --no-secondary"
{
for rep in 1 2; do
  echo "== N1 f16c rep $rep"
  one "r5-like: 1 stream, at head, vperm off, plan 22e, H1 copy" env ARP_ENC_STREAMS=1 ARP_F16C_VPERM=0 ARP_DT_ADAPTER_PLAN=22e ARP_DT_ADAPTER_H1_INPLACE=0 $N1 --no-encode-ahead
  one "2 streams, at head, vperm off, plan 22e                  " env ARP_F16C_VPERM=0 ARP_DT_ADAPTER_PLAN=22e $N1 --no-encode-ahead
  one "2 streams, at head, vperm on,  plan 22e                  " env ARP_DT_ADAPTER_PLAN=22e $N1 --no-encode-ahead
  one "2 streams, AHEAD,   vperm on,  plan 22e                  " env ARP_DT_ADAPTER_PLAN=22e $N1
  one "2 streams, AHEAD,   vperm on,  plan 22h                  " env ARP_DT_ADAPTER_PLAN=22h $N1
  one "2 streams, AHEAD,   vperm on,  plan 12h                  " env ARP_DT_ADAPTER_PLAN=12h $N1
  one "2 streams, AHEAD,   vperm on,  plan 11h                  " env ARP_DT_ADAPTER_PLAN=11h $N1
  one "1 stream,  AHEAD,   vperm on,  plan 22e                  " env ARP_ENC_STREAMS=1 ARP_DT_ADAPTER_PLAN=22e $N1
  one "plain f16 encoder, 2 streams, AHEAD                      " python bench.py --path policy --with-encoder --mode f16 --steps 20 --warmup 5 --cpu-seconds 0 --parity-frames 0 --no-secondary
  one "plain f16 encoder, 2 streams, at head                    " python bench.py --path policy --with-encoder --mode f16 --steps 20 --warmup 5 --cpu-seconds 0 --parity-frames 0 --no-secondary --no-encode-ahead
done
} > $O/r6_n1_ab.txt 2>&1
PL="python bench.py --path policy --mode f16 --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== policy alone rep $rep"
  one "no corrections          " $PL
  one "plan 22e, H1 copy (r5)  " env ARP_DT_ADAPTER_PLAN=22e ARP_DT_ADAPTER_H1_INPLACE=0 $PL --adapter-c
  for P in 22e 22h 12e 12h 21h 11h; do
    one "plan $P                 " env ARP_DT_ADAPTER_PLAN=$P $PL --adapter-c
  done
done
} > $O/r6_adapter_plans.txt 2>&1
python scripts/adapter_plan_gpu.py >> $O/r6_adapter_plans.txt 2>&1
python scripts/h5_first_pass.py 4096 > $O/r6_h5_first_pass.txt 2>&1
{
for rep in 1 2 3; do
  echo "== c_fc epilogue ablations, rep $rep (scripts/gemm256_bench.hip; both columns are the same kernel)"
  echo "-- full kernel";                    scripts/gemm256_bench_abl0.bin 2>&1 | grep -E "^c_fc |^qkv |^c_fc_half"
  echo "-- no epilogue stores (K loop + staging VALU only; G256_FLAGS=1)"; G256_FLAGS=1 scripts/gemm256_bench_abl0.bin 2>&1 | grep -E "^c_fc |^qkv |^c_fc_half"
  echo "-- no bias / QuickGELU arithmetic (ARP_G2_ABL=8)"; scripts/gemm256_bench_abl8.bin 2>&1 | grep -E "^c_fc |^qkv |^c_fc_half"
done
} > $O/r6_epilogue_ablation.txt 2>&1
{
for B in 256 512 1024; do
  one "ViT-B/16 batch $B" python bench.py --model ViT-B/16 --batch $B --cpu-seconds 0 --no-secondary --steps 12 --warmup 3
done
} > $O/r6_b16_batch.txt 2>&1
(time python bench.py) > $O/r6_bench_third.jsonl 2> $O/r6_bench_third.err
cp $O/bench_full.json $O/r6_bench_third_full.json
tail -8 $O/r6_gpu_suite_third.txt; cut -c1-200 $O/r6_n1_ab.txt; cut -c1-260 $O/r6_adapter_plans.txt; tail -20 $O/r6_h5_first_pass.txt; cat $O/r6_epilogue_ablation.txt | cut -c1-200; cat $O/r6_b16_batch.txt | cut -c1-200; tail -c 1500 $O/r6_bench_third.jsonl
