#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
./scripts/gemm256_abl4.bin > $O/r4_abl4.txt 2>&1
G256_FLAGS=1 ./scripts/gemm256_bench.bin > $O/r4_nostore.txt 2>&1
(timeout 1800 python -m pytest tests/test_clip_gpu.py tests/test_policy_gpu.py tests/test_m3ae_gpu.py tests/test_finetune_gpu.py -q -m gpu -s -k "full_size_parity or latency_path_full_size or heavy_tailed or sixteen_seeds or encoder_inside_full or 16bit_modes_track or online_reward_family or online_adapter" 2>&1 | grep -E "err|seeds|cosine|passed|failed" ) > $O/r4_parity_log.txt
(time python bench.py) > $O/r4_bench_default.json 2> $O/r4_bench_default.err
grep -E "^(c_proj|out_proj|c_fc )" $O/r4_abl4.txt $O/r4_nostore.txt | cut -c1-190; cat $O/r4_parity_log.txt | cut -c1-260; head -c 2600 $O/r4_bench_default.json; echo; tail -4 $O/r4_bench_default.err
