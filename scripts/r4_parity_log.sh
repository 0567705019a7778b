#!/bin/bash
# measured parity errors of the closing build, as printed by the tests themselves (-s)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 2400 python -m pytest tests/test_clip_gpu.py tests/test_policy_gpu.py tests/test_m3ae_gpu.py tests/test_finetune_gpu.py -q -m gpu -s -k "full_size_parity or latency_path_full_size or heavy_tailed or sixteen_seeds or encoder_inside_full or 16bit_modes_track or online_reward_family or online_adapter or hi_lo or image_text_input_on or forward_parity" 2>&1 | grep -E "err|seeds|cosine|passed|failed|differ" ) > $O/r4_parity_log.txt
tail -5 $O/r4_parity_log.txt
