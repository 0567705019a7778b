#!/bin/bash
# closing checks: build() is a no-op on an up-to-date tree, smoke(), the full GPU suite (clean log for profiles/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time python -c "import __graft_entry__ as g; g.build(); g.smoke()") > $O/r5_smoke.txt 2>&1
(time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5) > $O/r5_gpu_suite.txt 2>&1
tail -8 $O/r5_smoke.txt; cat $O/r5_gpu_suite.txt
