#!/bin/bash
# round 4 closing evidence: full GPU suite, default bench line, rocprofv3 summaries
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5) > $O/r4_t_all.txt 2>&1
(time python bench.py) > $O/r4_bench_default.json 2> $O/r4_bench_default.err
./scripts/prof_round.sh r4 > $O/prof_round.log 2>&1
cat $O/r4_t_all.txt; head -c 1800 $O/r4_bench_default.json; echo; tail -3 $O/r4_bench_default.err
