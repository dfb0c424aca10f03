#!/bin/bash
# round 5, GPU call 4: full GPU suite + default bench on the current build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=8 > $OUT/gputests_a.log 2>&1
echo "suite rc=$?" >> $OUT/gputests_a.log
tail -25 $OUT/gputests_a.log
timeout 400 python bench.py > $OUT/bench_default_a.json 2> $OUT/bench_default_a.err
cut -c1-900 $OUT/bench_default_a.json
