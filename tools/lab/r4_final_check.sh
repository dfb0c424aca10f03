#!/bin/bash
# round 4 closing run on the FINAL sources: PMC refresh (roofline.traffic record of this build), then exactly what the
# driver runs -- pytest -m gpu -x, smoke(), python bench.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4check; mkdir -p $OUT
bash tools/pmc_refresh.sh r4check > $OUT/pmc_refresh.log 2>&1
cp $OUT/pmc_gemm_nt.json profiles/pmc_gemm_nt.json
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gputests_x.log 2>&1
echo "suite rc=$?" >> $OUT/gputests_x.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/gputests_x.log; tail -2 $OUT/smoke.log; cut -c1-400 $OUT/bench_default.json; python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['roofline']['traffic'], d['roofline']['frac'], d['roofline']['encoder']['frac'])"
