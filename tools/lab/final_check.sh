#!/bin/bash
# the driver's round-end sequence on one fresh box: GPU tests (-x), smoke(), default bench
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6check
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gputests_x.log 2>&1; echo "suite rc=$?" >> $OUT/gputests_x.log
tail -3 $OUT/gputests_x.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6check/bench_default.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], 'nt8', r['frac'], 'traffic', r.get('traffic'), 'enc', r['encoder']['frac'], r['encoder']['ms'], 'sampler', d['sampler']['value'], 'cpu', d['cpu_baseline']['value'])
PY
